# bench.py's multi-rank branch at REAL size with both ranks on GPU 0 (gloo rendezvous): a code-path / memory check, not a measurement.
# usage: bench_two_ranks_one_gpu.sh [scale per rank = 21]
set -u
S=${1:-21}
cd /root/repo
export JOLT_BENCH_SHARE_GPU=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 WORLD_SIZE=2
RANK=1 LOCAL_RANK=1 timeout 900 python bench.py --gpus 2 --scale $S --steps 2 --warmup 1 --no-cpu-baseline > /tmp/r1.out 2> /tmp/r1.err &
P1=$!
RANK=0 LOCAL_RANK=0 timeout 900 python bench.py --gpus 2 --scale $S --steps 2 --warmup 1 --no-cpu-baseline > /tmp/r0.out 2> /tmp/r0.err
RC0=$?
wait $P1; RC1=$?
echo "rc $RC0 $RC1"
grep -v "^\[Gloo\]" /tmp/r0.out | cut -c1-1500
tail -3 /tmp/r0.err /tmp/r1.err | cut -c1-300
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
