# round 5, first GPU call: the 8-rank shared-GPU cases the round-4 review asked for (a code-path check, not a measurement)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" 2>/dev/null   # page the image in once
( time timeout 900 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu -k "test_sharded_stage_operators_prove_one_trace and 8-4-kw4" --durations=5 ) > gpurun_out/r05_8rank_stage_ops.txt 2>&1
echo "rc $?" >> gpurun_out/r05_8rank_stage_ops.txt
tail -15 gpurun_out/r05_8rank_stage_ops.txt
run_shared () {  # $1 world, $2 scale, $3 out
  export JOLT_BENCH_SHARE_GPU=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + $1 + $2)) WORLD_SIZE=$1
  local pids=""
  for r in $(seq 1 $(($1 - 1))); do
    RANK=$r LOCAL_RANK=$r timeout 600 python bench.py --gpus $1 --scale $2 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/b$r.out 2> /tmp/b$r.err &
    pids="$pids $!"
  done
  ( time RANK=0 LOCAL_RANK=0 timeout 600 python bench.py --gpus $1 --scale $2 --steps 1 --warmup 1 --no-cpu-baseline ) > $3 2> $3.err
  echo "rank0 rc $?" >> $3
  for p in $pids; do wait $p; echo "rc $?" >> $3; done
  grep -v "^\[Gloo\]" $3 | cut -c1-600
  tail -5 $3.err | cut -c1-400
  for r in $(seq 1 $(($1 - 1))); do tail -2 /tmp/b$r.err | cut -c1-300; done
}
run_shared 8 12 gpurun_out/r05_bench_gpus8_scale12_share_gpu.json
run_shared 8 14 gpurun_out/r05_bench_gpus8_scale14_share_gpu.json
