# the bench step with 1..4 concurrent MSM lanes: bench_lanes.sh
for L in 1 2 3 4; do
  JOLT_MSM_LANES=$L timeout 300 python /root/repo/bench.py --no-cpu-baseline --no-split --steps 4 --warmup 2 2>/dev/null > /tmp/l$L.json
  python -c "import json;d=json.load(open('/tmp/l$L.json'));print('lanes', $L, d['ms_per_step'])"
done
