# the bench step for several thread counts of the fixed-base bucket reduction
for D in 24 48 96 24 48; do
  JOLT_FX_REDUCE_DIV=$D timeout 300 python /root/repo/bench.py --no-cpu-baseline --no-split --steps 5 --warmup 2 2>/dev/null > /tmp/rd.json
  python -c "import json;d=json.load(open('/tmp/rd.json'));print('div', $D, d['ms_per_step'])"
done
