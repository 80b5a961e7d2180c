# the bench step with and without the sort token (JOLT_MSM_STAGGER), then the MSM / PCS parity tests with it on
for S in 0 1 0 1; do
  JOLT_MSM_STAGGER=$S timeout 300 python /root/repo/bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null > /tmp/sg.json
  python -c "import json;d=json.load(open('/tmp/sg.json'));print('stagger', $S, d['ms_per_step'], d['config']['ms_per_step_split'])"
done
JOLT_MSM_STAGGER=1 timeout 400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_pcs.py -q -m gpu -x 2>&1 | tail -2
