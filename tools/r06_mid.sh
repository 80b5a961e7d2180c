#!/bin/bash
# round 6, mid-round measurement: hot-set tests, hot-set bench line, kernel stats of one step, the opening's exposed time
OUT=gpurun_out/r06mid; mkdir -p $OUT
python -m pytest tests -q -m gpu -x -k "hot_set or rows_handle" 2>&1 | tail -4 > $OUT/pytest_hotset.txt; cat $OUT/pytest_hotset.txt
python bench.py --ram-addresses hotset --no-cpu-baseline --no-upload-rate --no-msm-roofline --steps 8 --warmup 2 > $OUT/bench_hotset.json 2>/dev/null
python bench.py --no-cpu-baseline --no-upload-rate --no-msm-roofline --steps 8 --warmup 2 > $OUT/bench_uniform.json 2>/dev/null
for f in bench_hotset bench_uniform; do python -c "import json; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['config'].get('ms_per_step_split'))"; done
bash tools/prof_step.sh r06mid/step > /dev/null 2>&1
head -45 $OUT/step/bench_kernel_stats.txt | cut -c1-150
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_os && timeout 500 rocprofv3 --kernel-trace -d /tmp/p_os -o o -- python $GRAFT_REPO_ROOT/tools/open_step.py 22 1 > $GRAFT_REPO_ROOT/$OUT/open_step.txt 2>&1; f=$(find /tmp/p_os -name "*.db" | head -1); python $GRAFT_REPO_ROOT/profiles/open_exposed.py "$f" 34 > $GRAFT_REPO_ROOT/$OUT/open_exposed_step.txt 2>&1 )
cat $OUT/open_exposed_step.txt | head -40
