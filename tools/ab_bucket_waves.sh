# A/B of the bucket kernels' occupancy attribute: rebuilds msm_fixed / msm / pcs / dory with -DJOLT_BUCKET_WAVES=<w> ON THE GPU BOX and times a 2^26-term fixed-base MSM
set -u
cd /root/repo
for W in 2 3 4; do
  touch jolt_amd/csrc/msm_kernels.hip.h
  JOLT_EXTRA_HIPCC_FLAGS="-DJOLT_BUCKET_WAVES=$W" python -m jolt_amd.build > /dev/null 2>&1 || { echo "build failed for $W"; continue; }
  echo "waves $W: $(JOLT_BENCH_PREFIXES="" timeout 300 python tools/bench_msm_fixed.py 26 23 2>/dev/null | tail -1)"
done
