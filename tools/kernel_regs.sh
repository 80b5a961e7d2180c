# VGPR / scratch use of the bucket kernels from the built objects: kernel_regs.sh
for o in msm_fixed msm dory pcs; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes jolt_amd/csrc/build/$o.o 2>/dev/null | grep -E "\.name:|\.vgpr_count|\.private_segment_fixed_size" | paste - - - | grep -E "buckets|onehot_sum" | sed 's/ \+/ /g' | cut -c1-200
done
