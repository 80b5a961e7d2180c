// TEMPORARY: MSM / HyperKZG entry points are implemented in msm.hip / hyperkzg.hip (next milestone).
#include "ctx.hpp"
#define UNSUP(sig) extern "C" __attribute__((visibility("default"))) int32_t sig { return JOLT_ERR_UNSUPPORTED; }
UNSUP(jolt_srs_upload_g1(jolt_ctx*, const jolt_g1_t*, size_t, jolt_srs**))
UNSUP(jolt_srs_setup_from_secret(jolt_ctx*, const jolt_fr_t*, size_t, const jolt_g1_t*, jolt_srs**))
UNSUP(jolt_srs_len(const jolt_srs*, size_t*))
UNSUP(jolt_srs_download(jolt_ctx*, const jolt_srs*, size_t, size_t, jolt_g1_t*))
UNSUP(jolt_srs_free(jolt_ctx*, jolt_srs*))
UNSUP(jolt_msm_g1(jolt_ctx*, const jolt_srs*, const jolt_fr_t*, size_t, jolt_g1_t*))
UNSUP(jolt_msm_g1_table(jolt_ctx*, const jolt_srs*, const jolt_table*, size_t, jolt_g1_t*))
UNSUP(jolt_hyperkzg_fold(jolt_ctx*, const jolt_table*, const jolt_fr_t*, size_t, jolt_table**))
UNSUP(jolt_hyperkzg_eval3(jolt_ctx*, jolt_table* const*, size_t, const jolt_fr_t*, jolt_fr_t*))
UNSUP(jolt_hyperkzg_rlc(jolt_ctx*, jolt_table* const*, size_t, const jolt_fr_t*, jolt_table**))
UNSUP(jolt_hyperkzg_witness_poly(jolt_ctx*, const jolt_table*, const jolt_fr_t*, jolt_table**))
UNSUP(jolt_host_g1_add(const jolt_g1_t*, const jolt_g1_t*, jolt_g1_t*))
UNSUP(jolt_host_g1_eq(const jolt_g1_t*, const jolt_g1_t*, int32_t*))
UNSUP(jolt_host_g1_serialize_compressed(const jolt_g1_t*, uint8_t*))
UNSUP(jolt_host_hyperkzg_commit(jolt_ctx*, const jolt_srs*, const jolt_table*, jolt_g1_t*))
UNSUP(jolt_host_hyperkzg_open(jolt_ctx*, const jolt_srs*, const jolt_table*, const jolt_fr_t*, size_t, uint64_t, jolt_g1_t*, jolt_g1_t*, jolt_fr_t*, jolt_fr_t*))
