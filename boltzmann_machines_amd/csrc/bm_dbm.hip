// bm_dbm.hip — DBM entry points (placeholder until the DBM kernels land).
#include "../../include/bm355.h"
#include "bm_common.h"
#define NI(name) do { bm::set_error(name ": not implemented yet"); return 99; } while (0)
extern "C" {
int bm_dbm_create(const bm_dbm_config *, bm_dbm **) { NI("bm_dbm_create"); }
int bm_dbm_destroy(bm_dbm *) { return 0; }
int bm_dbm_sync(bm_dbm *) { NI("bm_dbm_sync"); }
int bm_dbm_seed(bm_dbm *, uint64_t) { NI("bm_dbm_seed"); }
int bm_dbm_set_row_offset(bm_dbm *, int64_t, int64_t) { NI("bm_dbm_set_row_offset"); }
int bm_dbm_set_param(bm_dbm *, const char *, const float *, size_t) { NI("bm_dbm_set_param"); }
int bm_dbm_get_param(bm_dbm *, const char *, float *, size_t) { NI("bm_dbm_get_param"); }
int bm_dbm_dev_ptr(bm_dbm *, const char *, void **, size_t *) { NI("bm_dbm_dev_ptr"); }
int bm_dbm_train_step(bm_dbm *, const float *, float, float, int32_t, int32_t *, float *) { NI("bm_dbm_train_step"); }
int bm_dbm_grad_step(bm_dbm *, const float *, int32_t, int32_t *) { NI("bm_dbm_grad_step"); }
int bm_dbm_apply_step(bm_dbm *, int32_t, int32_t, float, float) { NI("bm_dbm_apply_step"); }
int bm_dbm_mean_field(bm_dbm *, const float *, float *, int32_t *) { NI("bm_dbm_mean_field"); }
int bm_dbm_reconstruct(bm_dbm *, const float *, float *) { NI("bm_dbm_reconstruct"); }
int bm_dbm_sample_v(bm_dbm *, int32_t, float *) { NI("bm_dbm_sample_v"); }
int bm_dbm_ais(bm_dbm *, int32_t, int32_t, int32_t, uint64_t, int64_t, float *) { NI("bm_dbm_ais"); }
int bm_dbm_log_proba(bm_dbm *, const float *, float *) { NI("bm_dbm_log_proba"); }
int bm_dbm_timer_start(bm_dbm *) { NI("bm_dbm_timer_start"); }
int bm_dbm_timer_stop(bm_dbm *, float *) { NI("bm_dbm_timer_stop"); }
}
