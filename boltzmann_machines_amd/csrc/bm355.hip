// bm355.hip — the single translation unit of libbm355.so.  The kernels live in headers
// (bm_kernels.h) shared by the RBM and DBM entry points, so both are compiled together; bm_rbm64.hip / bm_dbm64.hip are the
// float64 paths (own small kernels).
#include "bm_rbm.hip"
#include "bm_dbm.hip"
#include "bm_rbm64.hip"
#include "bm_dbm64.hip"
#include "bm_comm.hip"
#include "bm_xchg.hip"
