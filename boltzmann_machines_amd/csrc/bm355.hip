// bm355.hip — the single translation unit of libbm355.so.  The kernels live in headers
// (bm_kernels.h) shared by the RBM and DBM entry points, so both are compiled together.
#include "bm_rbm.hip"
#include "bm_dbm.hip"
