// Explicit instantiations of tdk::k_fast2 for one tile size and one recording mode (-DTD_INST_CPT=4|6|8
// -DTD_INST_PROG=0|1): the 54 instantiations are the bulk of the build, six translation units compile side by side
// (make -j).  tagdig.hip only declares them (TD_FAST2_EXTERN).
#include <hip/hip_runtime.h>
#define TD_INST_ONLY 1                // the non-template kernels of the shared headers belong to tagdig.hip
#include "kernel_fast2.hpp"
#define X(W, NQ) template __global__ void tdk::k_fast2<TD_INST_CPT, W, NQ, (TD_INST_PROG != 0)>(const tdk::FParams);
TD_FAST2_COMBOS(X)
#undef X
