// Explicit instantiations of tdk::k_fast4 (producer and consumer waves, kernel_fast4.hpp) for one recording mode
// (-DTD_INST_PROG=0|1): two translation units beside tagdig.hip and the six of k_fast2 (make -j).  tagdig.hip only declares
// them (TD_FAST4_EXTERN).
#include <hip/hip_runtime.h>
#define TD_INST_ONLY 1                // the non-template kernels of the shared headers belong to tagdig.hip
#include "kernel_fast4.hpp"
#define X(W, NQ) template __global__ void tdk::k_fast4<W, NQ, (TD_INST_PROG != 0)>(const tdk::FParams);
TD_FAST2_COMBOS(X)
#undef X
