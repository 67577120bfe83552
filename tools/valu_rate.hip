// Microbenchmark: integer VALU issue rate per SIMD on gfx950 (wave64), for 1..8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void k(uint32_t *out, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) a[i] = (a[i] ^ 0x0A0A0A0Au) + 0x7F7F7F7Fu;                        // xor + add
                if (KIND == 1) a[i] = __builtin_amdgcn_udot4(a[i], 0x08040201u, a[(i + 1) & 7], false);   // dot4
                if (KIND == 2) a[i] = __builtin_amdgcn_perm(a[i], 0x47544341u, a[(i + 1) & 7] & 0x03030303u) ^ a[i];  // and+perm+xor
                if (KIND == 3) a[i] = a[i] * 0x9E3779B1u + 1u;                                    // mul_lo + add
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    const char *names[4] = {"xor+add (2 ops)", "dot4 (1 op)", "and+perm+xor (3 ops)", "mul_lo+add (2 ops)"};
    const int ops[4] = {2, 1, 3, 2};
    for (int kind = 0; kind < 4; kind++)
        for (int wps = 1; wps <= 8; wps *= 2) {
            dim3 grid(256 * wps), block(256);   // 256-thread blocks: one wave per SIMD each
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters);
                if (kind == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters);
                if (kind == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, iters);
                if (kind == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double instr_per_simd = (double)iters * 16 * 8 * ops[kind] * wps;   // wave-instructions issued on one SIMD
            printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.1 GHz)\n",
                   names[kind], wps, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.1);
        }
    return 0;
}
