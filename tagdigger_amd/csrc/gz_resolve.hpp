// Step 4 of the chunk-parallel gzip decoder (par_inflate.hpp) on the GPU: the host threads decode an ordinary gzip
// stream into 16-bit SYMBOLS (a literal as itself, a copy that reaches into the unknown 32 KiB before its chunk as
// 0x8000 | position in that window) straight into pinned memory; the symbols go to the device by DMA, and here they
// become bytes where the count kernel wants them, and every 64 KiB of them gets its CRC-32 (the host joins those and
// checks each member's trailer).  What the reference reads through gzip.open (tagdigger_fun.py:240-241).
// Round 3: takes markers + CRC (15 % of the host decoder's thread time) and the reader's copy into the staging buffer
// off the 16 host threads that bound this tier.
#pragma once
#include <stdint.h>
#include "gpu_inflate.hpp"

namespace tdgz {

constexpr uint32_t BLOCK_SYMS = 65536;      // symbols (= output bytes) per block: one workgroup resolves it, one lane takes its CRC-32
constexpr uint32_t WINDOW = 32768;

struct Block {
    uint64_t src_off;       // its symbols (2 bytes each; 1 byte each when `narrow`) in the batch's upload
    uint64_t dst_off;       // where its bytes go in the batch's output (the kernels' d_out points behind the carried bytes)
    uint32_t len;           // symbols = bytes
    uint32_t win;           // its chunk's window: d_win + 32768 * win
    uint32_t min_idx;       // a marker below it points before the member's start: invalid stream
    uint32_t narrow;        // the symbols are bytes already (the batch's first chunk)
};

__global__ __launch_bounds__(256) void k_gz_resolve(const uint8_t *d_sym, const uint8_t *d_win, uint8_t *d_out, const Block *blk,
                                                    uint32_t nblk, uint32_t *flag) {
    const uint32_t b = blockIdx.x;
    if (b >= nblk) return;
    const Block k = blk[b];
    uint8_t *dst = d_out + k.dst_off;
    if (k.narrow) {
        const uint8_t *src = d_sym + k.src_off;
        for (uint32_t i = threadIdx.x * 8u; i < k.len; i += 256u * 8u) {
            if (i + 8u <= k.len) { uint64_t v; __builtin_memcpy(&v, src + i, 8); __builtin_memcpy(dst + i, &v, 8); }
            else for (uint32_t q = i; q < k.len; q++) dst[q] = src[q];
        }
        return;
    }
    const uint16_t *src = reinterpret_cast<const uint16_t *>(d_sym + k.src_off);       // (2-byte aligned: offsets are even)
    const uint8_t *win = d_win + (size_t)k.win * WINDOW;
    bool bad = false;
    for (uint32_t i = threadIdx.x * 8u; i < k.len; i += 256u * 8u) {
        const uint32_t n = k.len - i < 8u ? k.len - i : 8u;
        uint16_t s[8];
        if (n == 8u) __builtin_memcpy(s, src + i, 16);
        else for (uint32_t q = 0; q < 8u; q++) s[q] = q < n ? src[i + q] : (uint16_t)0;
        uint64_t v = 0;
#pragma unroll
        for (uint32_t q = 0; q < 8u; q++) {
            uint32_t x = s[q];
            if (x & 0x8000u) {
                const uint32_t idx = x & 0x7FFFu;
                bad |= idx < k.min_idx;
                x = win[idx];
            }
            v |= (uint64_t)(x & 0xFFu) << (8u * q);
        }
        if (n == 8u) __builtin_memcpy(dst + i, &v, 8);
        else for (uint32_t q = 0; q < n; q++) dst[i + q] = (uint8_t)(v >> (8u * q));
    }
    if (bad) atomicOr(flag, 1u);
}

// one lane per block: CRC-32 of its bytes (slicing-by-4 tables in LDS, as for the BGZF members)
__global__ __launch_bounds__(64) void k_gz_crc(const uint8_t *d_out, const Block *blk, uint32_t nblk, const uint32_t *crc_tables, uint32_t *crc) {
    __shared__ uint32_t T[1024];
    for (uint32_t k = threadIdx.x; k < 1024u; k += 64u) T[k] = crc_tables[k];
    __syncthreads();
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= nblk) return;
    crc[b] = tdinf::crc32_lds(d_out + blk[b].dst_off, blk[b].len, T);
}

}  // namespace tdgz
