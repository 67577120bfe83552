/* Canonical synthetic FASTQ of SURVEY.md Appendix B / BASELINE.md section 4:
 * a counter-based definition, so that any shard of the stream can be produced
 * independently and identically on the host (oracle/synth_ref.c) and on the
 * device (tagdigger_amd/csrc/synth.hip).  Bench / test infrastructure only --
 * nothing here is on the counting path.
 *
 * record(i) = "@r%012llu\n" SEQ(i) "\n+\n" 'I'*L "\n"        (2L+19 bytes, 219 at L=100)
 *   kind = rnd(i,0) % 100 :  <70 barcode+tag | <85 barcode+cutsite+random
 *                            <95 random      | else barcode+tag with one base -> 'N'
 *   j = rnd(i,1) % nbar, k = rnd(i,2) % ntags, c = rnd(i,3) % ncut, npos = rnd(i,4) % len(bc+tag)
 *   SEQ(i)[p] = body[p] while the body lasts, then "ACGT"[(rnd(i,16+p/32) >> 2*(p%32)) & 3]
 * Tags are expected to begin with a cut site (so a hit is barcode+tag).
 *
 * Two optional variants (both off in the canonical stream):
 *   skew      j and k are drawn from given distributions instead of uniformly: bar_cdf / tag_cdf are
 *             tables of nbar / ntags ascending 64-bit thresholds, the last one 2^64-1, and the draw is
 *             the first index whose threshold exceeds rnd(i,1) / rnd(i,2) (integer compares only, so
 *             host and device agree bit for bit whatever built the table: tagdigger_amd/synth.py, Zipf)
 *   adapter   read-through: in adapter_pct % of the reads of kind barcode+tag
 *             (N-kind included), chosen by rnd(i,5) % 100, the body is followed by `adapter`
 *             (what is left of the common cutter's site + the start of its adapter, reference
 *             tagdigger_fun.py:27-28) instead of random bases, as far as the read reaches
 */
#ifndef TD_SYNTH_SPEC_H
#define TD_SYNTH_SPEC_H
#include <stdint.h>

#ifdef __HIPCC__
#define TD_SYNTH_FN __host__ __device__ static inline
#else
#define TD_SYNTH_FN static inline
#endif

#define TD_SYNTH_BAR_STRIDE 16   /* bytes per barcode slot in the barcode table */
#define TD_SYNTH_CUT_STRIDE 16   /* bytes per concrete cut site                 */
#define TD_SYNTH_HDR_BYTES 15    /* "@r" + 12 digits + '\n'                     */
#define TD_SYNTH_ADAPTER_MAX 64  /* bytes of read-through sequence kept in the parameters */

typedef struct {
    uint64_t seed;
    uint32_t nbar, ntags, ncut;
    uint32_t read_len;      /* L */
    uint32_t cut_len;       /* all concrete cut sites share a length */
    uint32_t tag_stride;    /* bytes per tag slot in tag table */
    uint32_t adapter_pct;   /* 0 = no read-through */
    uint32_t adapter_len;   /* bytes of `adapter` in use (<= TD_SYNTH_ADAPTER_MAX) */
    const uint64_t *tag_cdf;   /* NULL = uniform; else ntags thresholds (address valid where the generator runs) */
    const uint64_t *bar_cdf;   /* NULL = uniform; else nbar thresholds */
    char adapter[TD_SYNTH_ADAPTER_MAX];
} td_synth_params;

/* first index whose threshold exceeds r (cdf ascending, cdf[n-1] = 2^64-1); uniform when cdf is NULL */
TD_SYNTH_FN uint32_t td_synth_pick(uint64_t r, uint32_t n, const uint64_t *cdf) {
    if (!cdf) return (uint32_t)(r % n);
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (r < cdf[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

TD_SYNTH_FN uint64_t td_mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31; return z;
}
TD_SYNTH_FN uint64_t td_rnd(uint64_t seed, uint64_t i, uint32_t stream) {
    return td_mix64(td_mix64(seed * 0x9E3779B97F4A7C15ULL + i) +
                    (uint64_t)(stream + 1) * 0xD1B54A32D192ED03ULL);
}
TD_SYNTH_FN uint32_t td_synth_record_bytes(uint32_t read_len) { return 2u * read_len + 19u; }

/* Writes record i (exactly td_synth_record_bytes bytes) at out. */
TD_SYNTH_FN void td_synth_record(const td_synth_params *P, uint64_t i,
                                 const char *bar_tab, const uint8_t *bar_len,
                                 const char *cut_tab,
                                 const char *tag_tab, const uint16_t *tag_len,
                                 uint8_t *out) {
    const uint32_t L = P->read_len;
    uint8_t *o = out;
    *o++ = '@'; *o++ = 'r';
    uint64_t v = i;
    for (int d = 11; d >= 0; d--) { o[d] = (uint8_t)('0' + (v % 10)); v /= 10; }
    o += 12; *o++ = '\n';

    const uint32_t kind100 = (uint32_t)(td_rnd(P->seed, i, 0) % 100u);
    const uint32_t kind = kind100 < 70 ? 0u : kind100 < 85 ? 1u : kind100 < 95 ? 2u : 3u;
    const uint32_t j = td_synth_pick(td_rnd(P->seed, i, 1), P->nbar, P->bar_cdf);
    const uint32_t k = td_synth_pick(td_rnd(P->seed, i, 2), P->ntags, P->tag_cdf);
    const uint32_t c = (uint32_t)(td_rnd(P->seed, i, 3) % P->ncut);
    const uint32_t bl = bar_len[j], tl = tag_len[k];
    uint32_t body = 0, npos = 0xFFFFFFFFu, through = 0;
    if (kind == 0 || kind == 3) body = bl + tl;
    else if (kind == 1) body = bl + P->cut_len;
    if (kind == 3) npos = (uint32_t)(td_rnd(P->seed, i, 4) % (bl + tl));
    if ((kind == 0 || kind == 3) && P->adapter_pct && (uint32_t)(td_rnd(P->seed, i, 5) % 100u) < P->adapter_pct)
        through = P->adapter_len;
    uint64_t rw = 0;
    for (uint32_t p = 0; p < L; p++) {
        if ((p & 31u) == 0) rw = td_rnd(P->seed, i, 16u + (p >> 5));
        char ch;
        if (p < body) {
            if (p < bl) ch = bar_tab[(uint64_t)j * TD_SYNTH_BAR_STRIDE + p];
            else if (kind == 1) ch = cut_tab[(uint64_t)c * TD_SYNTH_CUT_STRIDE + (p - bl)];
            else ch = tag_tab[(uint64_t)k * P->tag_stride + (p - bl)];
        } else if (p - body < through) {
            ch = P->adapter[p - body];
        } else {
            ch = "ACGT"[(rw >> (2u * (p & 31u))) & 3u];
        }
        if (p == npos) ch = 'N';
        *o++ = (uint8_t)ch;
    }
    *o++ = '\n'; *o++ = '+'; *o++ = '\n';
    for (uint32_t p = 0; p < L; p++) *o++ = 'I';
    *o++ = '\n';
}

/* What record i contributes to the count matrix when barcode+site and tag
 * sets are prefix-free and bc+tag fits in the read: kind 0 -> (j,k), else none.
 * Returns 1 and sets j,k for a counted hit. */
TD_SYNTH_FN int td_synth_hit(const td_synth_params *P, uint64_t i, uint32_t *j, uint32_t *k) {
    const uint32_t kind100 = (uint32_t)(td_rnd(P->seed, i, 0) % 100u);
    if (kind100 >= 70) return 0;
    *j = td_synth_pick(td_rnd(P->seed, i, 1), P->nbar, P->bar_cdf);
    *k = td_synth_pick(td_rnd(P->seed, i, 2), P->ntags, P->tag_cdf);
    return 1;
}
#endif
