/* Host reference of the synthetic FASTQ generator and of the count matrix it
 * implies -- TEST / BENCH INFRASTRUCTURE ONLY (see include/td_synth_spec.h).
 * The expected matrix is computed from the generator's own choices, without
 * parsing any FASTQ, so it is an independent known answer at any size. */
#include <stdint.h>
#include <stddef.h>
#include "../include/td_synth_spec.h"

void synth_fill_host(const td_synth_params *P, uint64_t first_read, uint64_t nreads,
                     const char *bar_tab, const uint8_t *bar_len, const char *cut_tab,
                     const char *tag_tab, const uint16_t *tag_len, uint8_t *out) {
    const uint64_t rb = td_synth_record_bytes(P->read_len);
    for (uint64_t r = 0; r < nreads; r++)
        td_synth_record(P, first_read + r, bar_tab, bar_len, cut_tab, tag_tab, tag_len, out + r * rb);
}

/* counts (nbar x ntags, row-major, uint64) += expected hits of reads
 * [first_read, first_read+nreads); returns the number of hits. */
uint64_t synth_expected(const td_synth_params *P, uint64_t first_read, uint64_t nreads, uint64_t *counts) {
    uint64_t hits = 0;
    for (uint64_t r = 0; r < nreads; r++) {
        uint32_t j, k;
        if (td_synth_hit(P, first_read + r, &j, &k)) { counts[(uint64_t)j * P->ntags + k]++; hits++; }
    }
    return hits;
}
