/* CPU ORACLE (C restatement) -- TEST INFRASTRUCTURE ONLY, never shipped or
 * linked into the product library.
 *
 * Same algorithm as oracle/tagdigger_oracle.py (which see for the pinning
 * story), restated in scalar C so that full-size parity runs and the
 * cpu_baseline leg of bench.py finish in seconds.  Reference citations are
 * into tagdigger_fun.py of the reference checkout.
 *
 *   trie build     tree_one_level / tree_recursive / build_sequence_tree :71-113
 *   lookup         sequence_index_lookup                                 :115-134
 *   record loop    find_tags_fastq                                       :239-277
 *
 * The set-up of find_tags_fastq (:197-233: asserts, cut-site enumeration,
 * strip-or-shift branch) stays in Python (oracle/tagdigger_oracle.py
 * prepare_lists) and hands this file the final string lists.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum {
    ORC_OK = 0,
    ORC_ERR_OVERLAP = -1,    /* AssertionError "Problematic sequence: idx" (:82) */
    ORC_ERR_EMPTY = -2,      /* IndexError on an empty sequence list (:76)       */
    ORC_ERR_ROOTLEAF_A = -3, /* lookup into a leaf root, first base A: IndexError(str)  */
    ORC_ERR_ROOTLEAF_C = -4, /* first base C: TypeError                                  */
    ORC_ERR_ROOTLEAF_GT = -5,/* first base G/T: IndexError(list)                         */
    ORC_ERR_NONASCII = -6,   /* byte >= 0x80 in a sequence line                          */
    ORC_ERR_NOMEM = -7,
    ORC_ERR_TASSEL = -8      /* header without a parsable count= value (ValueError)      */
};

/* child value: 0 absent, >0 node id, <0 leaf holding index (-v-1) */
typedef struct { int32_t ch[4]; } node_t;
typedef struct {
    node_t *nodes; size_t n, cap;
    int root_leaf;       /* 1 when the root itself is a leaf */
    int32_t root_idx;
} trie_t;

static int base_code(unsigned char c) {
    switch (c & 0xDF) {          /* ASCII upper-casing of :256 folded in */
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
    default: return -1;
    }
}

static int32_t new_node(trie_t *t) {
    if (t->n == t->cap) {
        size_t nc = t->cap ? t->cap * 2 : 1024;
        node_t *p = (node_t *)realloc(t->nodes, nc * sizeof(node_t));
        if (!p) return -1;
        t->nodes = p; t->cap = nc;
    }
    memset(&t->nodes[t->n], 0, sizeof(node_t));
    return (int32_t)t->n++;
}

/* members[] = positions into seqs/idx, all sharing the first `depth` bases,
 * in input order.  Returns child value, or INT32_MIN on error (err set). */
static int32_t grow(trie_t *t, const char *const *seqs, const uint32_t *idx,
                    uint32_t *members, size_t m, size_t depth, int *err, uint32_t *bad) {
    if (seqs[members[0]][depth] == '\0')            /* :76-77 first one ends: leaf, rest dropped */
        return -(int32_t)idx[members[0]] - 1;
    for (size_t i = 0; i < m; i++)                  /* :82 */
        if (seqs[members[i]][depth] == '\0') { *err = ORC_ERR_OVERLAP; *bad = idx[members[i]]; return INT32_MIN; }
    int32_t me = new_node(t);
    if (me < 0) { *err = ORC_ERR_NOMEM; return INT32_MIN; }
    /* stable 4-way partition, child order A,C,G,T (:80-85) */
    uint32_t *tmp = (uint32_t *)malloc(m * sizeof(uint32_t));
    if (!tmp) { *err = ORC_ERR_NOMEM; return INT32_MIN; }
    size_t cnt[4] = {0, 0, 0, 0}, off[4];
    for (size_t i = 0; i < m; i++) { int c = base_code((unsigned char)seqs[members[i]][depth]); if (c < 0) c = 3; cnt[c]++; }
    off[0] = 0; for (int c = 1; c < 4; c++) off[c] = off[c - 1] + cnt[c - 1];
    size_t pos[4] = {off[0], off[1], off[2], off[3]};
    for (size_t i = 0; i < m; i++) { int c = base_code((unsigned char)seqs[members[i]][depth]); if (c < 0) c = 3; tmp[pos[c]++] = members[i]; }
    memcpy(members, tmp, m * sizeof(uint32_t));
    free(tmp);
    for (int c = 0; c < 4; c++) {
        if (!cnt[c]) continue;
        int32_t v = grow(t, seqs, idx, members + off[c], cnt[c], depth + 1, err, bad);
        if (v == INT32_MIN) return INT32_MIN;
        t->nodes[me].ch[c] = v;       /* re-index: nodes may have moved */
    }
    return me + 1;                    /* node ids are stored +1 so 0 stays "absent" */
}

static int trie_build(trie_t *t, const char *const *seqs, uint32_t n, uint32_t numseq, uint32_t *bad) {
    memset(t, 0, sizeof(*t));
    if (n == 0) return ORC_ERR_EMPTY;
    if (new_node(t) < 0) return ORC_ERR_NOMEM;       /* node 0 = root placeholder */
    if (numseq == 1 && n == 1 && seqs[0][0] == '\0') { /* :109-110 */
        for (int c = 0; c < 4; c++) t->nodes[0].ch[c] = -1;  /* leaf idx 0 */
        return ORC_OK;
    }
    uint32_t *idx = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *members = (uint32_t *)malloc(n * sizeof(uint32_t));
    if (!idx || !members) return ORC_ERR_NOMEM;
    uint32_t nxt = 0;
    for (uint32_t i = 0; i < n; i++) { idx[i] = nxt; members[i] = i; if (++nxt == numseq) nxt = 0; }  /* :102-108 */
    int err = ORC_OK;
    int32_t v = grow(t, seqs, idx, members, n, 0, &err, bad);
    free(idx); free(members);
    if (v == INT32_MIN) return err;
    if (v < 0) { t->root_leaf = 1; t->root_idx = -v - 1; return ORC_OK; }
    /* v-1 is the real root; it was created right after the placeholder */
    t->nodes[0] = t->nodes[v - 1];
    return ORC_OK;
}

/* :115-134.  s..e is the (already left-stripped) read; returns idx, -1, or an
 * ORC_ERR_ROOTLEAF_* code (<= -3). */
static int32_t trie_lookup(const trie_t *t, const unsigned char *s, const unsigned char *e) {
    const node_t *nd = &t->nodes[0];
    int first = 1;
    for (; s < e; s++) {
        int c = base_code(*s);
        if (c < 0) return -1;
        if (first && t->root_leaf)
            return c == 0 ? ORC_ERR_ROOTLEAF_A : c == 1 ? ORC_ERR_ROOTLEAF_C : ORC_ERR_ROOTLEAF_GT;
        first = 0;
        int32_t v = nd->ch[c];
        if (v == 0) return -1;
        if (v < 0) return -v - 1;
        nd = &t->nodes[v - 1];
    }
    return -1;
}

static int is_strip(unsigned char c) {   /* str.strip() of :256, ASCII range */
    return c == ' ' || (c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x1F);
}

typedef struct { trie_t bar, tag; } orc_index;

void orc_free(orc_index *ix) {
    if (!ix) return;
    free(ix->bar.nodes); free(ix->tag.nodes); free(ix);
}

/* Build both tries.  On ORC_ERR_OVERLAP *bad holds the offending index. */
int orc_build(const char *const *barcut, uint32_t n_barcut, uint32_t barnum,
              const char *const *tags, uint32_t ntags, orc_index **out, uint32_t *bad) {
    orc_index *ix = (orc_index *)calloc(1, sizeof(orc_index));
    if (!ix) return ORC_ERR_NOMEM;
    int rc = trie_build(&ix->bar, barcut, n_barcut, barnum, bad);
    if (rc == ORC_OK) rc = trie_build(&ix->tag, tags, ntags, ntags, bad);
    if (rc != ORC_OK) { orc_free(ix); return rc; }
    *out = ix;
    return ORC_OK;
}

/* The record loop.  data[0..n) holds lines starting at global line index
 * first_line; reads whose ordinal exceeds maxreads_eff are not processed
 * (maxreads_eff = max(1, ceil(maxreads)) computed by the caller, :272-273).
 * counts is barnum x ntags, row-major, accumulated into (not cleared).
 * stats[0..2] += reads, barcode+site hits, tag hits; stats[3] = lines seen. */
int orc_count(const orc_index *ix, const uint8_t *data, uint64_t n, uint64_t first_line,
              uint64_t maxreads_eff, int tassel, const uint32_t *barcutlen, uint32_t ntags,
              uint64_t *counts, uint64_t *stats) {
    uint64_t line = first_line, i = 0;
    uint64_t weight = 1;
    uint64_t lines_seen = 0;
    while (i < n) {
        uint64_t s = i;
        while (i < n && data[i] != '\n' && data[i] != '\r') i++;
        uint64_t e = i;                           /* line body is [s,e) */
        if (i < n) { if (data[i] == '\r' && i + 1 < n && data[i + 1] == '\n') i += 2; else i++; }
        lines_seen++;
        unsigned ph = (unsigned)(line & 3);
        if (ph == 0 && tassel) {                  /* :251-253 */
            const char *key = "count=";
            int64_t at = -1;
            for (uint64_t p = s; p + 6 <= e; p++) if (!memcmp(data + p, key, 6)) { at = (int64_t)(p - s); break; }
            uint64_t p = s + (uint64_t)(at + 6);   /* find()==-1 gives slice [5:] */
            uint64_t q = e;
            while (p < q && is_strip(data[p])) p++;
            while (q > p && is_strip(data[q - 1])) q--;
            if (p >= q) return ORC_ERR_TASSEL;
            uint64_t v = 0; int neg = 0;
            if (data[p] == '+' || data[p] == '-') { neg = data[p] == '-'; p++; if (p >= q) return ORC_ERR_TASSEL; }
            for (; p < q; p++) { if (data[p] < '0' || data[p] > '9') return ORC_ERR_TASSEL; v = v * 10 + (data[p] - '0'); }
            weight = neg ? (uint64_t)(-(int64_t)v) : v;
        }
        if (ph == 1) {                            /* :254 */
            stats[0]++;
            for (uint64_t p = s; p < e; p++) if (data[p] >= 0x80) return ORC_ERR_NONASCII;
            const unsigned char *rs = data + s, *re = data + e;
            while (rs < re && is_strip(*rs)) rs++;
            int32_t bar = trie_lookup(&ix->bar, rs, re);             /* :257 */
            if (bar <= -3) return bar;
            if (bar >= 0) {
                stats[1]++;
                const unsigned char *ts = rs + barcutlen[bar];        /* :260 */
                int32_t tg = ts < re ? trie_lookup(&ix->tag, ts, re) : -1;
                if (tg <= -3) return tg;
                if (tg >= 0) { stats[2]++; counts[(uint64_t)bar * ntags + (uint32_t)tg] += tassel ? weight : 1; }
            }
            if ((line >> 2) + 1 >= maxreads_eff) { line++; break; }   /* :272-273 */
        }
        line++;
    }
    stats[3] = lines_seen;
    return ORC_OK;
}
