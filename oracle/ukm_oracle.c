/*
 * ukm_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See ukm_oracle.h.
 *
 * Restates, in plain C, what the reference's hot path computes.  File:line citations are
 * relative to /root/reference/unikmer/cmd/ ; third-party arithmetic follows SURVEY.md
 * Appendix B and is pinned by the known answers in tests/test_oracle_kat.py.
 *
 * Deliberate divergences from reference behaviour on DEGENERATE inputs (the reference
 * crashes, hangs or silently drops data there; product and oracle both implement the
 * intended set semantics instead — listed in DESIGN.md "Reference quirks"):
 *   - inter with an empty FIRST file: reference indexes mc[0] and panics (inter.go:209);
 *     here the result is empty.
 *   - diff against an empty sorted file: reference worker exits without publishing its map
 *     (diff.go:387-393) so the output becomes empty / later files are skipped; here an empty
 *     file subtracts nothing.  diff with one input: reference writes an empty file
 *     (diff.go:485-523); here the first stream is returned.
 *   - diff mixing unsorted-then-sorted files in one worker: reference loses the unsorted
 *     file's deletions (diff.go:379-453 rebuilds from mc1); here every file is subtracted.
 *   - merge -u without taxids: reference re-writes the value of the last failed read
 *     (util-sort.go:496-500); here the correct distinct set is produced.
 *   - sort/merge -u with taxids on EMPTY input writes one (0xFFFFFFFFFFFFFFFF, 0) record
 *     (sort.go:485-507); here the output is empty.
 * Reference quirk that IS reproduced: inter with an empty LATER file stops and returns the
 * current running result un-intersected (inter.go:211-217).
 */
#define _POSIX_C_SOURCE 200809L
#include "ukm_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------ */
/* kmers v0.1.0.  A=0 C=1 G=2 T/U=3; degenerate IUPAC letters collapse to */
/* their first alphabetical base; anything else is an illegal base.  Case-insensitive.   */
/* ------------------------------------------------------------------------------------ */
static int base2bit(uint8_t c) {
    switch (c) {
    case 'A': case 'a': case 'N': case 'n': case 'M': case 'm': case 'V': case 'v':
    case 'H': case 'h': case 'R': case 'r': case 'D': case 'd': case 'W': case 'w':
        return 0;
    case 'C': case 'c': case 'S': case 's': case 'B': case 'b': case 'Y': case 'y':
        return 1;
    case 'G': case 'g': case 'K': case 'k':
        return 2;
    case 'T': case 't': case 'U': case 'u':
        return 3;
    default:
        return 4;
    }
}

int ukmo_encode(const uint8_t *kmer, int k, uint64_t *code) {
    if (k <= 0 || k > 32) return -2;
    uint64_t c = 0;
    for (int i = 0; i < k; i++) {
        int b = base2bit(kmer[i]);
        if (b > 3) return -1;
        c = (c << 2) | (uint64_t)b; /* first base ends up most significant */
    }
    *code = c;
    return 0;
}

uint64_t ukmo_revcomp(uint64_t code, int k) {
    /* complement = XOR with all-ones over 2k bits; then reverse the 2-bit groups */
    uint64_t c = ~code;
    uint64_t r = 0;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (c & 3);
        c >>= 2;
    }
    return r;
}

uint64_t ukmo_canonical(uint64_t code, int k) {
    uint64_t rc = ukmo_revcomp(code, k);
    return rc < code ? rc : code;
}

void ukmo_decode(uint64_t code, int k, uint8_t *out) {
    static const char bit2base[4] = {'A', 'C', 'G', 'T'};
    for (int i = k - 1; i >= 0; i--) {
        out[i] = (uint8_t)bit2base[code & 3];
        code >>= 2;
    }
}

/* bio/sketches KmerIterator (count.go:321,363): every window i = 0..len-k in order; circular
 * extends the sequence by its first k-1 bases. */
int64_t ukmo_kmer_iter(const uint8_t *seq, uint64_t len, int k, int canonical, int circular,
                       uint64_t *out) {
    if (k <= 0 || k > 32) return -2;
    if (len < (uint64_t)k) return -1; /* ErrShortSeq */
    uint64_t total = circular ? len + (uint64_t)k - 1 : len;
    uint64_t mask = (k == 32) ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
    uint64_t fwd = 0, rev = 0;
    int64_t n = 0;
    for (uint64_t p = 0; p < total; p++) {
        int b = base2bit(seq[p < len ? p : p - len]);
        if (b > 3) return -2;
        fwd = ((fwd << 2) | (uint64_t)b) & mask;
        rev = (rev >> 2) | ((uint64_t)(3 - b) << (2 * (k - 1)));
        if (p + 1 >= (uint64_t)k) {
            uint64_t c = fwd;
            if (canonical && rev < fwd) c = rev;
            if (out) out[n] = c;
            n++;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* ntHash v1 (will-rowe/nthash v0.4.0).  Seeds = bcgsc ntHash v1.          */
/* ------------------------------------------------------------------------------------ */
#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL

static inline uint64_t rol64(uint64_t x, unsigned s) {
    s &= 63;
    return s ? (x << s) | (x >> (64 - s)) : x;
}
static inline uint64_t ror64(uint64_t x, unsigned s) {
    s &= 63;
    return s ? (x >> s) | (x << (64 - s)) : x;
}
static inline uint64_t nt_seed(uint8_t c) {
    switch (c) {
    case 'A': case 'a': return SEED_A;
    case 'C': case 'c': return SEED_C;
    case 'G': case 'g': return SEED_G;
    case 'T': case 't': case 'U': case 'u': return SEED_T;
    default: return 0; /* N and everything else */
    }
}
static inline uint64_t nt_cseed(uint8_t c) { /* seed of the complement base */
    switch (c) {
    case 'A': case 'a': return SEED_T;
    case 'C': case 'c': return SEED_G;
    case 'G': case 'g': return SEED_C;
    case 'T': case 't': case 'U': case 'u': return SEED_A;
    default: return 0;
    }
}

void ukmo_nthash_kmer(const uint8_t *kmer, int k, uint64_t *fwd, uint64_t *rev) {
    uint64_t f = 0, r = 0;
    for (int i = 0; i < k; i++) {
        f ^= rol64(nt_seed(kmer[i]), (unsigned)(k - 1 - i));
        r ^= rol64(nt_cseed(kmer[i]), (unsigned)i);
    }
    *fwd = f;
    *rev = r;
}

/* bio/sketches HashIterator -> nthash.Hasher.Next (count.go:319,361; dump.go:253-260):
 * rolling update  fwd' = rol(fwd,1) ^ rol(seed[out],k) ^ seed[in]
 *                 rev' = ror(rev,1) ^ ror(cseed[out],1) ^ rol(cseed[in],k-1)              */
int64_t ukmo_hash_iter(const uint8_t *seq, uint64_t len, int k, int canonical, int circular,
                       uint64_t *out) {
    if (k <= 0 || k > 64) return -2;
    if (len < (uint64_t)k) return -1;
    uint64_t total = circular ? len + (uint64_t)k - 1 : len;
    uint64_t fwd = 0, rev = 0;
    int64_t n = 0;
#define SEQ_AT(p) (seq[(p) < len ? (p) : (p)-len])
    for (int i = 0; i < k; i++) {
        uint8_t c = SEQ_AT((uint64_t)i);
        fwd ^= rol64(nt_seed(c), (unsigned)(k - 1 - i));
        rev ^= rol64(nt_cseed(c), (unsigned)i);
    }
    for (uint64_t p = (uint64_t)k - 1;; p++) {
        uint64_t h = fwd;
        if (canonical && rev < fwd) h = rev;
        if (out) out[n] = h;
        n++;
        if (p + 1 >= total) break;
        uint8_t cin = SEQ_AT(p + 1), cout = SEQ_AT(p + 1 - (uint64_t)k);
        fwd = rol64(fwd, 1) ^ rol64(nt_seed(cout), (unsigned)k) ^ nt_seed(cin);
        rev = ror64(rev, 1) ^ ror64(nt_cseed(cout), 1) ^ rol64(nt_cseed(cin), (unsigned)(k - 1));
    }
#undef SEQ_AT
    return n;
}

/* count.go:98 : maxHash := uint64(float64(^uint64(0)) / float64(scale)) */
uint64_t ukmo_max_hash(uint64_t scale) {
    double d = (double)(~(uint64_t)0) / (double)scale;
    if (d >= 18446744073709551616.0) return ~(uint64_t)0; /* scale==1: Go saturates on amd64? unused: scaled only when scale>1 */
    return (uint64_t)d;
}

/* bio/sketches MinimizerSketch: canonical ntHash of every window; for each
 * group of w consecutive windows the LEFTMOST minimum; emitted when the arg-min position
 * differs from the previous group's. */
int64_t ukmo_minimizer(const uint8_t *seq, uint64_t len, int k, int w, int circular,
                       uint64_t *out_hash, uint64_t *out_pos) {
    if (w <= 0) return -2;
    uint64_t total = circular ? len + (uint64_t)k - 1 : len;
    if (len < (uint64_t)k || total < (uint64_t)k + (uint64_t)w - 1) return -1;
    uint64_t nwin = total - (uint64_t)k + 1;
    uint64_t *h = (uint64_t *)malloc(nwin * sizeof(uint64_t));
    if (!h) return -3;
    int64_t r = ukmo_hash_iter(seq, len, k, 1, circular, h);
    if (r < 0) { free(h); return r; }
    int64_t n = 0;
    uint64_t prev = ~(uint64_t)0;
    for (uint64_t g = 0; g + (uint64_t)w <= nwin; g++) {
        uint64_t arg = g;
        for (uint64_t j = g + 1; j < g + (uint64_t)w; j++)
            if (h[j] < h[arg]) arg = j;
        if (arg != prev) {
            if (out_hash) out_hash[n] = h[arg];
            if (out_pos) out_pos[n] = arg;
            n++;
            prev = arg;
        }
    }
    free(h);
    return n;
}

/* count.go:285-375 driver over many records */
int64_t ukmo_count_windows(const uint8_t *bases, const uint64_t *rec_off, uint64_t n_rec, int k,
                           int hashed, int canonical, int circular, uint64_t max_hash,
                           uint64_t *out) {
    int64_t n = 0;
    uint64_t *tmp = NULL;
    uint64_t tmp_cap = 0;
    for (uint64_t r = 0; r < n_rec; r++) {
        const uint8_t *s = bases + rec_off[r];
        uint64_t len = rec_off[r + 1] - rec_off[r];
        if (len < (uint64_t)k) continue; /* ErrShortSeq -> skip record (count.go:323-328) */
        uint64_t nw = circular ? len : len - (uint64_t)k + 1;
        if (max_hash == 0 && out) {
            int64_t m = hashed ? ukmo_hash_iter(s, len, k, canonical, circular, out + n)
                               : ukmo_kmer_iter(s, len, k, canonical, circular, out + n);
            if (m < 0) { free(tmp); return m; }
            n += m;
            continue;
        }
        if (nw > tmp_cap) {
            free(tmp);
            tmp_cap = nw;
            tmp = (uint64_t *)malloc(tmp_cap * sizeof(uint64_t));
            if (!tmp) return -3;
        }
        int64_t m = hashed ? ukmo_hash_iter(s, len, k, canonical, circular, tmp)
                           : ukmo_kmer_iter(s, len, k, canonical, circular, tmp);
        if (m < 0) { free(tmp); return m; }
        for (int64_t i = 0; i < m; i++) {
            if (max_hash && tmp[i] > max_hash) continue; /* count.go:373 */
            if (out) out[n] = tmp[i];
            n++;
        }
    }
    free(tmp);
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* sorts: sortutil.Uint64s = ascending; CodeTaxidSlice sorts by Code only (kmers.go:44).  */
/* LSD radix, 8-bit digits, constant digits skipped; stable for pairs.                    */
/* ------------------------------------------------------------------------------------ */
void ukmo_sort_u64(uint64_t *keys, uint64_t n) {
    if (n < 2) return;
    uint64_t *tmp = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t *src = keys, *dst = tmp;
    uint64_t hist[8][256]; /* (on the stack: the test suite calls the oracle from several threads) */
    memset(hist, 0, sizeof(hist));
    for (uint64_t i = 0; i < n; i++) {
        uint64_t v = keys[i];
        for (int p = 0; p < 8; p++) hist[p][(v >> (8 * p)) & 255]++;
    }
    for (int p = 0; p < 8; p++) {
        uint64_t *h = hist[p];
        int constant = 0;
        for (int d = 0; d < 256; d++)
            if (h[d] == n) constant = 1;
        if (constant) continue;
        uint64_t sum = 0;
        for (int d = 0; d < 256; d++) { uint64_t c = h[d]; h[d] = sum; sum += c; }
        for (uint64_t i = 0; i < n; i++) {
            uint64_t v = src[i];
            dst[h[(v >> (8 * p)) & 255]++] = v;
        }
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != keys) memcpy(keys, src, n * sizeof(uint64_t));
    free(tmp);
}

void ukmo_sort_pairs(uint64_t *keys, uint32_t *taxids, uint64_t n) {
    if (n < 2) return;
    uint64_t *tk = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint32_t *tv = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint64_t *sk = keys, *dk = tk;
    uint32_t *sv = taxids, *dv = tv;
    uint64_t hist[256];
    for (int p = 0; p < 8; p++) {
        memset(hist, 0, sizeof(hist));
        for (uint64_t i = 0; i < n; i++) hist[(sk[i] >> (8 * p)) & 255]++;
        int constant = 0;
        for (int d = 0; d < 256; d++)
            if (hist[d] == n) constant = 1;
        if (constant) continue;
        uint64_t sum = 0;
        for (int d = 0; d < 256; d++) { uint64_t c = hist[d]; hist[d] = sum; sum += c; }
        for (uint64_t i = 0; i < n; i++) {
            uint64_t pos = hist[(sk[i] >> (8 * p)) & 255]++;
            dk[pos] = sk[i];
            dv[pos] = sv[i];
        }
        uint64_t *t = sk; sk = dk; dk = t;
        uint32_t *u = sv; sv = dv; dv = u;
    }
    if (sk != keys) {
        memcpy(keys, sk, n * sizeof(uint64_t));
        memcpy(taxids, sv, n * sizeof(uint32_t));
    }
    free(tk);
    free(tv);
}

/* ------------------------------------------------------------------------------------ */
/* taxonomy (bio/taxdump, SURVEY.md B5 — PARITY UNPINNED).  Contract of this build:       */
/*   LCA(0,x)=LCA(x,0)=0; LCA(x,x)=x; merged ids are remapped; an id absent from          */
/*   nodes.dmp (after remap) -> 0; otherwise the lowest common ancestor in the tree whose */
/*   root is the node with parent == child.  Algorithm here: ancestor-list walk (the      */
/*   product uses a depth-equalising climb; the two are independent implementations).     */
/* ------------------------------------------------------------------------------------ */
struct ukmo_tax {
    uint32_t *parent; /* dense, 0 = absent */
    uint32_t *merged; /* dense, 0 = not merged */
    uint64_t size;
};

ukmo_tax *ukmo_tax_create(const uint32_t *child, const uint32_t *parent, uint64_t n,
                          const uint32_t *merged_old, const uint32_t *merged_new, uint64_t m) {
    uint32_t mx = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (child[i] > mx) mx = child[i];
        if (parent[i] > mx) mx = parent[i];
    }
    for (uint64_t i = 0; i < m; i++) {
        if (merged_old[i] > mx) mx = merged_old[i];
        if (merged_new[i] > mx) mx = merged_new[i];
    }
    ukmo_tax *t = (ukmo_tax *)calloc(1, sizeof(ukmo_tax));
    t->size = (uint64_t)mx + 1;
    t->parent = (uint32_t *)calloc(t->size, sizeof(uint32_t));
    t->merged = (uint32_t *)calloc(t->size, sizeof(uint32_t));
    for (uint64_t i = 0; i < n; i++) t->parent[child[i]] = parent[i];
    /* a parent id that never appears as a child acts as a root of its own */
    for (uint64_t i = 0; i < n; i++)
        if (t->parent[parent[i]] == 0) t->parent[parent[i]] = parent[i];
    for (uint64_t i = 0; i < m; i++) t->merged[merged_old[i]] = merged_new[i];
    return t;
}

void ukmo_tax_destroy(ukmo_tax *t) {
    if (!t) return;
    free(t->parent);
    free(t->merged);
    free(t);
}

static uint32_t tax_resolve(const ukmo_tax *t, uint32_t a) {
    if (a >= t->size) return 0;
    if (t->parent[a]) return a;
    uint32_t b = t->merged[a];
    if (b && b < t->size && t->parent[b]) return b;
    return 0;
}

uint32_t ukmo_lca(const ukmo_tax *t, uint32_t a, uint32_t b) {
    if (a == 0 || b == 0) return 0;
    if (a == b) return a;
    a = tax_resolve(t, a);
    b = tax_resolve(t, b);
    if (a == 0 || b == 0) return 0;
    if (a == b) return a;
    uint32_t line[256];
    int nl = 0;
    uint32_t c = a;
    for (;;) {
        if (nl < 256) line[nl++] = c;
        uint32_t p = t->parent[c];
        if (p == c || p == 0) break;
        c = p;
    }
    c = b;
    for (;;) {
        for (int i = 0; i < nl; i++)
            if (line[i] == c) return c;
        uint32_t p = t->parent[c];
        if (p == c || p == 0) break;
        c = p;
    }
    return 0; /* different trees */
}

/* ------------------------------------------------------------------------------------ */
/* scans over a sorted stream                                                             */
/* ------------------------------------------------------------------------------------ */
uint64_t ukmo_unique(const uint64_t *keys, const uint32_t *taxids, uint64_t n, int mode,
                     const ukmo_tax *tax, uint64_t *out_keys, uint32_t *out_taxids) {
    uint64_t no = 0;
#define EMIT(code, tx) do { out_keys[no] = (code); if (taxids) out_taxids[no] = (tx); no++; } while (0)
    if (mode == UKMO_PLAIN) { /* sort.go:566-572 */
        for (uint64_t i = 0; i < n; i++) EMIT(keys[i], taxids ? taxids[i] : 0);
        return no;
    }
    /* run-wise restatement of sort.go:484-565 / util-sort.go:54-179: a run of equal codes has
     * `count` members and lca = left fold LCA(taxid_i, lca) (sort.go:491) */
    uint64_t i = 0;
    while (i < n) {
        uint64_t code = keys[i];
        uint32_t lca = taxids ? taxids[i] : 0;
        uint64_t count = 1;
        uint64_t j = i + 1;
        while (j < n && keys[j] == code) {
            if (taxids) lca = ukmo_lca(tax, taxids[j], lca);
            count++;
            j++;
        }
        if (mode == UKMO_UNIQUE) {
            EMIT(code, lca);
        } else if (mode == UKMO_REPEATED) { /* sort.go:508-532,551-565 */
            if (count > 1) EMIT(code, lca);
        } else if (mode == UKMO_SINGLETON) { /* count.go:475-486: marks[code] == false */
            if (count == 1) EMIT(code, lca);
        } else { /* UKMO_REPEATED_CHUNK, util-sort.go:61-91,145-179 */
            EMIT(code, lca);
            if (count > 1) EMIT(code, lca);
        }
        i = j;
    }
#undef EMIT
    return no;
}

/* ------------------------------------------------------------------------------------ */
/* k-way heap merge (util-sort.go:196-606).  container/heap ties are unspecified in Go;   */
/* here ties pop in stream-index order, which equals a stable sort of the concatenation.  */
/* ------------------------------------------------------------------------------------ */
typedef struct { uint64_t code; uint32_t taxid; int idx; } hentry;

static int hless(const hentry *a, const hentry *b) {
    if (a->code != b->code) return a->code < b->code;
    return a->idx < b->idx;
}
static void hpush(hentry *h, int *n, hentry e) {
    int i = (*n)++;
    h[i] = e;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!hless(&h[i], &h[p])) break;
        hentry t = h[i]; h[i] = h[p]; h[p] = t;
        i = p;
    }
}
static hentry hpop(hentry *h, int *n) {
    hentry top = h[0];
    h[0] = h[--(*n)];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && hless(&h[l], &h[m])) m = l;
        if (r < *n && hless(&h[r], &h[m])) m = r;
        if (m == i) break;
        hentry t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
    return top;
}

uint64_t ukmo_merge_k(const uint64_t *const *keys, const uint32_t *const *taxids,
                      const uint64_t *lens, int nstreams, int mode, int final_round,
                      const ukmo_tax *tax, uint64_t *out_keys, uint32_t *out_taxids) {
    hentry *heap = (hentry *)malloc(sizeof(hentry) * (size_t)(nstreams > 0 ? nstreams : 1));
    uint64_t *cur = (uint64_t *)calloc((size_t)(nstreams > 0 ? nstreams : 1), sizeof(uint64_t));
    int hn = 0;
    int has_tax = taxids != NULL;
    for (int s = 0; s < nstreams; s++)
        if (lens[s] > 0) {
            hentry e = {keys[s][0], (has_tax && taxids[s]) ? taxids[s][0] : 0, s};
            hpush(heap, &hn, e);
            cur[s] = 1;
        }
    uint64_t no = 0;
    int have = 0;
    uint64_t last = 0, count = 0;
    uint32_t lca = 0;
#define EMIT(code, tx) do { out_keys[no] = (code); if (has_tax) out_taxids[no] = (tx); no++; } while (0)
#define FLUSH_RUN() do { \
        if (mode == UKMO_UNIQUE) EMIT(last, lca); \
        else { /* repeated: util-sort.go:377-388,519-530 */ \
            if (!final_round) EMIT(last, lca); \
            if (count > 1) EMIT(last, lca); \
        } } while (0)
    while (hn > 0) {
        hentry e = hpop(heap, &hn);
        if (mode == UKMO_PLAIN) {
            EMIT(e.code, e.taxid);
        } else if (have && e.code == last) {
            if (has_tax) lca = ukmo_lca(tax, e.taxid, lca); /* util-sort.go:325,374 */
            count++;
        } else {
            if (have) FLUSH_RUN();
            have = 1;
            last = e.code;
            lca = e.taxid;
            count = 1;
        }
        int s = e.idx;
        if (cur[s] < lens[s]) {
            hentry ne = {keys[s][cur[s]], (has_tax && taxids[s]) ? taxids[s][cur[s]] : 0, s};
            cur[s]++;
            hpush(heap, &hn, ne);
        }
    }
    if (mode != UKMO_PLAIN && have) FLUSH_RUN();
#undef FLUSH_RUN
#undef EMIT
    free(heap);
    free(cur);
    return no;
}

/* ------------------------------------------------------------------------------------ */
/* open-addressing hash map uint64 -> (uint32 taxid, uint32 count), stands in for Go's    */
/* map[uint64]struct{} / map[uint64]uint32 / map[uint64]uint16 (union.go:75-77,           */
/* common.go:111).  Initial size mapInitSize = 1<<20 (util.go:43).                        */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t *keys;
    uint32_t *vals;
    uint32_t *cnts;
    uint8_t *used;
    uint64_t cap, n;
} map64;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
static void map_init(map64 *m, uint64_t cap) {
    uint64_t c = 1 << 21;
    while (c < cap * 2) c <<= 1;
    m->cap = c;
    m->n = 0;
    m->keys = (uint64_t *)malloc(c * sizeof(uint64_t));
    m->vals = (uint32_t *)malloc(c * sizeof(uint32_t));
    m->cnts = (uint32_t *)malloc(c * sizeof(uint32_t));
    m->used = (uint8_t *)calloc(c, 1);
}
static void map_free(map64 *m) {
    free(m->keys); free(m->vals); free(m->cnts); free(m->used);
}
static uint64_t map_slot(const map64 *m, uint64_t key) {
    uint64_t i = mix64(key) & (m->cap - 1);
    while (m->used[i] == 1 && m->keys[i] != key) i = (i + 1) & (m->cap - 1);
    return i;
}
static void map_grow(map64 *m) {
    map64 b;
    b.cap = m->cap * 2;
    b.n = m->n;
    b.keys = (uint64_t *)malloc(b.cap * sizeof(uint64_t));
    b.vals = (uint32_t *)malloc(b.cap * sizeof(uint32_t));
    b.cnts = (uint32_t *)malloc(b.cap * sizeof(uint32_t));
    b.used = (uint8_t *)calloc(b.cap, 1);
    for (uint64_t i = 0; i < m->cap; i++)
        if (m->used[i] == 1) {
            uint64_t j = map_slot(&b, m->keys[i]);
            b.used[j] = 1; b.keys[j] = m->keys[i]; b.vals[j] = m->vals[i]; b.cnts[j] = m->cnts[i];
        }
    map_free(m);
    *m = b;
}
/* returns slot; *found tells whether key existed */
static uint64_t map_get_or_insert(map64 *m, uint64_t key, int *found) {
    if ((m->n + 1) * 2 > m->cap) map_grow(m);
    uint64_t i = map_slot(m, key);
    if (m->used[i] == 1) { *found = 1; return i; }
    m->used[i] = 1; m->keys[i] = key; m->vals[i] = 0; m->cnts[i] = 0; m->n++;
    *found = 0;
    return i;
}

/* a stream without TaxId information (taxids[s] == NULL) among streams that carry it: its records read as taxid 0
 * (the reader returns the file's global taxid, 0 when there is none: SURVEY.md Appendix A notation) */
#define UKMO_TX(s, i) ((taxids && taxids[s]) ? taxids[s][i] : 0u)

/* union.go:186-305 */
uint64_t ukmo_union(const uint64_t *const *keys, const uint32_t *const *taxids,
                    const uint64_t *lens, int nstreams, uint32_t flags, const ukmo_tax *tax,
                    uint64_t *out_keys, uint32_t *out_taxids) {
    int has_tax = (flags & UKMO_F_TAXID) != 0;
    map64 m;
    map_init(&m, 1 << 20);
    for (int s = 0; s < nstreams; s++)
        for (uint64_t i = 0; i < lens[s]; i++) {
            int found;
            uint64_t slot = map_get_or_insert(&m, keys[s][i], &found);
            if (has_tax) { /* union.go:195-201 */
                if (!found) m.vals[slot] = UKMO_TX(s, i);
                else m.vals[slot] = ukmo_lca(tax, m.vals[slot], UKMO_TX(s, i));
            }
        }
    uint64_t n = 0;
    for (uint64_t i = 0; i < m.cap; i++)
        if (m.used[i] == 1) out_keys[n++] = m.keys[i];
    ukmo_sort_u64(out_keys, n); /* union.go:274,295 */
    if (has_tax)
        for (uint64_t i = 0; i < n; i++) out_taxids[i] = m.vals[map_slot(&m, out_keys[i])];
    map_free(&m);
    return n;
}

/* inter.go:188-286 */
uint64_t ukmo_inter(const uint64_t *const *keys, const uint32_t *const *taxids,
                    const uint64_t *lens, int nstreams, uint32_t flags, const ukmo_tax *tax,
                    uint64_t *out_keys, uint32_t *out_taxids) {
    int has_tax = (flags & UKMO_F_TAXID) != 0;
    int mix = (flags & UKMO_F_MIX_TAXID) != 0;
    if (nstreams <= 0) return 0;
    uint64_t n = lens[0];
    uint64_t *mc = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    uint32_t *mt = (uint32_t *)calloc((n ? n : 1), sizeof(uint32_t));
    uint8_t *hit = (uint8_t *)calloc((n ? n : 1), 1);
    memcpy(mc, keys[0], n * sizeof(uint64_t)); /* inter.go:189-200 */
    if ((has_tax || mix) && taxids && taxids[0]) memcpy(mt, taxids[0], n * sizeof(uint32_t));
    for (int s = 1; s < nstreams && n > 0; s++) {
        if (lens[s] == 0) break; /* inter.go:211-217: flagBreak, running result kept as is */
        const uint64_t *q = keys[s];
        const uint32_t *qt = (taxids && taxids[s]) ? taxids[s] : NULL;
        uint64_t ii = 0, jj = 0;
        memset(hit, 0, n);
        while (ii < n && jj < lens[s]) { /* inter.go:220-267 */
            if (mc[ii] < q[jj]) {
                ii++;
            } else if (mc[ii] == q[jj]) {
                uint32_t t = qt ? qt[jj] : 0;
                if (mix) {
                    if (mt[ii] == 0) mt[ii] = t;
                    else if (t == 0) { /* keep */ }
                    else mt[ii] = ukmo_lca(tax, mt[ii], t);
                } else if (has_tax) {
                    mt[ii] = ukmo_lca(tax, mt[ii], t);
                }
                hit[ii] = 1;
                ii++;
                jj++;
            } else {
                jj++;
            }
        }
        uint64_t w = 0; /* inter.go:269-278 */
        for (uint64_t i = 0; i < n; i++)
            if (hit[i]) { mc[w] = mc[i]; mt[w] = mt[i]; w++; }
        n = w;
    }
    memcpy(out_keys, mc, n * sizeof(uint64_t));
    if ((has_tax || mix) && out_taxids) memcpy(out_taxids, mt, n * sizeof(uint32_t));
    free(mc); free(mt); free(hit);
    return n;
}

/* diff.go:379-454 (single worker), then the survivor map -> sorted keys (diff.go:574-594) */
uint64_t ukmo_diff(const uint64_t *const *keys, const uint32_t *const *taxids,
                   const uint64_t *lens, int nstreams, const uint8_t *sorted_flags,
                   uint32_t flags, const ukmo_tax *tax, uint64_t *out_keys,
                   uint32_t *out_taxids) {
    int has_tax = (flags & UKMO_F_TAXID) != 0;
    int cmp = (flags & UKMO_F_CMP_TAXID) != 0;
    if (nstreams <= 0) return 0;
    uint64_t n = lens[0];
    uint64_t *mc = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    uint32_t *mt = (uint32_t *)calloc((n ? n : 1), sizeof(uint32_t));
    memcpy(mc, keys[0], n * sizeof(uint64_t));
    if (has_tax && taxids && taxids[0]) memcpy(mt, taxids[0], n * sizeof(uint32_t));
    for (int s = 1; s < nstreams && n > 0; s++) {
        uint64_t nq = lens[s];
        if (nq == 0) continue;
        uint64_t *qk = NULL;
        uint32_t *qv = NULL;
        const uint64_t *q = keys[s];
        const uint32_t *qt = (taxids && taxids[s]) ? taxids[s] : NULL;
        if (sorted_flags && !sorted_flags[s]) { /* unsorted file: same set semantics */
            qk = (uint64_t *)malloc(nq * sizeof(uint64_t));
            qv = (uint32_t *)calloc(nq, sizeof(uint32_t));
            memcpy(qk, q, nq * sizeof(uint64_t));
            if (qt) memcpy(qv, qt, nq * sizeof(uint32_t));
            ukmo_sort_pairs(qk, qv, nq);
            q = qk;
            qt = qv;
        }
        uint64_t ii = 0, jj = 0, w = 0;
        while (ii < n && jj < nq) { /* diff.go:395-434 */
            if (mc[ii] < q[jj]) {
                mc[w] = mc[ii]; mt[w] = mt[ii]; w++;
                ii++;
            } else if (mc[ii] == q[jj]) {
                uint32_t qtaxid = mt[ii], taxid = qt ? qt[jj] : 0;
                if (cmp && (qtaxid == taxid || ukmo_lca(tax, taxid, qtaxid) == qtaxid)) {
                    mc[w] = mc[ii]; mt[w] = mt[ii]; w++;
                }
                ii++;
                jj++;
            } else {
                jj++;
            }
        }
        for (; ii < n; ii++) { mc[w] = mc[ii]; mt[w] = mt[ii]; w++; } /* diff.go:435 */
        n = w;
        free(qk);
        free(qv);
    }
    /* survivor map (diff.go:449-453) collapses duplicate codes; sorted output (diff.go:587) */
    uint64_t w = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (w > 0 && out_keys[w - 1] == mc[i]) {
            if (has_tax && out_taxids) out_taxids[w - 1] = mt[i]; /* later map write wins */
            continue;
        }
        out_keys[w] = mc[i];
        if (has_tax && out_taxids) out_taxids[w] = mt[i];
        w++;
    }
    free(mc); free(mt);
    return w;
}

/* common.go:93-105 */
uint32_t ukmo_common_threshold(uint32_t nfiles, double proportion, uint32_t number) {
    if (number == 0) return (uint32_t)(uint16_t)((double)nfiles * proportion);
    return (uint32_t)(uint16_t)number;
}

/* common.go:220-344 */
uint64_t ukmo_common(const uint64_t *const *keys, const uint32_t *const *taxids,
                     const uint64_t *lens, int nstreams, uint32_t threshold, uint32_t flags,
                     const ukmo_tax *tax, uint64_t *out_keys, uint32_t *out_taxids) {
    int has_tax = (flags & UKMO_F_TAXID) != 0;
    map64 m;
    map_init(&m, 1 << 20);
    for (int s = 0; s < nstreams; s++)
        for (uint64_t i = 0; i < lens[s]; i++) {
            int found;
            uint64_t slot = map_get_or_insert(&m, keys[s][i], &found);
            if (s == 0) { /* common.go:232,244: first file sets count = 1, taxid overwritten */
                m.cnts[slot] = 1;
                if (has_tax) m.vals[slot] = UKMO_TX(s, i);
            } else {
                if (has_tax) { /* common.go:262-266 */
                    if (!found) m.vals[slot] = UKMO_TX(s, i);
                    else m.vals[slot] = ukmo_lca(tax, m.vals[slot], UKMO_TX(s, i));
                }
                m.cnts[slot] = (uint16_t)(m.cnts[slot] + 1); /* uint16 counts (common.go:111) */
            }
        }
    uint64_t n = 0;
    for (uint64_t i = 0; i < m.cap; i++)
        if (m.used[i] == 1 && m.cnts[i] >= threshold) out_keys[n++] = m.keys[i]; /* :331-335 */
    ukmo_sort_u64(out_keys, n); /* common.go:344 */
    if (has_tax)
        for (uint64_t i = 0; i < n; i++) out_taxids[i] = m.vals[map_slot(&m, out_keys[i])];
    map_free(&m);
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* cpu_baseline timing (bench.py): the reference algorithms, single thread.               */
/* ------------------------------------------------------------------------------------ */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double ukmo_time_union2(const uint64_t *a, uint64_t na, const uint64_t *b, uint64_t nb,
                        uint64_t *out, uint64_t *n_out) {
    const uint64_t *ks[2] = {a, b};
    uint64_t ls[2] = {na, nb};
    double t0 = now_s();
    *n_out = ukmo_union(ks, NULL, ls, 2, 0, NULL, out, NULL);
    return now_s() - t0;
}

double ukmo_time_inter2(const uint64_t *a, uint64_t na, const uint64_t *b, uint64_t nb,
                        uint64_t *out, uint64_t *n_out) {
    const uint64_t *ks[2] = {a, b};
    uint64_t ls[2] = {na, nb};
    double t0 = now_s();
    *n_out = ukmo_inter(ks, NULL, ls, 2, 0, NULL, out, NULL);
    return now_s() - t0;
}


/* ---- all-cores CPU bar: NOT the reference's algorithm (its set-op loops are
 * single-threaded, union.go:186-208 / inter.go:205-267) but the honest best a CPU does on the same
 * sorted inputs: value-range partitioned 2-pointer merges on every core, two passes (count, write).
 * op: 0 union, 1 inter, 2 diff.  Inputs strictly increasing.  Returns seconds. */
#include <omp.h>
static uint64_t lower_bound_u64(const uint64_t *a, uint64_t n, uint64_t x) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (a[m] < x) lo = m + 1; else hi = m; }
    return lo;
}
static uint64_t merge_range(int op, const uint64_t *a, uint64_t i, uint64_t ie, const uint64_t *b, uint64_t j,
                            uint64_t je, uint64_t *out) {
    uint64_t n = 0;
    while (i < ie && j < je) {
        uint64_t x = a[i], y = b[j];
        if (x < y) { if (op != 1) { if (out) out[n] = x; n++; } i++; }
        else if (y < x) { if (op == 0) { if (out) out[n] = y; n++; } j++; }
        else { if (op != 2) { if (out) out[n] = x; n++; } i++; j++; }
    }
    if (op != 1) for (; i < ie; i++) { if (out) out[n] = a[i]; n++; }
    if (op == 0) for (; j < je; j++) { if (out) out[n] = b[j]; n++; }
    return n;
}
double ukmo_time_setop2_allcores(int op, const uint64_t *a, uint64_t na, const uint64_t *b, uint64_t nb,
                                 uint64_t *out, uint64_t *n_out, int *threads_used) {
    int T = omp_get_max_threads();
    int P = T * 8; /* more parts than threads: load balance */
    if ((uint64_t)P > na + 1) P = (int)(na + 1);
    if (P < 1) P = 1;
    uint64_t *ai = (uint64_t *)malloc((size_t)(P + 1) * sizeof(uint64_t));
    uint64_t *bi = (uint64_t *)malloc((size_t)(P + 1) * sizeof(uint64_t));
    uint64_t *cnt = (uint64_t *)malloc((size_t)(P + 1) * sizeof(uint64_t));
    double t0 = now_s();
    for (int p = 0; p <= P; p++) {
        ai[p] = (p == P) ? na : na / (uint64_t)P * (uint64_t)p;
        bi[p] = (p == P) ? nb : (p == 0 ? 0 : lower_bound_u64(b, nb, a[ai[p]]));
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int p = 0; p < P; p++) cnt[p] = merge_range(op, a, ai[p], ai[p + 1], b, bi[p], bi[p + 1], NULL);
    uint64_t tot = 0;
    for (int p = 0; p < P; p++) { uint64_t c = cnt[p]; cnt[p] = tot; tot += c; }
#pragma omp parallel for schedule(dynamic, 1)
    for (int p = 0; p < P; p++) merge_range(op, a, ai[p], ai[p + 1], b, bi[p], bi[p + 1], out + cnt[p]);
    double dt = now_s() - t0;
    *n_out = tot;
    if (threads_used) *threads_used = T;
    free(ai); free(bi); free(cnt);
    return dt;
}
