/*
 * ukm_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's (shenwei356/unikmer @ 0.21.0) k-mer
 * encode / ntHash / sort / set-operation hot path.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library; the product
 * (unikmer_amd/, libunikmer_hip.so) never links, imports or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/unikmer/cmd/ unless noted).  The arithmetic that lives in
 * un-vendored third-party Go modules is restated from their published
 * algorithms and pinned by the reference's own known answers (tests/test_oracle_kat.py,
 * SURVEY.md Appendix C):
 *   - github.com/shenwei356/kmers v0.1.0          (2-bit encode / revcomp / canonical)   PINNED
 *   - github.com/will-rowe/nthash v0.4.0          (ntHash v1)                            PINNED
 *   - github.com/shenwei356/bio v0.13.3 sketches  (k-mer / hash / minimizer iterators)   PINNED
 *   - github.com/twotwotwo/sorts @bf5c1f2         (ascending sort)                       self-evident
 *   - github.com/shenwei356/bio v0.13.3 taxdump   (LCA)                  PARITY UNPINNED (no LCA
 *         value, nodes.dmp or taxid-bearing output exists anywhere in the reference tree)
 */
#ifndef UKM_ORACLE_H
#define UKM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* scan / merge modes (sort.go:484-572, util-sort.go:35-190,227-606) */
#define UKMO_PLAIN 0          /* keep every record */
#define UKMO_UNIQUE 1         /* one record per distinct code, taxid = LCA fold */
#define UKMO_REPEATED 2       /* codes seen >= 2 times, once each (sort -d / merge finalRound) */
#define UKMO_REPEATED_CHUNK 3 /* chunk protocol: every code once, repeated ones twice */
#define UKMO_SINGLETON 4      /* codes seen exactly once (count -u, count.go:424-432,475-486) */

/* ---- kmers v0.1.0 ---- */
int ukmo_encode(const uint8_t *kmer, int k, uint64_t *code); /* 0 ok, -1 illegal base, -2 bad k */
uint64_t ukmo_revcomp(uint64_t code, int k);
uint64_t ukmo_canonical(uint64_t code, int k);
void ukmo_decode(uint64_t code, int k, uint8_t *out);

/* ---- bio/sketches iterators over ONE sequence ----
 * return: number of values written; -1 = ErrShortSeq (len < k); -2 = illegal base */
int64_t ukmo_kmer_iter(const uint8_t *seq, uint64_t len, int k, int canonical, int circular,
                       uint64_t *out);
int64_t ukmo_hash_iter(const uint8_t *seq, uint64_t len, int k, int canonical, int circular,
                       uint64_t *out);
int64_t ukmo_minimizer(const uint8_t *seq, uint64_t len, int k, int w, int circular,
                       uint64_t *out_hash, uint64_t *out_pos);
/* one-shot (non rolling) ntHash of one k-mer: fwd and rev strands */
void ukmo_nthash_kmer(const uint8_t *kmer, int k, uint64_t *fwd, uint64_t *rev);
/* count.go:98 */
uint64_t ukmo_max_hash(uint64_t scale);

/* ---- multi-record drivers (count.go:285-375): concatenated bases + rec_off[n_rec+1];
 * records shorter than k are skipped (count.go:323-328); optional Scaled filter
 * (count.go:373) when max_hash != 0.  hashed=0 -> k-mer codes, hashed=1 -> ntHash. */
int64_t ukmo_count_windows(const uint8_t *bases, const uint64_t *rec_off, uint64_t n_rec, int k,
                           int hashed, int canonical, int circular, uint64_t max_hash,
                           uint64_t *out /* may be NULL to only count */);

/* ---- sorts ---- */
void ukmo_sort_u64(uint64_t *keys, uint64_t n);                    /* sortutil.Uint64s */
void ukmo_sort_pairs(uint64_t *keys, uint32_t *taxids, uint64_t n); /* by code only, stable */

/* ---- taxonomy (bio/taxdump; util.go:119-171) ---- */
typedef struct ukmo_tax ukmo_tax;
ukmo_tax *ukmo_tax_create(const uint32_t *child, const uint32_t *parent, uint64_t n,
                          const uint32_t *merged_old, const uint32_t *merged_new, uint64_t m);
void ukmo_tax_destroy(ukmo_tax *t);
uint32_t ukmo_lca(const ukmo_tax *t, uint32_t a, uint32_t b);

/* ---- scans over a sorted stream (sort.go:484-572; util-sort.go:35-190) ---- */
uint64_t ukmo_unique(const uint64_t *keys, const uint32_t *taxids /* NULL = no taxid */,
                     uint64_t n, int mode, const ukmo_tax *tax, uint64_t *out_keys,
                     uint32_t *out_taxids);

/* ---- k-way heap merge (util-sort.go:227-606) ---- */
uint64_t ukmo_merge_k(const uint64_t *const *keys, const uint32_t *const *taxids,
                      const uint64_t *lens, int nstreams, int mode, int final_round,
                      const ukmo_tax *tax, uint64_t *out_keys, uint32_t *out_taxids);

/* ---- set operations; output is always the SORTED (code, taxid) stream ---- */
#define UKMO_F_TAXID 1        /* records carry taxids */
#define UKMO_F_MIX_TAXID 2    /* inter --mix-taxid (inter.go:229-236) */
#define UKMO_F_CMP_TAXID 4    /* diff -t (diff.go:361-362,406-407) */
/* union.go:186-305 (hash map + sort of keys) */
uint64_t ukmo_union(const uint64_t *const *keys, const uint32_t *const *taxids,
                    const uint64_t *lens, int nstreams, uint32_t flags, const ukmo_tax *tax,
                    uint64_t *out_keys, uint32_t *out_taxids);
/* inter.go:188-286 (2-pointer against the running result) */
uint64_t ukmo_inter(const uint64_t *const *keys, const uint32_t *const *taxids,
                    const uint64_t *lens, int nstreams, uint32_t flags, const ukmo_tax *tax,
                    uint64_t *out_keys, uint32_t *out_taxids);
/* diff.go:341-454 single worker; sorted_flags[i]!=0 -> 2-pointer path, else map-delete path */
uint64_t ukmo_diff(const uint64_t *const *keys, const uint32_t *const *taxids,
                   const uint64_t *lens, int nstreams, const uint8_t *sorted_flags,
                   uint32_t flags, const ukmo_tax *tax, uint64_t *out_keys,
                   uint32_t *out_taxids);
/* common.go:220-344 (counting map, threshold, sort) */
uint64_t ukmo_common(const uint64_t *const *keys, const uint32_t *const *taxids,
                     const uint64_t *lens, int nstreams, uint32_t threshold, uint32_t flags,
                     const ukmo_tax *tax, uint64_t *out_keys, uint32_t *out_taxids);
/* common.go:93-105 */
uint32_t ukmo_common_threshold(uint32_t nfiles, double proportion, uint32_t number);

/* ---- timing helpers for bench.py's cpu_baseline leg: run op, return seconds ---- */
double ukmo_time_union2(const uint64_t *a, uint64_t na, const uint64_t *b, uint64_t nb,
                        uint64_t *out, uint64_t *n_out);
double ukmo_time_inter2(const uint64_t *a, uint64_t na, const uint64_t *b, uint64_t nb,
                        uint64_t *out, uint64_t *n_out);

/* all-cores sorted merge (SURVEY.md 8(d)(ii) "honest best-CPU bar"; not the reference's algorithm) */
double ukmo_time_setop2_allcores(int op, const uint64_t *a, uint64_t na, const uint64_t *b, uint64_t nb,
                                 uint64_t *out, uint64_t *n_out, int *threads_used);

#ifdef __cplusplus
}
#endif
#endif
