/*
 * unikmer_hip.h — C ABI of libunikmer_hip.so, the MI355X (gfx950) implementation of the
 * k-mer encode / ntHash / sort / set-operation hot path of shenwei356/unikmer v0.21.0.
 *
 * The reference has no FFI; the seam is the Go package API of its third-party modules plus
 * a few in-tree loops.  Each entry point below names the reference
 * interface it replaces (paths relative to /root/reference/unikmer/cmd/).  INTEGRATION.md
 * shows the cgo binding a maintainer would add on the Go side.
 *
 * Conventions
 *  - Every call returns int: 0 = UKM_OK, negative = error class; the message is available
 *    through ukm_last_error() (thread-local).  Nothing here ever calls exit()/abort()
 *    (the reference's checkError -> os.Exit(-1), util-cli.go:39-44, stays on the Go side).
 *  - Array arguments may be HOST or DEVICE pointers, independently per argument
 *    (hipPointerGetAttributes decides).  Device pointers are used in place; host arrays are
 *    staged through the context's device workspace.  Scalar outputs (n_out) are host
 *    pointers.  All buffers are caller-owned; nothing is retained after the call returns
 *    (cgo pointer-passing rules), except the taxonomy which is copied into the context.
 *  - Outputs are caller-allocated with capacity `out_cap` (elements); upper bounds:
 *    union <= sum(n), inter <= n[0], diff <= n[0], common <= sum(n), unique <= n (2n for
 *    UKM_REPEATED_CHUNK), encode/nthash <= number of windows.  Too small -> UKM_ERR_CAPACITY.
 *  - A ukm_ctx owns one HIP stream and one growable device workspace; it is NOT thread-safe.
 *    Use one ctx per calling OS thread (the reference calls these seams from several
 *    goroutines: sort.go:257, diff.go:280).  There is no global mutable state.
 *  - Records are (code uint64, taxid uint32) in structure-of-arrays form (kmers.go:24-27 is
 *    the AoS Go struct).  `taxids == NULL` means "no taxid information".
 *  - Sorted inputs are required where the reference requires the sorted flag
 *    (inter.go:139-141, diff.go:115-117); the library verifies sortedness on the fly where
 *    that is free and returns UKM_ERR_UNSORTED.
 */
#ifndef UNIKMER_HIP_H
#define UNIKMER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UKM_OK 0
#define UKM_ERR_INVALID (-1)      /* bad argument */
#define UKM_ERR_HIP (-2)          /* HIP runtime failure (message has the hipError string) */
#define UKM_ERR_NOMEM (-3)        /* device or host allocation failed */
#define UKM_ERR_ILLEGAL_BASE (-4) /* kmers.ErrIllegalBase: a window contains a non-IUPAC byte */
#define UKM_ERR_UNSORTED (-5)     /* an input that must be sorted is not */
#define UKM_ERR_NO_TAXONOMY (-6)  /* taxids present but ukm_taxonomy_load was not called */
#define UKM_ERR_CAPACITY (-7)     /* out_cap too small; *n_out holds the required size */
#define UKM_ERR_K (-8)            /* k out of range (1..32 codes, 1..64 hashes; count.go:81-87) */
#define UKM_ERR_PEER (-9)         /* a collective call: another rank reported a failure; no rank went on (see that rank) */

/* scan / merge modes: sort.go:484-572 (-u / -d / plain), util-sort.go:35-190 (chunk protocol) */
#define UKM_PLAIN 0
#define UKM_UNIQUE 1
#define UKM_REPEATED 2
#define UKM_REPEATED_CHUNK 3
#define UKM_SINGLETON 4      /* codes seen exactly once: `count -u` (count.go:424-432,475-486) */

/* 2-way set operations */
#define UKM_OP_UNION 0
#define UKM_OP_INTER 1
#define UKM_OP_DIFF 2

/* flags */
#define UKM_F_MIX_TAXID 2u   /* inter --mix-taxid, inter.go:229-236 */
#define UKM_F_CMP_TAXID 4u   /* diff -t/--compare-taxid, diff.go:361-362,406-407 */
#define UKM_F_DEVICE_STREAMS 256u /* n-way calls (union / inter / diff / common): every keys[i] / taxids[i] is a DEVICE pointer.
                                   * Without it each pointer is classified with hipPointerGetAttributes (host arrays are
                                   * staged), which for a fold over 1000 files is 2000 driver queries = most of the call's
                                   * host time.  A host that keeps decoded .unik streams on the device sets it. */

typedef struct ukm_ctx ukm_ctx;

/* ---- library / context -------------------------------------------------------------- */
const char *ukm_last_error(void);
int ukm_version(void);                       /* 1000*major + minor */
int ukm_device_count(int *n);
int ukm_ctx_create(int device, ukm_ctx **out);
int ukm_ctx_destroy(ukm_ctx *ctx);
/* borrow a caller's hipStream_t (e.g. the framework's current stream) instead of the ctx's own
 * non-blocking stream; NULL is HIP's default (null) stream.  Work the caller enqueued on that
 * stream before a ukm_* call is ordered before the call's kernels. */
int ukm_ctx_set_stream(ukm_ctx *ctx, void *hip_stream);
int ukm_ctx_sync(ukm_ctx *ctx);
/* pre-size the device workspace so that later calls do not allocate */
int ukm_ctx_reserve(ukm_ctx *ctx, uint64_t bytes);
/* give the device workspace back to the driver (it grows to what the largest call needed and is kept for the next one:
 * a 100-file union of 1e10 records leaves 160 GB behind); the next call allocates again */
int ukm_ctx_trim(ukm_ctx *ctx);
/* device-memory helpers for hosts without their own allocator (the cgo shim) */
int ukm_dev_alloc(ukm_ctx *ctx, uint64_t bytes, void **dptr);
int ukm_dev_free(ukm_ctx *ctx, void *dptr);
int ukm_copy(ukm_ctx *ctx, void *dst, const void *src, uint64_t bytes); /* any direction, synchronous */
/* Streaming uploads / downloads (the host side of count.go:285-299: FASTA/Q is read in chunks while the device
 * works on the previous chunk).  ukm_host_alloc gives page-locked host memory; ukm_copy_async enqueues a copy
 * (any direction) on the context's TRANSFER stream and returns at once: it starts after the compute work the
 * context was given before the call and then runs beside later compute calls.  ukm_copy_fence orders every
 * LATER compute call behind the transfers issued so far (device-side wait, the host does not block);
 * ukm_copy_sync blocks the host until they are done (before a host source buffer is reused or a host
 * destination is read).  Double buffering: copy_async(chunk i+1) ; compute(chunk i) ; copy_fence ; ... */
int ukm_host_alloc(ukm_ctx *ctx, uint64_t bytes, void **hptr);
int ukm_host_free(ukm_ctx *ctx, void *hptr);
int ukm_copy_async(ukm_ctx *ctx, void *dst, const void *src, uint64_t bytes);
int ukm_copy_fence(ukm_ctx *ctx);
int ukm_copy_sync(ukm_ctx *ctx);
/* ms between hipEvents recorded on the ctx stream (a) around the DOMINANT kernel of the most
 * recent compute call (the tiled set-op kernel for ukm_setop2; falls back to (b) when a call
 * records none) and (b) around all device work of the call */
int ukm_last_kernel_ms(ukm_ctx *ctx, float *ms);
int ukm_last_call_ms(ukm_ctx *ctx, float *ms);
/* diagnostic: which internal route answered the most recent n-way call (ukm_union / ukm_merge_k / ukm_common):
 * 0 none / 2-way only, 1 pairwise tree of 2-way kernels, 2 multi-level k-way streaming merge, 3 LDS hash-probe union,
 * 4 single-pass range merge (one LDS tile per value range), 5 the same pass counting the records of every code
 * (ukm_common below the number of files), 6 ukm_common / ukm_merge_k -d by counting hash probes, 7 keep-everything merge
 * by placement (counts per code, runs written in one piece).  Tests use it to see that a knob took effect. */
int ukm_last_route(ukm_ctx *ctx);

/* ---- route policy as API (round 5).  The n-way entry points choose between several internal routes (ukm_last_route) by
 *      the shape of their inputs; the thresholds can be overridden PER CONTEXT:
 *        ukm_ctx_set_option(ctx, key, value) / ukm_ctx_unset_option / ukm_ctx_get_option (is_set = 0: the library decides).
 *      Keys a host may care about (value semantics as the UKM_<KEY> developer variables of DESIGN.md 4.12):
 *        "punion"  0 never take the hash-probe union / counting probes, 1 whenever the shape allows (size thresholds
 *                  ignored), 2 also without the hit-rate and load guards;   "punion_tax" 0: records with taxids never;
 *        "punion_ranked" 0: files with one taxid each go through the generic taxid tables;
 *        "place" 0 / 1 keep-everything merge by placement never / whenever possible;   "srmerge" 0 / 1 single-pass merge;
 *        "kway" 1 k-way merge also for tiny inputs;   "no_kway" 1 pairwise tree only;   "no_fold" / "no_pfold" 1 the
 *        one-launch range / probe folds of inter and diff off;   "pfold_tax" 0;   "common_probe" 0;   "sort_local" 0 all
 *        radix passes through HBM;   "win_strip" / "nthash_strip" 0 / 1;   "force_ticket" 1 dispatch-order independent kernels
 *        (a context whose look-back watchdog has fired keeps them whatever this option says);   "sort_counting" 0 digit
 *        passes inside every LDS bucket, "sort_fan" 0 no size-class fan-out;   "setop_src" 0 / 1 / 2 the two-launch source-word
 *        route of a 2-way operation with per-record taxids never / for inter / also for union (default: 0 on a taxonomy
 *        with one-byte clade codes, else 1);   "setop_defer" 0 the LCAs of a 2-way union / inter with per-record taxids inside the
 *        merge step instead of densely behind it;   "punion_clade" / "srmerge_clade" 0 / 1
 *        clade codes in the probe tables / the single pass's emit never / always.
 *      The environment is read ONCE, when a context is created: every UKM_* variable present then is the context's default
 *      for the matching key; no compute call calls getenv (a context created under UKM_ENV_LIVE=1 -- the test suite, which
 *      flips knobs between calls -- keeps looking).  An explicitly set option always wins.
 *      ukm_ctx_get_stat: "punion_attempts" = base sets the last hash-probe union / counting-probe call built (2: its retry
 *      with four times the files ran), "workspace_bytes" = device workspace currently held by the context, "sort_fused_hist" = sorts of this context whose first
 *      digit histogram came from the kernel that produced the keys (ukm_count) instead of a pass of their own. */
int ukm_ctx_set_option(ukm_ctx *ctx, const char *key, long long value);
int ukm_ctx_unset_option(ukm_ctx *ctx, const char *key);
int ukm_ctx_get_option(ukm_ctx *ctx, const char *key, long long *value, int *is_set);
int ukm_ctx_get_stat(ukm_ctx *ctx, const char *key, unsigned long long *value);

/* ---- taxonomy: replaces taxdump.NewTaxonomyFromNCBI / LoadMergedNodesFromNCBI / LCA
 *      (util.go:119-171; 14 taxondb.LCA call sites).
 *      child/parent = the first two columns of nodes.dmp; merged_* = merged.dmp (may be NULL).
 *      Contract (taxdump parity is unpinned): LCA(0,x)=LCA(x,0)=0; LCA(x,x)=x;
 *      merged ids are remapped; ids absent from nodes.dmp -> 0.
 *      The device tables are dense in the taxid (>= 27 bytes per id up to the largest one, 16 more per id for every four
 *      levels of depth): NCBI's dump takes ~0.7 GB; a dump with sparse huge ids is refused (UKM_ERR_NOMEM, message says
 *      how much it would need) when that exceeds the device's free memory -- counted after the context's cached workspace
 *      has been given back and with the tables being replaced credited.  A load either replaces the context's taxonomy
 *      completely or, on any error, leaves the previous one in place (one exception: new tables that only fit WITHOUT the
 *      old ones make the old ones go first; a failure after that leaves the context without a taxonomy). */
int ukm_taxonomy_load(ukm_ctx *ctx, const uint32_t *child, const uint32_t *parent, uint64_t n,
                      const uint32_t *merged_old, const uint32_t *merged_new, uint64_t m);
int ukm_taxonomy_max_taxid(ukm_ctx *ctx, uint32_t *max_taxid); /* taxdump.MaxTaxid, util.go:169 */
int ukm_lca(ukm_ctx *ctx, const uint32_t *a, const uint32_t *b, uint64_t n, uint32_t *out);

/* ---- encode: replaces sketches.NewKmerIterator(seq,k,canonical,circular).NextKmer()
 *      (count.go:321,363) = kmers v0.1.0 2-bit encode + canonical.
 *      bases = concatenated records, rec_off[n_rec+1] = record boundaries.  Records shorter
 *      than k are skipped (sketches.ErrShortSeq, count.go:323-328).  Output = every window of
 *      every record in order. */
int ukm_encode_kmers(ukm_ctx *ctx, const uint8_t *bases, const uint64_t *rec_off,
                     uint64_t n_rec, int k, int canonical, int circular, uint64_t *out,
                     uint64_t out_cap, uint64_t *n_out);

/* ---- ntHash: replaces sketches.NewHashIterator(...).NextHash() (count.go:319,361) and
 *      nthash.NewHasher(&seq,k).Next(canonical) (dump.go:253-260) = ntHash v1; fused with the
 *      Scaled-MinHash filter `code > maxHash -> skip` (count.go:98,373-375) when max_hash != 0.
 *      Output keeps window order. */
int ukm_nthash(ukm_ctx *ctx, const uint8_t *bases, const uint64_t *rec_off, uint64_t n_rec,
               int k, int canonical, int circular, uint64_t max_hash, uint64_t *out,
               uint64_t out_cap, uint64_t *n_out);
/* ---- minimizer sketch: replaces sketches.NewMinimizerSketch(seq,k,w,circular).NextMinimizer()
 *      (count.go:316,357; bio v0.13.3, SURVEY.md B3, KATs C-7/C-8): canonical ntHash of every
 *      window; for each group of w consecutive windows of a record the LEFTMOST minimum, emitted
 *      when the arg-min position differs from the previous group's.  Records with fewer than w
 *      windows give nothing.  max_hash != 0 applies the Scaled filter to the emitted values
 *      (count.go:373-375).  out_pos (may be NULL) = window index of each minimizer inside its
 *      record (sketch.Index()).  1 <= w <= 1024.  Output keeps record/group order. */
int ukm_minimizer(ukm_ctx *ctx, const uint8_t *bases, const uint64_t *rec_off, uint64_t n_rec,
                  int k, int w, int circular, uint64_t max_hash, uint64_t *out,
                  uint64_t *out_pos, uint64_t out_cap, uint64_t *n_out);
/* count.go:98  maxHash = uint64(float64(^uint64(0)) / float64(scale)) */
uint64_t ukm_max_hash(uint64_t scale);
/* ---- `count` in one call: replaces the body of the Run closure count.go:285-436 (iterator + `m[code] = struct{}{}` per
 *      k-mer, the -u / -d marks of count.go:424-432) and its sort count.go:581: every window of every record (codes when
 *      hashed = 0, ntHash v1 with the Scaled filter max_hash != 0 when hashed = 1) -> sort -> mode UKM_UNIQUE (the distinct
 *      set), UKM_REPEATED (`-d`: codes seen at least twice) or UKM_SINGLETON (`-u`: exactly once), sorted ascending (`-s`).
 *      The windows stay in the context's device workspace (8 B per base while the call runs); only the result is written to
 *      out[out_cap] (host or device).  The same result as ukm_encode_kmers / ukm_nthash + ukm_sort_u64 + ukm_unique with one
 *      stream synchronisation instead of three.  Too small an out_cap: UKM_ERR_CAPACITY, *n_out = the size needed. */
int ukm_count(ukm_ctx *ctx, const uint8_t *bases, const uint64_t *rec_off, uint64_t n_rec, int k, int canonical,
              int circular, int hashed, uint64_t max_hash, int mode, uint64_t *out, uint64_t out_cap, uint64_t *n_out);

/* ---- sorts: replace sortutil.Uint64s (count.go:581, union.go:274,295, sort.go:463 ...) and
 *      sorts.Quicksort(CodeTaxidSlice) (sort.go:268,331,457).  In place, ascending by code;
 *      pairs are sorted by code only, stably.  key_bits = number of significant low bits
 *      (2k for k-mer codes, 0 or 64 for hashes): higher radix passes are skipped.
 *      Inputs of 2^32 records or more are sorted as 2^31-record chunks and merged on the
 *      device (the reference's own `sort -m` protocol, in HBM). */
int ukm_sort_u64(ukm_ctx *ctx, uint64_t *keys, uint64_t n, int key_bits);
int ukm_sort_pairs(ukm_ctx *ctx, uint64_t *keys, uint32_t *taxids, uint64_t n, int key_bits);

/* ---- scans over a sorted stream: replace sort.go:484-572 and dumpCodes2File /
 *      dumpCodesTaxids2File (util-sort.go:35-190).  taxids may be NULL. */
int ukm_unique(ukm_ctx *ctx, const uint64_t *keys, const uint32_t *taxids, uint64_t n,
               int mode, uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap,
               uint64_t *n_out);

/* ---- k-way merge: replaces mergeChunksFile (util-sort.go:227-606).  Streams are expected
 *      to be sorted (chunk files); an unsorted one is tolerated (the call then sorts the
 *      concatenation instead of merging).  mode/final_round as the reference's
 *      unique/repeated/finalRound arguments; equal codes keep stream order. */
int ukm_merge_k(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids,
                const uint64_t *lens, int nstreams, int mode, int final_round,
                uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out);

/* ---- 2-way set operation on two SORTED streams — the device hot path that the n-way
 *      entries below are folded from, and what bench.py measures. */
int ukm_setop2(ukm_ctx *ctx, int op, const uint64_t *a_keys, const uint32_t *a_taxids,
               uint64_t na, const uint64_t *b_keys, const uint32_t *b_taxids, uint64_t nb,
               uint32_t flags, uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap,
               uint64_t *n_out);

/* ---- n-way set operations; output is the sorted (code, taxid) stream.
 *      union  : union.go:186-305  (inputs need not be sorted)
 *      inter  : inter.go:188-286  (all inputs sorted)
 *      diff   : diff.go:341-454   (first input sorted; sorted_flags[i]==0 marks unsorted ones,
 *                                  NULL = all sorted)
 *      common : common.go:220-344 (threshold = number of files, common.go:93-105) */
int ukm_union(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids,
              const uint64_t *lens, int nstreams, uint32_t flags, uint64_t *out_keys,
              uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out);
int ukm_inter(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids,
              const uint64_t *lens, int nstreams, uint32_t flags, uint64_t *out_keys,
              uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out);
int ukm_diff(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids,
             const uint64_t *lens, int nstreams, const uint8_t *sorted_flags, uint32_t flags,
             uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out);
int ukm_common(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids,
               const uint64_t *lens, int nstreams, uint32_t threshold, uint32_t flags,
               uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out);
uint32_t ukm_common_threshold(uint32_t nfiles, double proportion, uint32_t number);

/* ---- per-FILE taxids (round 5).  The reference's documented taxid workflow is ONE taxid per file: `unikmer count -t 511145`
 *      (README.md:170) stores it in the .unik header (count.go:466-468 writer.SetGlobalTaxid) and unik.Reader.ReadCodeWithTaxid
 *      then hands that value out with EVERY record, so union.go:187-201, inter.go:190,211-239, diff.go:404-409 and
 *      common.go:262-266 fold it like a per-record taxid.  The _ft entry points take that value as a scalar instead of an
 *      n-element array of copies:
 *        file_taxids (host array [nstreams], or NULL) / a_file_taxid, b_file_taxid: the taxid of every record of a stream
 *        whose per-record pointer (taxids[i] / a_taxids / b_taxids) is NULL; 0 = the stream has no taxid information (what
 *        a NULL pointer alone means in the entry points above, which are these with file_taxids = NULL).
 *      Results are identical to passing the expanded arrays.  What the device does instead: two streams with one taxid
 *      each run the plain-key kernel and write one of three values (A's, B's, their LCA) per output record; `inter`,
 *      `diff` (also -t: whether file j can take a matched code away is one decision per file) and `common` over all files
 *      are the plain operation and a fill with one value worked out once; the hash-probe `union` / `common` below the number
 *      of files / `merge -d` read no taxid and look no pre-order number up for such a file; the k-way merges get the array
 *      built on the device.  Streams with per-record taxids and streams with one per file may be mixed freely. */
int ukm_setop2_ft(ukm_ctx *ctx, int op, const uint64_t *a_keys, const uint32_t *a_taxids, uint32_t a_file_taxid,
                  uint64_t na, const uint64_t *b_keys, const uint32_t *b_taxids, uint32_t b_file_taxid, uint64_t nb,
                  uint32_t flags, uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out);
int ukm_union_ft(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids, const uint32_t *file_taxids,
                 const uint64_t *lens, int nstreams, uint32_t flags, uint64_t *out_keys, uint32_t *out_taxids,
                 uint64_t out_cap, uint64_t *n_out);
int ukm_inter_ft(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids, const uint32_t *file_taxids,
                 const uint64_t *lens, int nstreams, uint32_t flags, uint64_t *out_keys, uint32_t *out_taxids,
                 uint64_t out_cap, uint64_t *n_out);
int ukm_diff_ft(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids, const uint32_t *file_taxids,
                const uint64_t *lens, int nstreams, const uint8_t *sorted_flags, uint32_t flags, uint64_t *out_keys,
                uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out);
int ukm_common_ft(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids, const uint32_t *file_taxids,
                  const uint64_t *lens, int nstreams, uint32_t threshold, uint32_t flags, uint64_t *out_keys,
                  uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out);
int ukm_merge_k_ft(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids, const uint32_t *file_taxids,
                   const uint64_t *lens, int nstreams, int mode, int final_round, uint64_t *out_keys,
                   uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out);

/* ---- multi-GPU helper: split points of a sorted stream for prefix sharding (SURVEY.md
 *      §8(e)): cuts[g] = lower_bound(keys, splitters[g]) for g in [0, n_split). */
int ukm_partition_points(ukm_ctx *ctx, const uint64_t *keys, uint64_t n,
                         const uint64_t *splitters, int n_split, uint64_t *cuts);

/* ---- multi-GPU exchange over RCCL / xGMI (new): one process and one ukm_ctx per GPU.
 *      The reference is single-process; this is the step that lets a host shard the code space by high-bits prefix:
 *        ukm_prefix_splitters -> ukm_partition_points (cuts of every local sorted file) -> ukm_shard_exchange (slice g
 *        of every rank travels to rank g) -> ukm_union / ukm_merge_k of the received slices -> the 1-GPU operation on
 *        the rank's range; the ranks' results concatenated in rank order are the global sorted result.
 *      ukm_comm_get_unique_id: 128 opaque bytes (ncclUniqueId) made by ONE rank and handed to the others by the host
 *      (file, socket, MPI ...).  ukm_comm_init is collective.  RCCL is loaded on first use (dlopen): a single-GPU host
 *      needs no RCCL at all.
 *      ukm_shard_exchange: send_counts[nranks] (host) = number of consecutive records of keys/taxids for each rank in
 *      rank order (their sum is the stream length); recv_counts[nranks] (host, out) = records received from each rank;
 *      the received slices are stored back to back in source-rank order (each is sorted; together they are this rank's
 *      range of the logical file).  keys / taxids / out_* may be host or device pointers. */
#define UKM_COMM_ID_BYTES 128
int ukm_comm_get_unique_id(void *id);
int ukm_comm_init(ukm_ctx *ctx, int nranks, int rank, const void *id);
int ukm_comm_destroy(ukm_ctx *ctx);
int ukm_comm_info(ukm_ctx *ctx, int *nranks, int *rank);
int ukm_prefix_splitters(int key_bits, int nranks, uint64_t *splitters);
int ukm_shard_exchange(ukm_ctx *ctx, const uint64_t *keys, const uint32_t *taxids, const uint64_t *send_counts,
                       uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *recv_counts, uint64_t *n_out);
/*      Capacity is decided COLLECTIVELY: the slice sizes travel together with every rank's out_cap, and either all
 *      ranks exchange or all ranks return UKM_ERR_CAPACITY (n_out = what this rank would have received), so a short
 *      buffer on one rank can never leave its peers blocked in RCCL.  ukm_shard_plan is that decision as a pure host
 *      function (all = [source rank][nranks slice sizes | out_cap of the source rank], the gathered matrix; bit 63 of
 *      the capacity word = UKM_SHARD_HAS_TAXIDS, "this rank passed taxids": ranks that disagree all return UKM_ERR_INVALID
 *      before anything is posted -- a mixed call would leave taxid transfers unmatched).
 *      Hosts that move many files use the two-step form: ukm_shard_counts (send_counts[nfiles][nranks] ->
 *      recv_counts[nfiles][nranks], ONE all-gather and host round trip for all files; size the receive buffers
 *      from it), then ukm_shard_exchange_known per file, which has no gather and no host round trip in front of the
 *      transfers (own slice: device-to-device copy; peers: one ncclSend / ncclRecv group on the context's stream; like
 *      every entry point the call returns when its stream work is done).  It takes no collective decision: a rank whose
 *      buffer is too small, or whose own entries of send_counts / recv_counts disagree, still takes part (what arrives is
 *      dropped in a scratch buffer of the largest slice) and returns UKM_ERR_CAPACITY / UKM_ERR_INVALID afterwards.  A
 *      device failure in front of the transfers (no memory for staging) leaves the peers blocked in RCCL: destroy the
 *      communicator, as after any lost rank.  Files WITH taxids: pass ukm_shard_counts_tax the flags has_taxids[nfiles]
 *      (1: this rank will hand ukm_shard_exchange_known a taxid array for file f); the flags ride in the same gather and
 *      ranks that disagree about a file all get UKM_ERR_INVALID here, before any transfer is posted
 *      (ukm_shard_counts_plan: that decision as a pure host function over the gathered words, [rank][nfiles * nranks slice
 *      sizes | nfiles flags, 2 = not declared]). */
#define UKM_SHARD_HAS_TAXIDS (1ull << 63)
#define UKM_SHARD_RANK_FAILED (~0ull) /* ukm_shard_splitters: a rank's record-count word when its preparation failed */
int ukm_shard_plan(int nranks, int rank, const uint64_t *all, uint64_t *recv_counts, uint64_t *n_out);
/*      Sampled splitters (k-mer codes are not uniform in their top bits -- README.md:177-180, sorted
 *      k-mers start AAAAAAAAA... -- so equal-width ranges leave the ranks unevenly loaded): ukm_shard_splitters is
 *      collective; every rank passes the sorted files it holds and gets the same nranks + 1 boundaries, cut so that the
 *      ranks receive about the same number of records (1024 regular samples per rank, one all-gather).  Use them in
 *      place of ukm_prefix_splitters; any non-decreasing boundaries give the same concatenated result.  A rank that
 *      fails locally (bad argument, no memory for staging) still takes part in the gather -- with UKM_SHARD_RANK_FAILED as
 *      its record count -- and returns its error afterwards; EVERY other rank then returns UKM_ERR_PEER, so all hosts
 *      abort the exchange together (nobody goes on to ukm_shard_counts with a rank missing); only a rank that cannot even
 *      allocate the 8 KB gather buffers hangs its peers (destroy the communicator, as after any lost rank).  Host arrays
 *      are sampled where they lie.
 *      ukm_shard_splitters_plan is the decision as a pure host function over the gathered words
 *      ([rank][1 + per_rank] = record count, samples). */
int ukm_shard_splitters(ukm_ctx *ctx, const uint64_t *const *keys, const uint64_t *lens, int nfiles, int key_bits,
                        uint64_t *splitters);
int ukm_shard_splitters_plan(int nranks, int per_rank, const uint64_t *all, int key_bits, uint64_t *splitters);
int ukm_shard_counts(ukm_ctx *ctx, const uint64_t *send_counts, int nfiles, uint64_t *recv_counts);
int ukm_shard_counts_tax(ukm_ctx *ctx, const uint64_t *send_counts, int nfiles, const uint8_t *has_taxids, uint64_t *recv_counts);
int ukm_shard_counts_plan(int nranks, int rank, int nfiles, const uint64_t *all, uint64_t *recv_counts);
int ukm_shard_exchange_known(ukm_ctx *ctx, const uint64_t *keys, const uint32_t *taxids, const uint64_t *send_counts,
                             const uint64_t *recv_counts, uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap,
                             uint64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif
