// ukm_srmerge.h — internal entry of the single-pass many-stream merge / union (ukm_srmerge.hip)
#pragma once
#include "ukm_internal.h"
#include "ukm_kway.h"  // UKM_KWAY_UNION / UKM_KWAY_MERGE

// developer / test knob UKM_SRMERGE: 0 = never, 1 = whenever the shape allows it (size thresholds ignored).
// Unset: the library's own choice (many streams, enough records).
int ukm_srmerge_mode(const ukm_ctx *c);
// Same contract as ukm_dev_kway: all pointers are device pointers; *fallback = true: the inputs are not for this
// path (too few / too many streams, an unsorted stream, one code with more copies than a tile holds) and the caller's
// multi-level merge answers; nothing that matters was written.
// threshold > 1 with UKM_KWAY_UNION: only the codes that have at least that many records (`common` below the number of
// files: common.go:331-335), TaxId as for the union.
int ukm_dev_srmerge(ukm_ctx *c, int op, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S,
                    bool tax, u64 *out, u32 *tout, u64 out_cap, u64 *n_out, bool *fallback, u32 threshold = 0);
