// ukm_kway.h — internal entry of the k-way streaming merge (ukm_kway.hip)
#pragma once
#include "ukm_internal.h"

#define UKM_KWAY_UNION 0  /* one record per distinct code, TaxId = LCA over all occurrences */
#define UKM_KWAY_MERGE 1  /* every record kept, equal codes in stream order */

// developer knobs: UKM_NO_KWAY=1 keeps the pairwise tree, UKM_KWAY_K = 4 | 8 | 16 sets the fan-in
bool ukm_kway_enabled(const ukm_ctx *c);
int ukm_kway_fanin(const ukm_ctx *c);
int ukm_dev_kway(ukm_ctx *c, int op, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S,
                 bool tax, u64 *out, u32 *tout, u64 out_cap, u64 *n_out, bool *fallback);
