// ukm_pfold.h — internal: `inter` / `diff` over many sorted sets by LDS hash probes (ukm_pfold.hip)
#pragma once
#include "ukm_internal.h"

bool ukm_pfold_enabled(const ukm_ctx *c);  // UKM_NO_PFOLD=1 switches it off (developer knob)
// Same contract as ukm_dev_range_fold (ukm_fold.h).  *fallback = true: not this path (inter --mix-taxid, diff -t, a
// duplicate or all-ones code, an unsorted stream, a shape it does not fit): the caller tries the range fold next.
int ukm_dev_probe_fold(ukm_ctx *c, int op, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax,
                       u32 flags, u64 *out, u32 *tout, u64 out_cap, u64 *n_out, bool *fallback);
