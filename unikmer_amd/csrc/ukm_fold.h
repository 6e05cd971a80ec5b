// ukm_fold.h — internal: `inter` / `diff` over many sorted streams as one range-partitioned launch (ukm_fold.hip)
#pragma once
#include "ukm_internal.h"

bool ukm_fold_enabled(const ukm_ctx *c);  // UKM_NO_FOLD=1 switches it off (developer knob)
int ukm_dev_range_fold(ukm_ctx *c, int op, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax,
                       u32 flags, u64 *out, u32 *tout, u64 out_cap, u64 *n_out, bool *fallback);
