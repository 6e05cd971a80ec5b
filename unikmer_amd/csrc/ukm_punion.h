// ukm_punion.h — internal: `union` of many heavily overlapping sorted sets by LDS hash probes (ukm_punion.hip)
#pragma once
#include "ukm_internal.h"

// developer / test knob UKM_PUNION: 0 = never, 1 = whenever the shape allows it (size thresholds ignored),
// 2 = as 1 and without the hit-rate guard.  Unset: the library's own choice.
int ukm_punion_mode(const ukm_ctx *c);
// UKM_PUNION_TAX=0: records with TaxIds never take this path
int ukm_punion_tax_mode(const ukm_ctx *c);
// *fallback = true: not applicable to these inputs (low overlap, unsorted stream, miss buffer overflow): the
// caller's k-way merge answers; nothing was written that matters.
// tax: the records carry TaxIds (taxids[j] may be null: all 0); the result's TaxId is the LCA over every record of a code.
int ukm_dev_probe_union(ukm_ctx *c, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax, u64 *out,
                        u32 *tout, u64 out_cap, u64 *n_out, bool *fallback, const u32 *ctax = nullptr, bool overlap_known = false);
// (ctax, may be null: the ONE taxid of a file whose taxids[j] is null -- the .unik header's global taxid; such a file's
//  records load no taxid and look no pre-order number up)
// `common` below the number of files through the same tables with a record count per entry; keys[0] = the first file as a
// sorted duplicate-free set (first_once), or -- !first_once -- every record of every file counts (`merge -d`).  *fallback as above.
int ukm_dev_probe_common(ukm_ctx *c, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax,
                         u32 threshold, u64 *out, u32 *tout, u64 out_cap, u64 *n_out, bool *fallback, bool first_once = true,
                         const u32 *ctax = nullptr);
// Keep-everything merge of many files that share most of their codes, by placement (ukm_punion.hip, pl_merge_kernel):
// developer knob UKM_PLACE: 0 = never, 1 = whenever the shape allows it.  *fallback as above.
int ukm_place_mode(const ukm_ctx *c);
int ukm_dev_place_merge(ukm_ctx *c, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax, u64 *out,
                        u32 *tout, u64 out_cap, u64 *n_out, bool *fallback, const u32 *ctax = nullptr);
