// Arguments of the row-block training kernel (rowblock.hip), shared with the step driver (runtime.hip).
#pragma once
#include "mt_common.h"

struct MtRbBlock {
  int32_t slot, row0;        // slot < 0: padding entry; row0: first row (bits 0-19), part (20-23), parts of the block (24-27)
};

struct MtRbArgs {
  const MtGemmGroup* fwd[MT_MAX_LAYERS];       // per layer: the pack's forward groups (indexed by physical slot)
  const MtGemmGroup* dgrad[MT_MAX_LAYERS];     // per layer: its input-gradient groups
  const MtHeadSlot* head;
  const MtRbBlock* blocks;
  float* scratch;                              // head records, (slot * max_chunks + chunk) * rec
  int32_t n_hidden, max_chunks, rec, pad;
  int32_t* flags;                              // split blocks: one flag per work-list entry + one error word (null: no entry is split)
  int32_t n_flags, pad2;
  unsigned long long* prof;                    // diagnostics: 16 clock stamps per workgroup (null: none are taken)
};

