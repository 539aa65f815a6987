// Host-side step driver for a pack of MLP trials (gfx950).
//
// The Python layer describes a pack once per epoch as a plan: device pointers to the per-layer group
// and tile tables, the gather / head / optimiser descriptor tables and the parameter slabs.  The two
// entry points below then enqueue whole training steps (or a whole validation sweep) from C++ with no
// interpreter in the loop and no host-device synchronisation: the host only ever waits at the end of
// an epoch, when it reads the metric table.  Per step the launch sequence is
//   gather -> L x forward GEMM(+bias+ReLU) -> head+loss+dH (+ its reduce) -> for l = L-1..0 { wgrad(+dbias) and
//   dgrad(+ReLU mask) of layer l in ONE launch } -> [grad sq-norm partials] -> fused per-trial optimiser
//   (+bf16 shadow refresh), i.e. 2L + 4 (+1) launches regardless of how many trials share the GPU.
#include "mt_bodies.h"
#include "mt_rowblock.h"
#include <stdlib.h>


extern "C" {
int mt_gemm_launch(int mode, const MtGemmGroup* groups, const MtTile* tiles, int ntiles, int step, int flags,
                   int arow0, int i_override, int ksplits, size_t split_stride, int nstage, hipStream_t stream);
int mt_gemm_bwd_launch(const MtGemmGroup* wgroups, const MtTile* wtiles, int n_wtiles, const MtGemmGroup* dgroups,
                       const MtTile* dtiles, int n_dtiles, int step, int ksplits, size_t split_stride, int nstage,
                       hipStream_t stream);
int mt_gather_launch(const MtGatherSlot* slots, const int32_t* ids, int n_active, int max_batch, const float* X,
                     const float* Y, uint32_t n_rows, int F, int Fpad, int hb, int step, hipStream_t stream);
int mt_head_loss_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_batch, int max_width,
                        int step, float* scratch, hipStream_t stream);
int mt_head_eval_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_width, int row0, int nrows,
                        int max_rows, float* scratch, hipStream_t stream);
int mt_optim_launch(float* P, const float* G, float* M, float* V, void* Pb, size_t slot_stride, const MtOptSlot* hp,
                    const int32_t* ids, int n_active, int max_params, float* partials, int max_blocks, int use_clip,
                    int step, int gsplits, size_t gsplit_stride, hipStream_t stream);
int mt_gemm_bwd_head_launch(const MtGemmGroup* wgroups, const MtTile* wtiles, int n_wtiles, const MtGemmGroup* dgroups,
                            const MtTile* dtiles, int n_dtiles, int step, int ksplits, size_t split_stride, int nstage,
                            MtHeadReduceArgs hr, hipStream_t stream);
int mt_head_loss_partials_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_batch, int max_width,
                                 int step, float* scratch, hipStream_t stream);
int mt_gemm_f32_launch(int mode, const MtGemmGroup* groups, const MtTile* tiles, int ntiles, int step, int flags,
                       int arow0, int i_override, int ksplits, size_t split_stride, int tile_edge, hipStream_t stream);
int mt_gemm_f32_bwd_head_launch(const MtGemmGroup* wgroups, const MtTile* wtiles, int n_wtiles, const MtGemmGroup* dgroups,
                                const MtTile* dtiles, int n_dtiles, int step, int ksplits, size_t split_stride,
                                int tile_edge, MtHeadReduceArgs hr, hipStream_t stream);
int mt_gather_f32_launch(const MtGatherSlot* slots, const int32_t* ids, int n_active, int max_batch, const float* X,
                         const float* Y, uint32_t n_rows, int F, int Fpad, int hb, int step, hipStream_t stream);
int mt_head_loss_f32_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_batch, int max_width,
                            int step, float* scratch, int with_reduce, hipStream_t stream);
int mt_head_eval_f32_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_width, int row0, int nrows,
                            int max_rows, float* scratch, hipStream_t stream);
int mt_optim_gather_launch(float* P, const float* G, float* M, float* V, void* Pb, size_t slot_stride, const MtOptSlot* hp,
                           const int32_t* ids, int n_active, int max_params, float* partials, int max_blocks, int use_clip,
                           int step, int gsplits, size_t gsplit_stride, const MtGatherSlot* gslots, const float* X,
                           const float* Y, uint32_t n_rows, int F, int Fpad, int hb, int max_batch, hipStream_t stream);
int mt_mlp_rowblock_launch(const MtRbArgs* args, int n_blocks, int mt, int step, hipStream_t stream);
}

struct MtMlpPlan {
  int32_t n_hidden;      // hidden (GEMM) layers; the scalar head is handled by the head kernels
  int32_t n_active;      // occupied slots; active_ids lists their physical slot numbers
  int32_t max_batch, max_width, max_params;
  int32_t F, Fpad, hb;
  uint32_t n_rows;
  int32_t use_clip;
  const float* X;
  const float* Y;
  const MtGatherSlot* gather;
  const MtGemmGroup* fwd_groups[MT_MAX_LAYERS];
  const MtTile* fwd_tiles[MT_MAX_LAYERS];
  const MtGemmGroup* dgrad_groups[MT_MAX_LAYERS];
  const MtTile* dgrad_tiles[MT_MAX_LAYERS];
  const MtGemmGroup* wgrad_groups[MT_MAX_LAYERS];
  const MtTile* wgrad_tiles[MT_MAX_LAYERS];
  int32_t fwd_ntiles[MT_MAX_LAYERS];
  int32_t dgrad_ntiles[MT_MAX_LAYERS];
  int32_t wgrad_ntiles[MT_MAX_LAYERS];
  const MtHeadSlot* head;
  float* P;
  float* G;
  float* M;
  float* V;
  void* Pb;
  size_t slot_stride;
  const MtOptSlot* opt;
  float* partials;        // max_blocks floats per physical slot (gradient sq-norm partial sums)
  const int32_t* active_ids;
  float* head_scratch;    // mt_head_scratch_floats(physical slots, max batch, max width) floats
  int32_t max_blocks;
  int32_t ksplits;        // wgrad contraction splits; G holds that many slabs of gsplit_stride floats
  size_t gsplit_stride;
  int32_t ksplits0;       // contraction splits of layer 0's wgrad (its tile list is built with this count)
  int32_t f32;            // 1: exact-f32 pack (f32 activations, GEMMs read P, Pb is null); 0: bf16 MFMA with f32 masters
  // tile edge of this layer's forward / backward tile lists: 0 = 128; 1 = 256 (bf16 packs only); 2 = 64, 3 = 32, 4 = 32 deep,
  // 5 = 64 deep (f32 packs only)
  int32_t fwd_big[MT_MAX_LAYERS];
  int32_t bwd_big[MT_MAX_LAYERS];
  // Row-block form of the step (exact-f32 packs; rb_nblocks > 0 selects it): one launch takes every 16 * rb_mt rows of
  // every trial through all forward layers, the head and all input gradients (rowblock.hip), a second one computes the
  // weight gradients of ALL layers (wg_all_*: the per-layer group tables concatenated, group = layer * capacity + slot,
  // each group carrying its own split count) and adds up the head records, the third is the optimiser.
  const MtRbBlock* rb_blocks;
  int32_t rb_nblocks, rb_mt;
  const MtGemmGroup* wg_all_groups;
  const MtTile* wg_all_tiles;
  int32_t wg_all_ntiles, wg_all_edge;
  int32_t* rb_flags;      // exchange flags of split row blocks: rb_nflags + 1 words (null: no block is split)
  // Later phases of an epoch: once the trials with the largest batches have run out of steps, the chain continues on a
  // work list without them (and with the freed CUs given to further splits).  List k serves steps >= rb_ph_from[k]
  // (ascending; rb_ph_nblocks[k] == 0 ends the table).  Steps whose number is only known on the device (captured
  // blocks) stay on the first list, which holds every trial.
  const MtRbBlock* rb_ph_blocks[3];
  int32_t rb_ph_nblocks[3];
  int32_t rb_ph_from[3];
  int32_t rb_nflags, rb_pad;
};

thread_local const int* mt_step_base_tls = nullptr;

#define MT_TRY(x)            \
  do {                       \
    int _e = (x);            \
    if (_e != 0) return _e;  \
  } while (0)

// plan code -> tile_edge argument of the f32 launchers: 0 = 128-tiles, 2 = 64, 3 = 32 (two stages); 4 = 32-tiles with eight
// stages, 5 = 64-tiles with four (the latency forms)
static inline int f32_edge(int code) {
  switch (code) {
    case 2: return 64;
    case 3: return 32;
    case 4: return 33;
    case 5: return 66;
    default: return 128;
  }
}

// 256-tile launches of bf16 packs use the eight-wave pipelined form (258; 256 is the four-wave one, 30-50 % slower)
static int big_form() { return 258; }

static int plan_gather(const MtMlpPlan* p, int step, hipStream_t stream) {
  return (p->f32 ? mt_gather_f32_launch : mt_gather_launch)(p->gather, p->active_ids, p->n_active, p->max_batch, p->X, p->Y,
                                                            p->n_rows, p->F, p->Fpad, p->hb, step, stream);
}

static int plan_forward(const MtMlpPlan* p, int l, int step, int arow0, int rows, hipStream_t stream) {
  if (p->f32)
    return mt_gemm_f32_launch(0, p->fwd_groups[l], p->fwd_tiles[l], p->fwd_ntiles[l], step, MT_GEMM_FLAG_RELU, arow0, rows,
                              1, 0, f32_edge(p->fwd_big[l]), stream);
  return mt_gemm_launch(0, p->fwd_groups[l], p->fwd_tiles[l], p->fwd_ntiles[l], step, MT_GEMM_FLAG_RELU, arow0, rows, 1, 0,
                        p->fwd_big[l] ? big_form() : 0, stream);
}

static int mlp_enqueue_steps(const MtMlpPlan* p, int step0, int nsteps, hipStream_t stream, bool first_gather) {
  if (p->n_hidden < 1 || p->n_hidden > MT_MAX_LAYERS) return -10;
  if (p->f32 && p->Pb != nullptr) return -16;
  const int L = p->n_hidden;
  // The mini-batch of a step is gathered by the optimiser launch of the step before it; whoever starts a run of
  // steps (first_gather) gathers the first one.  The head kernel leaves partial records that the last layer's backward
  // launch adds up (a 256-tile launch cannot carry them, so then the head reduces on its own).
  const bool head_rides = p->f32 || !p->bwd_big[L - 1];
  const MtHeadReduceArgs hr = {p->head, p->active_ids, p->head_scratch, p->n_active,
                               (p->max_batch + HL_ROWS - 1) / HL_ROWS, ((p->max_width + 7) / 8) * 8 + 8};
  const MtHeadReduceArgs no_hr = {nullptr, nullptr, nullptr, 0, 0, 0};
  if (first_gather) MT_TRY(plan_gather(p, step0, stream));
  if (p->rb_nblocks > 0) {
    if (!p->f32 || p->rb_blocks == nullptr || p->wg_all_groups == nullptr || p->wg_all_tiles == nullptr) return -17;
    MtRbArgs ra = {};
    for (int l = 0; l < L; ++l) { ra.fwd[l] = p->fwd_groups[l]; ra.dgrad[l] = p->dgrad_groups[l]; }
    ra.head = p->head;
    ra.flags = p->rb_flags;
    ra.n_flags = p->rb_nflags;
    ra.scratch = p->head_scratch;
    ra.n_hidden = L;
    ra.max_chunks = hr.max_chunks;
    ra.rec = hr.rec;
    for (int step = step0; step < step0 + nsteps; ++step) {
      ra.blocks = p->rb_blocks;
      int nblocks = p->rb_nblocks;
      if (mt_step_base_tls == nullptr)
        for (int k = 0; k < 3 && p->rb_ph_nblocks[k] > 0 && step >= p->rb_ph_from[k]; ++k) {
          ra.blocks = p->rb_ph_blocks[k];
          nblocks = p->rb_ph_nblocks[k];
        }
      if (ra.flags != nullptr && nblocks > ra.n_flags) return -18;
      MT_TRY(mt_mlp_rowblock_launch(&ra, nblocks, p->rb_mt, step, stream));
      MT_TRY(mt_gemm_f32_bwd_head_launch(p->wg_all_groups, p->wg_all_tiles, p->wg_all_ntiles, nullptr, nullptr, 0, step,
                                         p->ksplits, p->gsplit_stride, f32_edge(p->wg_all_edge), hr, stream));
      MT_TRY(mt_optim_gather_launch(p->P, p->G, p->M, p->V, p->Pb, p->slot_stride, p->opt, p->active_ids, p->n_active,
                                    p->max_params, p->partials, p->max_blocks, p->use_clip, step, p->ksplits,
                                    p->gsplit_stride, p->gather, p->X, p->Y, p->n_rows, p->F, p->Fpad, p->hb,
                                    p->max_batch, stream));
    }
    return 0;
  }
  for (int step = step0; step < step0 + nsteps; ++step) {
    for (int l = 0; l < L; ++l) MT_TRY(plan_forward(p, l, step, 0, 0, stream));
    if (p->f32)
      MT_TRY(mt_head_loss_f32_launch(p->head, p->active_ids, p->n_active, p->max_batch, p->max_width, step,
                                     p->head_scratch, 0, stream));
    else if (head_rides)
      MT_TRY(mt_head_loss_partials_launch(p->head, p->active_ids, p->n_active, p->max_batch, p->max_width, step,
                                          p->head_scratch, stream));
    else
      MT_TRY(mt_head_loss_launch(p->head, p->active_ids, p->n_active, p->max_batch, p->max_width, step,
                                 p->head_scratch, stream));
    for (int l = L - 1; l >= 0; --l) {   // layer 0 has no input gradient: its dgrad tile list is empty
      const int ks = l == 0 ? p->ksplits0 : p->ksplits;
      const int nd = l > 0 ? p->dgrad_ntiles[l] : 0;
      const MtHeadReduceArgs rider = (l == L - 1 && head_rides) ? hr : no_hr;
      if (p->f32)
        MT_TRY(mt_gemm_f32_bwd_head_launch(p->wgrad_groups[l], p->wgrad_tiles[l], p->wgrad_ntiles[l], p->dgrad_groups[l],
                                           p->dgrad_tiles[l], nd, step, ks, p->gsplit_stride, f32_edge(p->bwd_big[l]),
                                           rider, stream));
      else
        MT_TRY(mt_gemm_bwd_head_launch(p->wgrad_groups[l], p->wgrad_tiles[l], p->wgrad_ntiles[l], p->dgrad_groups[l],
                                       p->dgrad_tiles[l], nd, step, ks, p->gsplit_stride, p->bwd_big[l] ? big_form() : 0, rider,
                                       stream));
    }
    MT_TRY(mt_optim_gather_launch(p->P, p->G, p->M, p->V, p->Pb, p->slot_stride, p->opt, p->active_ids, p->n_active,
                                  p->max_params, p->partials, p->max_blocks, p->use_clip, step, p->ksplits,
                                  p->gsplit_stride, p->gather, p->X, p->Y, p->n_rows, p->F, p->Fpad, p->hb,
                                  p->max_batch, stream));
  }
  return 0;
}

extern "C" int mt_mlp_train_steps(const MtMlpPlan* p, int step0, int nsteps, hipStream_t stream) {
  return mlp_enqueue_steps(p, step0, nsteps, stream, true);
}

// ------------------------------------------------------------------------------------------------
// Graph replay.  A block of `nsteps` consecutive steps is captured ONCE per plan into a hipGraph whose kernels
// take their step number as (immediate k) + (*counter); the last node adds nsteps to the counter.  Replaying the
// graph r times therefore runs steps [c, c + r*nsteps) of the epoch for one host call per block instead of one per
// kernel (about 11 launches per step otherwise), which is what lets several independent chains of a small pack be
// fed concurrently from one host thread.  Everything the kernels read besides the counter sits in descriptor
// tables that the engine rewrites in place between epochs, so a graph stays valid until the pack's membership
// changes.
__global__ void mt_counter_add_kernel(int* counter, int delta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *counter = *counter + delta;
}

struct MtStepGraph {
  hipGraph_t graph;
  hipGraphExec_t exec;
  int nsteps;
};

extern "C" void* mt_mlp_graph_create(const MtMlpPlan* p, int nsteps, int* counter, hipStream_t stream, int* err) {
  *err = 0;
  if (nsteps < 1 || counter == nullptr || stream == nullptr) { *err = -11; return nullptr; }
  if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { *err = -12; return nullptr; }
  mt_step_base_tls = counter;
  int rc = mlp_enqueue_steps(p, 0, nsteps, stream, false);     // the caller gathers step 0 of the epoch (mt_mlp_gather_at)
  mt_step_base_tls = nullptr;
  if (rc == 0) {
    hipLaunchKernelGGL(mt_counter_add_kernel, dim3(1), dim3(64), 0, stream, counter, nsteps);
    rc = (int)hipGetLastError();
  }
  hipGraph_t graph = nullptr;
  const hipError_t e_end = hipStreamEndCapture(stream, &graph);
  if (rc != 0 || e_end != hipSuccess || graph == nullptr) {
    if (graph != nullptr) (void)hipGraphDestroy(graph);
    *err = rc != 0 ? rc : -13;
    return nullptr;
  }
  hipGraphExec_t exec = nullptr;
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGraphDestroy(graph);
    *err = -14;
    return nullptr;
  }
  MtStepGraph* g = new MtStepGraph{graph, exec, nsteps};
  return g;
}

extern "C" int mt_mlp_graph_replay(void* handle, int times, hipStream_t stream) {
  MtStepGraph* g = static_cast<MtStepGraph*>(handle);
  if (g == nullptr) return -11;
  for (int i = 0; i < times; ++i)
    if (hipGraphLaunch(g->exec, stream) != hipSuccess) return -15;
  return 0;
}

extern "C" void mt_mlp_graph_destroy(void* handle) {
  MtStepGraph* g = static_cast<MtStepGraph*>(handle);
  if (g == nullptr) return;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
}

// Gather the mini-batch of step *counter (the first step of an epoch that is then driven by graph replays).
extern "C" int mt_mlp_gather_at(const MtMlpPlan* p, int* counter, hipStream_t stream) {
  mt_step_base_tls = counter;
  const int rc = plan_gather(p, 0, stream);
  mt_step_base_tls = nullptr;
  return rc;
}

// Direct launches of steps [*counter, *counter + nsteps) (the remainder of an epoch that is not a whole block).
extern "C" int mt_mlp_train_steps_at(const MtMlpPlan* p, int* counter, int nsteps, hipStream_t stream) {
  if (nsteps <= 0) return 0;
  mt_step_base_tls = counter;
  const int rc = mlp_enqueue_steps(p, 0, nsteps, stream, false);
  mt_step_base_tls = nullptr;
  if (rc != 0) return rc;
  hipLaunchKernelGGL(mt_counter_add_kernel, dim3(1), dim3(64), 0, stream, counter, nsteps);
  return (int)hipGetLastError();
}

// Validation sweep: the forward tables of an evaluation plan all read rows of ONE shared (bf16 or f32) copy of
// the validation set (layer 0's A pointer), chunk by chunk; the head kernel adds each chunk's error
// sums into the per-slot f64 metric table.
extern "C" int mt_mlp_eval(const MtMlpPlan* p, int n_val_rows, int chunk_rows, hipStream_t stream) {
  if (p->n_hidden < 1 || p->n_hidden > MT_MAX_LAYERS || chunk_rows <= 0) return -10;
  const int L = p->n_hidden;
  for (int r0 = 0; r0 < n_val_rows; r0 += chunk_rows) {
    const int rows = (n_val_rows - r0) < chunk_rows ? (n_val_rows - r0) : chunk_rows;
    for (int l = 0; l < L; ++l) MT_TRY(plan_forward(p, l, 0, l == 0 ? r0 : 0, rows, stream));
    MT_TRY((p->f32 ? mt_head_eval_f32_launch : mt_head_eval_launch)(p->head, p->active_ids, p->n_active, p->max_width, r0,
                                                                    rows, chunk_rows, p->head_scratch, stream));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// The per-round record table, produced on the device.  After a round (one training epoch + one validation sweep of every
// resident trial) the scheduler needs, per trial, {trial index, epoch, mean training loss, validation loss, validation
// MSE, validation MAPE}, and every rank needs every rank's rows.  mt_metric_table_kernel writes exactly the buffer that
// is all-gathered -- [capacity + 1][6] f64, empty slots as rows of -1, the last row {-2, seconds of the round, live
// trials, seconds of its training part, 0, 0} -- from the loss accumulators, the f64 metric sums and a small per-slot
// table of host-known numbers, so nothing passes through the host between the last validation kernel and the
// collective.  The round's duration comes from the GPU's constant-rate clock (100 MHz): mt_stamp_kernel stores it at
// the start of the round and after the training part.
struct MtTableMeta {
  double trial_index;
  double epoch;          // the epoch number this round completes
  int32_t steps;         // optimiser steps of the round (the training loss accumulator holds their sum)
  int32_t live;          // 0: empty slot
};

__global__ void mt_stamp_kernel(unsigned long long* slot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *slot = wall_clock64();
}

extern "C" int mt_stamp_launch(void* slot, hipStream_t stream) {
  hipLaunchKernelGGL(mt_stamp_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<unsigned long long*>(slot));
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void mt_metric_table_kernel(const MtTableMeta* __restrict__ meta,
                                                              const float* __restrict__ loss_acc,
                                                              const double* __restrict__ metrics, int capacity, int n_val,
                                                              int n_live, const unsigned long long* __restrict__ stamps,
                                                              double* __restrict__ table) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > capacity) return;
  double* row = table + (size_t)i * 6;
  if (i == capacity) {
    const unsigned long long now = wall_clock64();
    row[0] = -2.0;
    row[1] = (double)(now - stamps[0]) * 1e-8;               // 100 MHz ticks -> seconds
    row[2] = (double)n_live;
    row[3] = (double)(stamps[1] - stamps[0]) * 1e-8;
    row[4] = 0.0;
    row[5] = 0.0;
    return;
  }
  const MtTableMeta m = meta[i];
  if (!m.live) {
#pragma unroll
    for (int k = 0; k < 6; ++k) row[k] = -1.0;
    return;
  }
  const double nv = (double)n_val;
  row[0] = m.trial_index;
  row[1] = m.epoch;
  // the same f64 quotients the host path computes from the same accumulators: the two paths agree bit for bit
  row[2] = (double)loss_acc[i] / (double)(m.steps > 0 ? m.steps : 1);
  row[3] = metrics[i * 3 + 0] / nv;
  row[4] = metrics[i * 3 + 1] / nv;
  row[5] = 100.0 * metrics[i * 3 + 2] / nv;
}

extern "C" int mt_metric_table_launch(const void* meta, const float* loss_acc, const double* metrics, int capacity, int n_val,
                                      int n_live, const void* stamps, double* table, hipStream_t stream) {
  if (capacity <= 0 || n_val <= 0) return -2;
  hipLaunchKernelGGL(mt_metric_table_kernel, dim3((capacity + 1 + 255) / 256), dim3(256), 0, stream,
                     reinterpret_cast<const MtTableMeta*>(meta), loss_acc, metrics, capacity, n_val, n_live,
                     reinterpret_cast<const unsigned long long*>(stamps), table);
  return (int)hipGetLastError();
}

extern "C" int mt_table_meta_sizeof() { return (int)sizeof(MtTableMeta); }

// Digest of the source files this library was built from (ops/build.py passes it as -DMT_SRC_DIGEST).
#ifndef MT_SRC_DIGEST
#define MT_SRC_DIGEST ""
#endif
static const char mt_digest_string[] = "mt-src-digest:" MT_SRC_DIGEST;
extern "C" const char* mt_source_digest() { return mt_digest_string + 14; }

extern "C" int mt_abi_sizes(int* out) {
  out[0] = (int)sizeof(MtGemmGroup);
  out[1] = (int)sizeof(MtTile);
  out[2] = (int)sizeof(MtGatherSlot);
  out[3] = (int)sizeof(MtHeadSlot);
  out[4] = (int)sizeof(MtOptSlot);
  out[5] = (int)sizeof(MtMlpPlan);
  return 6;
}
