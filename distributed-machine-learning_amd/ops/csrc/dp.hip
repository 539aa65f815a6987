// Step driver of a data-parallel MLP trial (gfx950): one oversized trial, every rank of the node holds a shard of the
// table and a replica of the parameters, gradients are averaged with RCCL all-reduce over xGMI.
//
// The whole optimiser step is enqueued from here with no interpreter in the loop and no host synchronisation:
//   gather (keyed permutation of this rank's shard, bf16 rows) -> L x forward GEMM(+bias+ReLU) -> head + loss + dH
//   -> bucket "head": f32 -> bf16, all-reduce
//   -> for l = L-1..0 { weight gradient of layer l (split-K slabs, + bias gradient); slabs summed and rounded to the
//        layer's bf16 bucket; the bucket's all-reduce starts on the communication stream; input gradient of layer l
//        (+ReLU mask) runs on the compute stream meanwhile }
//   -> compute stream waits for the last all-reduce -> fused optimiser on bf16 gradients (+ bf16 weight shadow).
// All GEMMs are the tile kernels of gemm.hip (mt_gemm_single_launch / mt_wgrad_single_launch).  Buckets are one layer
// each: a width-8192 layer is 134 MB in bf16 -- on 7 xGMI links of ~153 GB/s a ring all-reduce of that size takes about
// as long as the layer's input-gradient GEMM it hides behind.  bf16 buckets halve the bytes on the links; the master
// parameters and both optimiser moments stay f32.
//
// Exact-f32 form (plan.f32 = 1, selected by the trial's dtype): the table, activations, gradients and buckets are f32, every
// GEMM runs on the exact-f32 tile kernels of gemm_f32.hip, the split-K slabs of a layer are added in index order straight
// into the layer's f32 bucket (which is what is all-reduced and what the optimiser reads), and there is no shadow.  Twice
// the bytes on the links and a matrix instruction sixteen times slower -- the form for checking a result, or for a trial
// whose loss surface does not survive bf16 gradients -- with the same launch sequence and the same overlap.
//
// The collective is called through a function pointer handed in by the Python side (parallel/rccl.py resolves
// ncclAllReduce from the RCCL library already loaded into the process and owns the communicator), so this library has
// no link-time dependency on RCCL; with one rank no collective is issued at all.
#include "mt_bodies.h"


extern "C" {
int mt_gemm_single_launch(int mode, const void* A, const void* B, void* C, const void* aux, int M, int N, int K, int lda,
                          int ldb, int ldc, int ldaux, int flags, hipStream_t stream);
int mt_wgrad_single_launch(const void* dY, const void* X, float* C, float* db, int M, int out, int in, int ksplits,
                           size_t split_stride, hipStream_t stream);
int mt_head_loss_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_batch, int max_width, int step,
                        float* scratch, hipStream_t stream);
int mt_head_eval_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_width, int row0, int nrows,
                        int max_rows, float* scratch, hipStream_t stream);
int mt_gemm_f32_single_launch(int mode, const void* A, const void* B, void* C, const void* aux, void* aux_out, int I, int J,
                              int KK, int lda, int ldb, int ldc, int flags, hipStream_t stream);
int mt_wgrad_f32_single_launch(const void* dY, const void* X, float* arena, int M, int I, int J, int ksplits, size_t slab,
                               int with_bias, hipStream_t stream);
int mt_slab_sum_launch(const float* arena, float* out, long n, int ksplits, size_t slab, hipStream_t stream);
int mt_head_loss_f32_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_batch, int max_width, int step,
                            float* scratch, int with_reduce, hipStream_t stream);
int mt_head_eval_f32_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_width, int row0, int nrows,
                            int max_rows, float* scratch, hipStream_t stream);
int mt_optim_launch(float* P, const float* G, float* M, float* V, void* Pb, size_t slot_stride, const MtOptSlot* hp,
                    const int32_t* ids, int n_active, int max_params, float* partials, int max_blocks, int use_clip, int step,
                    int gsplits, size_t gsplit_stride, hipStream_t stream);
int mt_optim_launch_bf16g(float* P, const void* Gb, float* M, float* V, void* Pb, size_t slot_stride, const MtOptSlot* hp,
                          const int32_t* ids, int n_active, int max_params, int step, hipStream_t stream);
}

// ncclAllReduce(sendbuff, recvbuff, count, datatype, op, comm, stream)
typedef int (*MtAllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);

struct MtDpPlan {
  int32_t n_hidden;            // hidden layers; the scalar head is layer index n_hidden in the offset tables
  int32_t batch;               // rows per step on THIS rank
  int32_t F;                   // features (multiple of 8: 16-byte rows in either element type)
  int32_t hb;                  // half bits of the Feistel permutation over n_rows
  uint32_t n_rows;             // rows of this rank's shard
  int32_t world;
  int32_t f32;                 // 1: exact-f32 form (f32 table, activations, buckets; no shadow), 0: bf16 form
  int32_t use_clip;            // f32 form: clip the averaged gradient to the slot's global-norm bound in the optimiser launch
  int32_t clip_blocks;         // floats in clip_partials
  int32_t pad0;
  int32_t widths[MT_MAX_LAYERS];
  int32_t ksplits[MT_MAX_LAYERS];
  int64_t w_off[MT_MAX_LAYERS + 1], b_off[MT_MAX_LAYERS + 1];        // element offsets into the flat slabs
  int64_t bucket_lo[MT_MAX_LAYERS + 1], bucket_hi[MT_MAX_LAYERS + 1];
  int64_t numel;               // elements of one slab (a multiple of 8)
  const void* X;               // bf16 (f32 form: f32) [n_rows][F]
  const float* Y;              // f32 [n_rows]
  void* X0;                    // [batch][F], element type of X
  float* ybuf;                 // f32 [batch]
  void* H[MT_MAX_LAYERS];      // [batch][width_l], element type of X
  void* dZ[MT_MAX_LAYERS];     // [batch][width_l], element type of X
  float* P;                    // f32 master parameters
  void* W;                     // bf16 shadow (what the GEMMs read); f32 form: unused, the GEMMs read P
  float* G32;                  // f32 gradient staging: max(ksplits) slabs of numel
  void* Gb;                    // gradient buckets (what is all-reduced and what the optimiser reads): bf16, f32 form: f32
  void* Gr;                    // where the all-reduce delivers and what the optimiser reads: Gb itself (in place) unless the
                               // caller wants the one-rank rehearsal to move bytes (a one-rank in-place all-reduce is a no-op)
  float* M;
  float* V;
  const MtHeadSlot* head;      // one slot
  const MtOptSlot* opt;        // one slot
  const int32_t* ids;          // {0}
  float* head_scratch;
  MtAllReduceFn all_reduce;    // null or world == 1: no collective
  void* comm;
  int32_t nccl_bf16, nccl_avg; // enum values of the RCCL build in use
  int32_t nccl_f32, pad1;
  float* clip_partials;        // f32 form with use_clip: per-block squared norms
  hipStream_t comm_stream;
  hipEvent_t ready[MT_MAX_LAYERS + 1];   // bucket l is complete on the compute stream
  hipEvent_t reduced;                    // every bucket of the step has been reduced
};

// ---- keyed gather of bf16 rows ------------------------------------------------------------------------------------
// X / X0 are addressed in 16-byte chunks: cpr chunks per row (F / 8 bf16 or F / 4 f32 elements each).
__global__ __launch_bounds__(256) void mt_dp_gather_kernel(const u32x4* __restrict__ X, const float* __restrict__ Y,
                                                           u32x4* __restrict__ X0, float* __restrict__ y, uint32_t n_rows,
                                                           int cpr, int hb, int batch, int step, uint32_t k0, uint32_t k1,
                                                           uint32_t k2, uint32_t k3) {
  const int rows_per_block = 256 / cpr;
  const int tid = threadIdx.x;
  const int r = tid / cpr, c = tid - r * cpr;
  if (r >= rows_per_block) return;
  const int i = blockIdx.x * rows_per_block + r;
  if (i >= batch) return;
  const uint32_t key[4] = {k0, k1, k2, k3};
  const uint32_t src = mt_permute((uint32_t)step * (uint32_t)batch + (uint32_t)i, n_rows, key, hb);
  X0[(size_t)i * cpr + c] = X[(size_t)src * cpr + c];
  if (c == 0) y[i] = Y[src];
}

// ---- bucket: ordered sum of the f32 split slabs, rounded once to bf16 ---------------------------------------------------
__global__ __launch_bounds__(256) void mt_slab_cast_kernel(const float* __restrict__ G32, size_t stride, int ksplits,
                                                           uint16_t* __restrict__ Gb, size_t lo, size_t hi) {
  const size_t i = lo + ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= hi) return;                        // lo, hi are multiples of 8
  f32x4 a = *reinterpret_cast<const f32x4*>(G32 + i), b = *reinterpret_cast<const f32x4*>(G32 + i + 4);
  for (int s = 1; s < ksplits; ++s) {
    a += *reinterpret_cast<const f32x4*>(G32 + (size_t)s * stride + i);
    b += *reinterpret_cast<const f32x4*>(G32 + (size_t)s * stride + i + 4);
  }
  *reinterpret_cast<u32x4*>(Gb + i) =
      u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
}

#define DP_TRY(x)            \
  do {                       \
    int _e = (x);            \
    if (_e != 0) return _e;  \
  } while (0)

static int dp_bucket(const MtDpPlan* p, int l, int ksplits, hipStream_t stream) {
  const size_t lo = (size_t)p->bucket_lo[l], hi = (size_t)p->bucket_hi[l];
  if (hi <= lo) return 0;
  if ((lo & 7) || (hi & 7)) return -21;
  if (p->f32) {
    float* buf = reinterpret_cast<float*>(p->Gb) + lo;
    if (mt_slab_sum_launch(p->G32 + lo, buf, (long)(hi - lo), ksplits, (size_t)p->numel, stream) != 0) return -22;
    if (p->world > 1 && p->all_reduce != nullptr) {
      if (hipEventRecord(p->ready[l], stream) != hipSuccess) return -23;
      if (hipStreamWaitEvent(p->comm_stream, p->ready[l], 0) != hipSuccess) return -23;
      if (p->all_reduce(buf, reinterpret_cast<float*>(p->Gr) + lo, hi - lo, p->nccl_f32, p->nccl_avg, p->comm,
                        p->comm_stream) != 0)
        return -24;
    }
    return 0;
  }
  const size_t chunks = (hi - lo) / 8;
  hipLaunchKernelGGL(mt_slab_cast_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream, p->G32,
                     (size_t)p->numel, ksplits, reinterpret_cast<uint16_t*>(p->Gb), lo, hi);
  if (hipGetLastError() != hipSuccess) return -22;
  if (p->world > 1 && p->all_reduce != nullptr) {
    // the bucket's all-reduce runs on the communication stream, behind the kernel that completed the bucket
    if (hipEventRecord(p->ready[l], stream) != hipSuccess) return -23;
    if (hipStreamWaitEvent(p->comm_stream, p->ready[l], 0) != hipSuccess) return -23;
    uint16_t* buf = reinterpret_cast<uint16_t*>(p->Gb) + lo;
    if (p->all_reduce(buf, reinterpret_cast<uint16_t*>(p->Gr) + lo, hi - lo, p->nccl_bf16, p->nccl_avg, p->comm,
                      p->comm_stream) != 0)
      return -24;
  }
  return 0;
}

static int dp_forward(const MtDpPlan* p, const void* X0, int rows, hipStream_t stream) {
  const uint16_t* W = reinterpret_cast<const uint16_t*>(p->W);
  const void* a = X0;
  int k = p->F;
  for (int l = 0; l < p->n_hidden; ++l) {
    const int w = p->widths[l];
    if (p->f32)
      DP_TRY(mt_gemm_f32_single_launch(0, a, p->P + p->w_off[l], p->H[l], p->P + p->b_off[l], nullptr, rows, w, k, k, k, w,
                                       MT_GEMM_FLAG_RELU, stream));
    else
      DP_TRY(mt_gemm_single_launch(0, a, W + p->w_off[l], p->H[l], p->P + p->b_off[l], rows, w, k, k, k, w, 0,
                                   MT_GEMM_FLAG_RELU, stream));
    a = p->H[l];
    k = w;
  }
  return 0;
}

// key: the four Feistel round keys of (seed, trial, rank, epoch); opt_step: optimiser steps taken before step0.
extern "C" int mt_dp_train_steps(const MtDpPlan* p, const uint32_t* key, int step0, int nsteps, int opt_step,
                                 hipStream_t stream) {
  const int L = p->n_hidden;
  const int cpr = p->f32 ? p->F / 4 : p->F / 8;               // 16-byte chunks per table row
  if (L < 1 || L > MT_MAX_LAYERS || p->batch <= 0 || (p->F & 7) || cpr > 256) return -10;
  for (int l = 0; l < L; ++l)
    if (p->widths[l] & 7) return -10;
  if (p->f32 && p->use_clip && p->clip_partials == nullptr) return -10;
  if (p->Gr == nullptr) return -10;
  const uint16_t* W = reinterpret_cast<const uint16_t*>(p->W);
  const int rows_per_block = 256 / cpr;
  for (int s = 0; s < nsteps; ++s) {
    const int step = step0 + s;
    hipLaunchKernelGGL(mt_dp_gather_kernel, dim3((p->batch + rows_per_block - 1) / rows_per_block), dim3(256), 0, stream,
                       reinterpret_cast<const u32x4*>(p->X), p->Y, reinterpret_cast<u32x4*>(p->X0), p->ybuf, p->n_rows,
                       cpr, p->hb, p->batch, step, key[0], key[1], key[2], key[3]);
    if (hipGetLastError() != hipSuccess) return -20;
    DP_TRY(dp_forward(p, p->X0, p->batch, stream));
    if (p->f32)
      DP_TRY(mt_head_loss_f32_launch(p->head, p->ids, 1, p->batch, p->widths[L - 1], 0, p->head_scratch, 1, stream));
    else
      DP_TRY(mt_head_loss_launch(p->head, p->ids, 1, p->batch, p->widths[L - 1], 0, p->head_scratch, stream));
    DP_TRY(dp_bucket(p, L, 1, stream));
    for (int l = L - 1; l >= 0; --l) {
      const int w = p->widths[l], k = l > 0 ? p->widths[l - 1] : p->F;
      const void* a_in = l > 0 ? p->H[l - 1] : p->X0;
      if (p->f32) {
        // weights and bias of a layer are adjacent in the slab (w * k is a multiple of 8), which is the arena layout of
        // the f32 weight-gradient launch: dW at the slice start, db right behind it
        if (p->b_off[l] != p->w_off[l] + (int64_t)w * k) return -25;
        DP_TRY(mt_wgrad_f32_single_launch(p->dZ[l], a_in, p->G32 + p->w_off[l], p->batch, w, k, p->ksplits[l],
                                          (size_t)p->numel, 1, stream));
      } else {
        DP_TRY(mt_wgrad_single_launch(p->dZ[l], a_in, p->G32 + p->w_off[l], p->G32 + p->b_off[l], p->batch, w, k,
                                      p->ksplits[l], (size_t)p->numel, stream));
      }
      DP_TRY(dp_bucket(p, l, p->ksplits[l], stream));
      if (l > 0) {
        if (p->f32)
          DP_TRY(mt_gemm_f32_single_launch(1, p->dZ[l], p->P + p->w_off[l], p->dZ[l - 1], p->H[l - 1], nullptr, p->batch, k, w,
                                           w, k, k, MT_GEMM_FLAG_MASK, stream));
        else
          DP_TRY(mt_gemm_single_launch(1, p->dZ[l], W + p->w_off[l], p->dZ[l - 1], p->H[l - 1], p->batch, k, w, w, k, k, k,
                                       MT_GEMM_FLAG_MASK, stream));
      }
    }
    if (p->world > 1 && p->all_reduce != nullptr) {
      if (hipEventRecord(p->reduced, p->comm_stream) != hipSuccess) return -23;
      if (hipStreamWaitEvent(stream, p->reduced, 0) != hipSuccess) return -23;
    }
    if (p->f32)
      DP_TRY(mt_optim_launch(p->P, reinterpret_cast<const float*>(p->Gr), p->M, p->V, nullptr, (size_t)p->numel, p->opt, p->ids,
                             1, (int)p->numel, p->clip_partials, p->use_clip ? p->clip_blocks : (1 << 30), p->use_clip,
                             opt_step + s, 1, (size_t)0, stream));
    else
      DP_TRY(mt_optim_launch_bf16g(p->P, p->Gr, p->M, p->V, p->W, (size_t)p->numel, p->opt, p->ids, 1, (int)p->numel,
                                   opt_step + s, stream));
  }
  return 0;
}

// Validation: forward over rows [0, n) of a bf16 table in chunks of at most `chunk` rows (<= plan batch capacity given by
// the activation buffers); the head kernel adds each chunk's error sums into the slot's f64 metric table.
extern "C" int mt_dp_eval(const MtDpPlan* p, const void* Xval, int n, int chunk, hipStream_t stream) {
  const int L = p->n_hidden;
  if (L < 1 || L > MT_MAX_LAYERS || chunk <= 0) return -10;
  for (int r0 = 0; r0 < n; r0 += chunk) {
    const int rows = (n - r0) < chunk ? (n - r0) : chunk;
    const void* x = p->f32 ? (const void*)(reinterpret_cast<const float*>(Xval) + (size_t)r0 * p->F)
                           : (const void*)(reinterpret_cast<const uint16_t*>(Xval) + (size_t)r0 * p->F);
    DP_TRY(dp_forward(p, x, rows, stream));
    if (p->f32)
      DP_TRY(mt_head_eval_f32_launch(p->head, p->ids, 1, p->widths[L - 1], r0, rows, chunk, p->head_scratch, stream));
    else
      DP_TRY(mt_head_eval_launch(p->head, p->ids, 1, p->widths[L - 1], r0, rows, chunk, p->head_scratch, stream));
  }
  return 0;
}

extern "C" int mt_dp_plan_sizeof() { return (int)sizeof(MtDpPlan); }
