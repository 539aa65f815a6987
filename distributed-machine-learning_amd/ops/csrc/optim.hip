// Fused per-trial optimiser step over a pack's flat parameter slab (gfx950).
//
// Every trial slot owns a contiguous region of the f32 master-parameter slab P (and the same region
// of G, M, V and of the bf16 shadow Pb that the GEMMs read).  One launch updates all slots: block
// (x, s) handles elements [x*2048, x*2048 + 2048) of slot s with that slot's own optimiser family,
// learning rate (including the warm-up / decay factor of its own step count), weight decay, betas and
// clip scale, read from a small descriptor table -- so heterogeneous trials still cost one launch.
// The same pass refreshes the bf16 shadow; a separate "cast" pass never exists.  Pb = null (the exact-f32 engine, whose
// GEMMs read the f32 master parameters directly) skips the shadow: 24 bytes per parameter instead of 30.
//
// Global-norm clipping: mt_sqnorm_kernel writes one partial sum of squares per (slot, block); each
// optimiser block re-adds its slot's partials in index order (a fixed order, so bitwise reproducible)
// and scales gradients by min(1, clip / (norm + 1e-6)).  Slots of more than OPT_TOTAL_BLOCKS blocks get their sum from
// one workgroup of mt_sqnorm_total_kernel instead (also a fixed order).
#include "mt_optim_body.h"

__global__ __launch_bounds__(OPT_THREADS) void mt_sqnorm_kernel(const float* __restrict__ G, size_t slot_stride,
                                                                const MtOptSlot* __restrict__ hp,
                                                                const int32_t* __restrict__ ids,
                                                                float* __restrict__ partials, int max_blocks, int step_imm,
                                                                int gsplits, size_t gsplit_stride,
                                                                const int* __restrict__ step_base) {
  const int step = mt_step(step_imm, step_base);
  const int slot = ids[blockIdx.y];
  const MtOptSlot h = hp[slot];
  float acc = 0.f;
  const int base = blockIdx.x * OPT_PER_BLOCK + threadIdx.x * 8;
  if (h.first_splits > 0 && base < h.first_end) gsplits = h.first_splits;
  if (step < h.active_steps && h.clip > 0.f && base < h.n_params) {
    const float* g = G + (size_t)slot * slot_stride + base;
    const int n = (h.n_params - base) < 8 ? (h.n_params - base) : 8;
    for (int e = 0; e < n; ++e) {
      float v = g[e];
      for (int sp = 1; sp < gsplits; ++sp) v += g[(size_t)sp * gsplit_stride + e];   // same order as the update
      acc += v * v;
    }
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[(size_t)slot * max_blocks + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Sum of a big slot's partials, in a fixed order (256 consecutive runs added one after the other, then the 256 run sums
// in index order), left in the slot's first partial.  Slots of at most OPT_TOTAL_BLOCKS blocks are not touched: their
// optimiser blocks add the partials themselves.
__global__ __launch_bounds__(OPT_THREADS) void mt_sqnorm_total_kernel(const MtOptSlot* __restrict__ hp,
                                                                      const int32_t* __restrict__ ids,
                                                                      float* __restrict__ partials, int max_blocks,
                                                                      int step_imm, const int* __restrict__ step_base) {
  const int step = mt_step(step_imm, step_base);
  const int slot = ids[blockIdx.x];
  const MtOptSlot h = hp[slot];
  const int nb = (h.n_params + OPT_PER_BLOCK - 1) / OPT_PER_BLOCK;
  if (step >= h.active_steps || h.clip <= 0.f || nb <= OPT_TOTAL_BLOCKS) return;
  __shared__ float run[OPT_THREADS];
  const int per = (nb + OPT_THREADS - 1) / OPT_THREADS;
  const int i0 = threadIdx.x * per, i1 = (i0 + per < nb) ? i0 + per : nb;
  float a = 0.f;
  for (int i = i0; i < i1; ++i) a += partials[(size_t)slot * max_blocks + i];
  run[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int t = 0; t < OPT_THREADS; ++t) tot += run[t];
    partials[(size_t)slot * max_blocks] = tot;
  }
}

template <bool G_BF16>
__global__ __launch_bounds__(OPT_THREADS) void mt_optim_kernel(float* __restrict__ P, const float* __restrict__ G,
                                                               float* __restrict__ M, float* __restrict__ V,
                                                               uint16_t* __restrict__ Pb, size_t slot_stride,
                                                               const MtOptSlot* __restrict__ hp,
                                                               const int32_t* __restrict__ ids,
                                                               const float* __restrict__ partials, int max_blocks,
                                                               int step_imm, int gsplits, size_t gsplit_stride,
                                                               const int* __restrict__ step_base) {
  optim_body<G_BF16>(P, G, M, V, Pb, slot_stride, hp, ids, partials, max_blocks, mt_step(step_imm, step_base), gsplits,
                     gsplit_stride, blockIdx.x, blockIdx.y);
}

// Optimiser of step t and mini-batch gather of step t + 1 in one launch: the gather only needs step t's forward and
// weight-gradient kernels to be done with the input buffers (they are: stream order), so its workgroups ride behind the
// optimiser's in the grid's x dimension.
struct MtGatherArgs {
  const MtGatherSlot* slots;
  const float* X;
  const float* Y;
  uint32_t n_rows;
  int32_t F, Fpad, hb, blocks;      // blocks: gather workgroups per slot (0 = no gather in this launch)
};

template <bool F32>      // F32: the rider gathers f32 rows (exact-f32 engine) instead of bf16 rows
__global__ __launch_bounds__(OPT_THREADS) void mt_optim_gather_kernel(float* __restrict__ P, const float* __restrict__ G,
                                                                      float* __restrict__ M, float* __restrict__ V,
                                                                      uint16_t* __restrict__ Pb, size_t slot_stride,
                                                                      const MtOptSlot* __restrict__ hp,
                                                                      const int32_t* __restrict__ ids,
                                                                      const float* __restrict__ partials, int max_blocks,
                                                                      int step_imm, int gsplits, size_t gsplit_stride,
                                                                      const int* __restrict__ step_base, int nb_opt,
                                                                      MtGatherArgs ga) {
  const int step = mt_step(step_imm, step_base);
  if ((int)blockIdx.x < nb_opt)
    optim_body<false>(P, G, M, V, Pb, slot_stride, hp, ids, partials, max_blocks, step, gsplits, gsplit_stride, blockIdx.x,
                      blockIdx.y);
  else
    mt_gather_body<F32>(ga.slots, ids, ga.X, ga.Y, ga.n_rows, ga.F, ga.Fpad, ga.hb, step + 1, (int)blockIdx.x - nb_opt,
                   blockIdx.y);
}

extern "C" int mt_optim_gather_launch(float* P, const float* G, float* M, float* V, void* Pb, size_t slot_stride,
                                      const MtOptSlot* hp, const int32_t* ids, int n_active, int max_params,
                                      float* partials, int max_blocks, int use_clip, int step, int gsplits,
                                      size_t gsplit_stride, const MtGatherSlot* gslots, const float* X, const float* Y,
                                      uint32_t n_rows, int F, int Fpad, int hb, int max_batch, hipStream_t stream) {
  if (gsplits < 1) return -6;
  if (n_active <= 0 || max_params <= 0) return 0;
  if (slot_stride % 8 != 0) return -2;
  const bool f32 = (Pb == nullptr);       // no bf16 shadow: the pack trains in exact f32 and gathers f32 rows
  const int cpr = f32 ? Fpad / 4 : Fpad / 8;
  if (Fpad % 8 != 0 || cpr > 256 || Fpad < F) return -2;
  const int nb = (max_params + OPT_PER_BLOCK - 1) / OPT_PER_BLOCK;
  if (nb > max_blocks) return -4;
  const int rows_per_block = 256 / cpr;
  MtGatherArgs ga = {gslots, X, Y, n_rows, F, Fpad, hb, (max_batch + rows_per_block - 1) / rows_per_block};
  dim3 block(OPT_THREADS);
  if (use_clip) {
    if (partials == nullptr) return -3;
    hipLaunchKernelGGL(mt_sqnorm_kernel, dim3(nb, n_active), block, 0, stream, G, slot_stride, hp, ids, partials, max_blocks,
                       step, gsplits, gsplit_stride, mt_step_base_tls);
    if (nb > OPT_TOTAL_BLOCKS)
      hipLaunchKernelGGL(mt_sqnorm_total_kernel, dim3(n_active), block, 0, stream, hp, ids, partials, max_blocks, step,
                         mt_step_base_tls);
  }
  if (f32)
    hipLaunchKernelGGL(mt_optim_gather_kernel<true>, dim3(nb + ga.blocks, n_active), block, 0, stream, P, G, M, V,
                       (uint16_t*)nullptr, slot_stride, hp, ids, use_clip ? partials : nullptr, max_blocks, step, gsplits,
                       gsplit_stride, mt_step_base_tls, nb, ga);
  else
    hipLaunchKernelGGL(mt_optim_gather_kernel<false>, dim3(nb + ga.blocks, n_active), block, 0, stream, P, G, M, V,
                       reinterpret_cast<uint16_t*>(Pb), slot_stride, hp, ids, use_clip ? partials : nullptr, max_blocks,
                       step, gsplits, gsplit_stride, mt_step_base_tls, nb, ga);
  return (int)hipGetLastError();
}

extern "C" int mt_optim_launch(float* P, const float* G, float* M, float* V, void* Pb, size_t slot_stride,
                               const MtOptSlot* hp, const int32_t* ids, int n_active, int max_params,
                               float* partials, int max_blocks, int use_clip, int step, int gsplits,
                               size_t gsplit_stride, hipStream_t stream) {
  const int n_slots = n_active;
  if (gsplits < 1) return -6;
  if (n_slots <= 0 || max_params <= 0) return 0;
  if (slot_stride % 8 != 0) return -2;
  const int nb = (max_params + OPT_PER_BLOCK - 1) / OPT_PER_BLOCK;
  if (nb > max_blocks) return -4;     // partials holds max_blocks floats per physical slot
  dim3 grid(nb, n_slots), block(OPT_THREADS);
  if (use_clip) {
    if (partials == nullptr) return -3;
    hipLaunchKernelGGL(mt_sqnorm_kernel, grid, block, 0, stream, G, slot_stride, hp, ids, partials, max_blocks, step, gsplits,
                       gsplit_stride, mt_step_base_tls);
    if (nb > OPT_TOTAL_BLOCKS)
      hipLaunchKernelGGL(mt_sqnorm_total_kernel, dim3(n_slots), block, 0, stream, hp, ids, partials, max_blocks, step,
                         mt_step_base_tls);
  }
  hipLaunchKernelGGL(mt_optim_kernel<false>, grid, block, 0, stream, P, G, M, V, reinterpret_cast<uint16_t*>(Pb),
                     slot_stride, hp, ids, use_clip ? partials : nullptr, max_blocks, step, gsplits, gsplit_stride,
                     mt_step_base_tls);
  return (int)hipGetLastError();
}

// Same update with bf16 gradients and no clipping pass (the caller clips, if at all, before the all-reduce).
extern "C" int mt_optim_launch_bf16g(float* P, const void* Gb, float* M, float* V, void* Pb, size_t slot_stride,
                                     const MtOptSlot* hp, const int32_t* ids, int n_active, int max_params, int step,
                                     hipStream_t stream) {
  if (n_active <= 0 || max_params <= 0) return 0;
  if (slot_stride % 8 != 0) return -2;
  const int nb = (max_params + OPT_PER_BLOCK - 1) / OPT_PER_BLOCK;
  dim3 grid(nb, n_active), block(OPT_THREADS);
  hipLaunchKernelGGL(mt_optim_kernel<true>, grid, block, 0, stream, P, reinterpret_cast<const float*>(Gb), M, V,
                     reinterpret_cast<uint16_t*>(Pb), slot_stride, hp, ids, nullptr, 0, step, 1, (size_t)0, mt_step_base_tls);
  return (int)hipGetLastError();
}
