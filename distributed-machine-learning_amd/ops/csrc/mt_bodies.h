// Bodies of the two small per-step kernels that also run as extra workgroups inside a neighbouring launch
// (gfx950).  A training step of a small pack is a chain of dependent launches whose cost is mostly the launch
// boundaries; work that only has to finish "before the optimiser" (the head-gradient reduction) or "before the next
// forward" (the next step's mini-batch gather) rides along in a launch that is there anyway, chosen by block index.
#pragma once
#include "mt_common.h"

#define HL_ROWS 16       // rows per workgroup of the head/loss kernel = rows per partial record (= the rows of one MFMA
                         // row tile, so that the row-block training kernel writes the same records: rowblock.hip)

// Loss value and its derivative w.r.t. the prediction: 0 squared error, 1 absolute error, 2 Huber (beta 1),
// 3 absolute percentage error.
__device__ __forceinline__ void loss_and_grad(int type, float out, float y, float& l, float& g) {
  const float d = out - y;
  const float a = fabsf(d);
  const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
  switch (type) {
    case 0: l = d * d; g = 2.f * d; break;
    case 1: l = a; g = sgn; break;
    case 2: if (a < 1.f) { l = 0.5f * d * d; g = d; } else { l = a - 0.5f; g = sgn; } break;
    default: { const float den = fabsf(y) + 1e-8f; l = 100.f * a / den; g = 100.f * sgn / den; } break;
  }
}

struct MtHeadReduceArgs {      // by-value kernel argument of the launch that carries the head reduction
  const MtHeadSlot* slots;
  const int32_t* ids;
  const float* scratch;
  int32_t n_slots, max_chunks, rec;
};

// Mini-batch gather, workgroup (bx, by) of a grid (ceil(max_batch / rows_per_block), active slots), 256 threads.
// F32 = false: rows are written as bf16, one 16-byte chunk (8 features) per thread; F32 = true: as f32, one 16-byte
// chunk (4 features) per thread.  Fpad is a multiple of 8 either way.
template <bool F32>
__device__ __forceinline__ void mt_gather_body(const MtGatherSlot* __restrict__ slots, const int32_t* __restrict__ ids,
                                               const float* __restrict__ X, const float* __restrict__ Y, uint32_t n_rows,
                                               int F, int Fpad, int hb, int step, int bx, int by) {
  const MtGatherSlot s = slots[ids[by]];
  if (step >= s.active_steps) return;
  constexpr int EPC = F32 ? 4 : 8;           // elements per 16-byte output chunk
  const int cpr = Fpad / EPC;                // chunks per row
  const int rows_per_block = 256 / cpr;
  const int tid = threadIdx.x;
  const int r = tid / cpr, c = tid - r * cpr;
  if (r >= rows_per_block) return;
  const int i = bx * rows_per_block + r;
  if (i >= s.B) return;
  const uint32_t pos = (uint32_t)step * (uint32_t)s.B + (uint32_t)i;
  const uint32_t src = mt_permute(pos, n_rows, s.key, hb);
  const float* xr = X + (size_t)src * F;
  float v[EPC];
  const int f0 = c * EPC;
  if ((F & 3) == 0 && f0 + EPC <= F) {
#pragma unroll
    for (int q = 0; q < EPC / 4; ++q) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xr + f0 + 4 * q);
      v[4 * q] = a[0]; v[4 * q + 1] = a[1]; v[4 * q + 2] = a[2]; v[4 * q + 3] = a[3];
    }
  } else {
#pragma unroll
    for (int e = 0; e < EPC; ++e) v[e] = (f0 + e < F) ? xr[f0 + e] : 0.f;
  }
  if constexpr (F32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(s.X0) + (size_t)i * Fpad + f0) = f32x4{v[0], v[1], v[2], v[3]};
  } else {
    u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(s.X0) + (size_t)i * Fpad + f0) = o;
  }
  if (c == 0) s.y[i] = Y[src];
}

// Adds the partial records {gw[0..W), gb, loss} that the head/loss kernel wrote for slot ids[bx], in chunk order.
__device__ __forceinline__ void mt_head_reduce_body(const MtHeadSlot* __restrict__ slots, const int32_t* __restrict__ ids,
                                                    int step, const float* __restrict__ scratch, int max_chunks, int rec,
                                                    int bx) {
  const int slot = ids[bx];
  const MtHeadSlot s = slots[slot];
  if (step >= s.active_steps) return;
  const int nchunk = (s.B + HL_ROWS - 1) / HL_ROWS;
  const float* base = scratch + (size_t)slot * max_chunks * rec;
  for (int k = threadIdx.x; k < s.W + 2; k += 256) {
    const int col = (k < s.W) ? k : rec - 2 + (k - s.W);
    float acc = 0.f;
#pragma unroll 8
    for (int c = 0; c < nchunk; ++c) acc += base[(size_t)c * rec + col];     // independent loads, ordered adds
    if (k < s.W) s.gw[k] = acc;
    else if (k == s.W) s.gb[0] = acc;
    else s.loss_acc[0] += acc;
  }
}
