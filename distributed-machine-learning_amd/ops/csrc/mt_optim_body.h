// Per-slot fused optimiser update of one 2048-element block of a slot's slab region (gfx950); shared by the optimiser
// launches (optim.hip) and the resident epoch kernel (resident.hip).
#pragma once
#include "mt_bodies.h"

#define OPT_THREADS 256
#define OPT_PER_BLOCK 2048   // 8 elements per thread
// A slot of more than this many blocks (a million parameters) has its clipping norm added up once, by
// mt_sqnorm_total_kernel, instead of by every one of its optimiser blocks: the per-block loop is nb loads long, and a
// 38-million-parameter Transformer trial has 18 thousand blocks.
#define OPT_TOTAL_BLOCKS 512

__device__ __forceinline__ float lr_factor(int t, int warmup, int total) {
  if (warmup > 0 && t < warmup) return (float)(t + 1) / (float)warmup;
  if (total > 0) {
    const int span = (total - warmup) > 1 ? (total - warmup) : 1;
    const float f = (float)(total - t) / (float)span;
    return f > 0.f ? f : 0.f;
  }
  return 1.f;
}

template <bool G_BF16>   // G_BF16: gradients are bf16 (data-parallel trials all-reduce bf16 gradient buckets)
__device__ __forceinline__ void optim_body(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ M,
                                           float* __restrict__ V, uint16_t* __restrict__ Pb, size_t slot_stride,
                                           const MtOptSlot* __restrict__ hp, const int32_t* __restrict__ ids,
                                           const float* __restrict__ partials, int max_blocks, int step, int gsplits,
                                           size_t gsplit_stride, int bx, int by) {
  const int slot = ids[by];
  const MtOptSlot h = hp[slot];
  if (step >= h.active_steps) return;
  const int base = bx * OPT_PER_BLOCK + threadIdx.x * 8;
  if (bx * OPT_PER_BLOCK >= h.n_params) return;

  float gscale = 1.f;
  if (h.clip > 0.f && partials != nullptr) {
    const int nb = (h.n_params + OPT_PER_BLOCK - 1) / OPT_PER_BLOCK;
    float tot = 0.f;
    if (nb <= OPT_TOTAL_BLOCKS) {
      for (int i = 0; i < nb; ++i) tot += partials[(size_t)slot * max_blocks + i];   // uniform -> scalar loads
    } else {
      tot = partials[(size_t)slot * max_blocks];      // mt_sqnorm_total_kernel left the slot's sum here
    }
    const float c = h.clip / (sqrtf(tot) + 1e-6f);
    gscale = c < 1.f ? c : 1.f;
  }
  if (base >= h.n_params) return;
  // layer 0 (tiny contraction tiles, few of them) may split its weight gradient further than the other layers
  if (h.first_splits > 0 && base < h.first_end) gsplits = h.first_splits;

  const int t = h.step_base + step;                 // optimiser steps already taken
  const float lr = h.lr * lr_factor(t, h.warmup, h.total);
  const float bc1 = 1.f - powf(h.beta1, (float)(t + 1));
  const float bc2 = 1.f - powf(h.beta2, (float)(t + 1));
  const float inv_bc2_sqrt = 1.f / sqrtf(bc2);
  const size_t off = (size_t)slot * slot_stride + base;
  const int n = (h.n_params - base) < 8 ? (h.n_params - base) : 8;

  float p[8], g[8], m[8], v[8];
  if (n == 8) {
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(P + off), p1 = *reinterpret_cast<const f32x4*>(P + off + 4);
    f32x4 g0, g1;
    if (G_BF16) {
      const u32x4 gb = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(G) + off);
      g0 = f32x4{bf16_bits_to_f32(gb[0] & 0xFFFFu), bf16_bits_to_f32(gb[0] >> 16), bf16_bits_to_f32(gb[1] & 0xFFFFu),
                 bf16_bits_to_f32(gb[1] >> 16)};
      g1 = f32x4{bf16_bits_to_f32(gb[2] & 0xFFFFu), bf16_bits_to_f32(gb[2] >> 16), bf16_bits_to_f32(gb[3] & 0xFFFFu),
                 bf16_bits_to_f32(gb[3] >> 16)};
    } else {
      g0 = *reinterpret_cast<const f32x4*>(G + off);
      g1 = *reinterpret_cast<const f32x4*>(G + off + 4);
      for (int sp = 1; sp < gsplits; ++sp) {      // split-K weight gradients: add the partial slabs in order
        g0 += *reinterpret_cast<const f32x4*>(G + (size_t)sp * gsplit_stride + off);
        g1 += *reinterpret_cast<const f32x4*>(G + (size_t)sp * gsplit_stride + off + 4);
      }
    }
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(M + off), m1 = *reinterpret_cast<const f32x4*>(M + off + 4);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(V + off), v1 = *reinterpret_cast<const f32x4*>(V + off + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p[e] = p0[e]; p[e + 4] = p1[e]; g[e] = g0[e]; g[e + 4] = g1[e];
      m[e] = m0[e]; m[e + 4] = m1[e]; v[e] = v0[e]; v[e + 4] = v1[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool ok = e < n;
      p[e] = ok ? P[off + e] : 0.f;
      g[e] = !ok ? 0.f : (G_BF16 ? bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(G)[off + e]) : G[off + e]);
      if (ok && !G_BF16)
        for (int sp = 1; sp < gsplits; ++sp) g[e] += G[(size_t)sp * gsplit_stride + off + e];
      m[e] = ok ? M[off + e] : 0.f; v[e] = ok ? V[off + e] : 0.f;
    }
  }

#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float gg = g[e] * gscale;
    switch (h.opt_type) {
      case 0:   // Adam, L2 regularisation folded into the gradient
      case 1: { // AdamW, decoupled decay
        if (h.opt_type == 0) gg += h.wd * p[e];
        else p[e] *= (1.f - lr * h.wd);
        m[e] = h.beta1 * m[e] + (1.f - h.beta1) * gg;
        v[e] = h.beta2 * v[e] + (1.f - h.beta2) * gg * gg;
        const float denom = sqrtf(v[e]) * inv_bc2_sqrt + h.eps;
        p[e] -= (lr / bc1) * (m[e] / denom);
      } break;
      case 2: { // SGD with momentum (buffer = M)
        gg += h.wd * p[e];
        m[e] = h.momentum * m[e] + gg;
        p[e] -= lr * (h.momentum > 0.f ? m[e] : gg);
      } break;
      default: { // RMSprop, alpha = 0.99 (V = square average, M = momentum buffer)
        gg += h.wd * p[e];
        v[e] = 0.99f * v[e] + 0.01f * gg * gg;
        const float avg = sqrtf(v[e]) + h.eps;
        if (h.momentum > 0.f) { m[e] = h.momentum * m[e] + gg / avg; p[e] -= lr * m[e]; }
        else p[e] -= lr * gg / avg;
      } break;
    }
  }

  if (n == 8) {
    *reinterpret_cast<f32x4*>(P + off) = f32x4{p[0], p[1], p[2], p[3]};
    *reinterpret_cast<f32x4*>(P + off + 4) = f32x4{p[4], p[5], p[6], p[7]};
    *reinterpret_cast<f32x4*>(M + off) = f32x4{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<f32x4*>(M + off + 4) = f32x4{m[4], m[5], m[6], m[7]};
    *reinterpret_cast<f32x4*>(V + off) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(V + off + 4) = f32x4{v[4], v[5], v[6], v[7]};
    if (Pb != nullptr)
      *reinterpret_cast<u32x4*>(Pb + off) = u32x4{pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]),
                                                  pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7])};
  } else {
    for (int e = 0; e < n; ++e) {
      P[off + e] = p[e]; M[off + e] = m[e]; V[off + e] = v[e];
      if (Pb != nullptr) Pb[off + e] = (uint16_t)f32_to_bf16_bits(p[e]);
    }
  }
}
