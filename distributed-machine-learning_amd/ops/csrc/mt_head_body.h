// Regression head + loss + loss gradient for HL_ROWS batch rows of one slot (gfx950): the body of the head/loss launch
// (head_loss.hip), also called by the resident epoch kernel (resident.hip).
#pragma once
#include "mt_bodies.h"

#define HL_THREADS 256
#define HL_WAVES 4

// Eight consecutive activations of one row as the lane holds them: packed bf16 (one 16-byte load) or f32 (two).
template <bool F32> struct HRow8;
template <> struct HRow8<false> {
  u32x4 v;
  __device__ __forceinline__ void load(const void* H, size_t idx, bool ok) {
    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(H) + (ok ? idx : (size_t)0));
    if (!ok) v = u32x4{0u, 0u, 0u, 0u};
  }
  __device__ __forceinline__ float get(int e) const {
    return (e & 1) ? bf16_bits_to_f32(v[e >> 1] >> 16) : bf16_bits_to_f32(v[e >> 1] & 0xFFFFu);
  }
  static __device__ __forceinline__ void store(void* D, size_t idx, const float (&o)[8]) {
    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(D) + idx) =
        u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
  }
};
template <> struct HRow8<true> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const void* H, size_t idx, bool ok) {
    const float* p = reinterpret_cast<const float*>(H) + (ok ? idx : (size_t)0);
    a = *reinterpret_cast<const f32x4*>(p);
    b = *reinterpret_cast<const f32x4*>(p + 4);
    if (!ok) { a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a; }
  }
  __device__ __forceinline__ float get(int e) const { return e < 4 ? a[e] : b[e - 4]; }
  static __device__ __forceinline__ void store(void* D, size_t idx, const float (&o)[8]) {
    float* p = reinterpret_cast<float*>(D) + idx;
    *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{o[4], o[5], o[6], o[7]};
  }
};

// Training: grid (row_chunks, active slots).  Workgroup (c, a) handles rows [c*HL_ROWS, (c+1)*HL_ROWS) of
// slot ids[a] and writes one partial record {gw[0..W), gb, loss} to scratch; mt_head_reduce_kernel then adds
// the records of a slot in chunk order (fixed order => bitwise reproducible, no float atomics).
#define HL_RB 4          // rows a wave keeps in flight at once

// red: HL_WAVES x (NCHUNK * 512) floats of LDS, red_s: HL_WAVES x 2 floats of LDS (the caller owns them and makes sure no
// wave still uses them when the body starts).
template <int NCHUNK, bool F32>
__device__ __forceinline__ void mt_head_loss_body(const MtHeadSlot* __restrict__ slots, const int32_t* __restrict__ ids,
                                                  int step, float* __restrict__ scratch, int max_chunks, int rec, int bx,
                                                  int by, float (*red)[NCHUNK * 512], float (*red_s)[2]) {
  const int slot = ids[by];
  const MtHeadSlot s = slots[slot];
  if (step >= s.active_steps) return;
  const int row_lo = bx * HL_ROWS;
  if (row_lo >= s.B) return;
  const int row_hi = (row_lo + HL_ROWS < s.B) ? row_lo + HL_ROWS : s.B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const void* H = s.H;
  void* dH = s.dH;

  float w[NCHUNK][8], gw[NCHUNK][8];
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = c * 512 + lane * 8 + e;
      w[c][e] = (k < s.W) ? s.w[k] : 0.f;
      gw[c][e] = 0.f;
    }
  const float bias = s.b[0];
  const float inv_b = 1.f / (float)s.B;
  float loss_sum = 0.f, gb = 0.f;

  // rows wave, wave+4, ... of the chunk; HL_RB of them are loaded before the first is reduced, so the
  // row loads overlap instead of forming a chain of dependent memory round trips
  for (int i0 = row_lo + wave; i0 < row_hi; i0 += HL_WAVES * HL_RB) {
    HRow8<F32> hv[HL_RB][NCHUNK];
#pragma unroll
    for (int r = 0; r < HL_RB; ++r) {
      const int i = i0 + r * HL_WAVES;
#pragma unroll
      for (int c = 0; c < NCHUNK; ++c) {
        const int k0 = c * 512 + lane * 8;
        hv[r][c].load(H, (size_t)i * s.ldh + k0, (i < row_hi) && (k0 < s.W));
      }
    }
#pragma unroll
    for (int r = 0; r < HL_RB; ++r) {
      const int i = i0 + r * HL_WAVES;
      if (i >= row_hi) break;
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) dot = __builtin_fmaf(hv[r][c].get(e), w[c][e], dot);     // explicit: rowblock.hip repeats this chain
      const float out = wave_sum(dot) + bias;
      float l, g;
      loss_and_grad(s.loss_type, out, s.y[i], l, g);
      const float dout = g * inv_b;
      loss_sum += l;
      gb += dout;
#pragma unroll
      for (int c = 0; c < NCHUNK; ++c) {
        const int k0 = c * 512 + lane * 8;
        if (k0 < s.W) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float h = hv[r][c].get(e);
            gw[c][e] = __builtin_fmaf(dout, h, gw[c][e]);
            o[e] = h > 0.f ? dout * w[c][e] : 0.f;
          }
          HRow8<F32>::store(dH, (size_t)i * s.ldh + k0, o);
        }
      }
    }
  }

#pragma unroll
  for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][c * 512 + lane * 8 + e] = gw[c][e];
  if (lane == 0) { red_s[wave][0] = loss_sum; red_s[wave][1] = gb; }
  __syncthreads();
  float* out = scratch + ((size_t)slot * max_chunks + bx) * rec;
  for (int k = tid; k < s.W; k += HL_THREADS)
    out[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  if (tid == 0) {
    out[rec - 2] = (red_s[0][1] + red_s[1][1]) + (red_s[2][1] + red_s[3][1]);
    out[rec - 1] = ((red_s[0][0] + red_s[1][0]) + (red_s[2][0] + red_s[3][0])) * inv_b;
  }
}
