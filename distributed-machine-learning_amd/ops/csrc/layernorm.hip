// Fused dropout + residual add + LayerNorm, forward and backward (gfx950, f32 rows, one wave per row).
//
//   forward : z = residual + x * keep * keep_scale;  y = (z - mean(z)) * rstd(z) * w + b
//             writes y, z and the per-row {mean, rstd}
//   backward: dz = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w,  xhat = (z - mean) * rstd
//             writes dres = dz, dx = dz * keep * keep_scale and per-block partial sums of
//             dw = sum_rows dy * xhat, db = sum_rows dy (summed over blocks by the caller, fixed order).
// A row (d <= 2048, d % 4 == 0) is held in registers as up to 8 float4 per lane: one global read of
// each input, one write of each output, wave-shuffle reductions, no LDS in forward.
#include "mt_common.h"

#define LN_THREADS 256
#define LN_MAXV 8           // float4 per lane: 64 lanes * 4 * 8 = 2048 columns
#define LN_ROWS_PER_WAVE 8  // backward: rows per wave, so a block covers 32 rows

__global__ __launch_bounds__(LN_THREADS) void mt_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                               const uint8_t* __restrict__ keep,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               float* __restrict__ y, float* __restrict__ z,
                                                               float* __restrict__ stats, int M, int d, float eps,
                                                               float keep_scale, const int64_t* __restrict__ seed,
                                                               uint32_t thresh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  const int nv = d >> 2;
  const size_t base = (size_t)row * d;
  f32x4 v[LN_MAXV];
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < LN_MAXV; ++t) {
    const int c = lane + 64 * t;
    v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nv) {
      f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + 4 * c);
      const f32x4 rv = *reinterpret_cast<const f32x4*>(res + base + 4 * c);
      if (keep != nullptr || seed != nullptr) {
        const uint32_t kw = mt_keep_word4(keep, seed, thresh, base + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] *= ((kw >> (8 * e)) & 0xFFu) ? keep_scale : 0.f;
      }
      v[t] = xv + rv;
      *reinterpret_cast<f32x4*>(z + base + 4 * c) = v[t];
      sum += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
    }
  }
  const float mean = wave_sum(sum) / (float)d;
  float var = 0.f;
#pragma unroll
  for (int t = 0; t < LN_MAXV; ++t)
    if (lane + 64 * t < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float c0 = v[t][e] - mean; var += c0 * c0; }
    }
  const float rstd = rsqrtf(wave_sum(var) / (float)d + eps);
#pragma unroll
  for (int t = 0; t < LN_MAXV; ++t) {
    const int c = lane + 64 * t;
    if (c < nv) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + 4 * c), bv = *reinterpret_cast<const f32x4*>(b + 4 * c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[t][e] - mean) * rstd * wv[e] + bv[e];
      *reinterpret_cast<f32x4*>(y + base + 4 * c) = o;
    }
  }
  if (lane == 0) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
}

// Narrow rows (d <= 256: one float4 per lane).  One row per wave leaves a wave with two 16-byte loads in flight and
// two serial reductions; here a wave takes LN_SMALL_ROWS rows at once -- every load is issued before the first
// reduction, out-of-range lanes read a clamped address and select zero (no branch around a load), and the
// reductions of the rows interleave.
#define LN_SMALL_ROWS 4
__global__ __launch_bounds__(LN_THREADS) void mt_ln_fwd_small_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                     const uint8_t* __restrict__ keep,
                                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                                     float* __restrict__ y, float* __restrict__ z,
                                                                     float* __restrict__ stats, int M, int d, float eps,
                                                                     float keep_scale, const int64_t* __restrict__ seed,
                                                                     uint32_t thresh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * LN_SMALL_ROWS;
  if (row0 >= M) return;
  const int nv = d >> 2;
  const bool col_ok = lane < nv;
  const int c4 = col_ok ? 4 * lane : 0;
  f32x4 v[LN_SMALL_ROWS];
  float sum[LN_SMALL_ROWS];
#pragma unroll
  for (int i = 0; i < LN_SMALL_ROWS; ++i) {
    const bool ok = col_ok && (row0 + i < M);
    const size_t off = (size_t)(row0 + i < M ? row0 + i : row0) * d + c4;
    f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(res + off);
    if (keep != nullptr || seed != nullptr) {        // uniform
      const uint32_t kw = mt_keep_word4(keep, seed, thresh, off);
#pragma unroll
      for (int e = 0; e < 4; ++e) xv[e] *= ((kw >> (8 * e)) & 0xFFu) ? keep_scale : 0.f;
    }
    const f32x4 zv = xv + rv;
    v[i] = ok ? zv : f32x4{0.f, 0.f, 0.f, 0.f};
    sum[i] = (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  float mean[LN_SMALL_ROWS], rstd[LN_SMALL_ROWS];
#pragma unroll
  for (int i = 0; i < LN_SMALL_ROWS; ++i) mean[i] = wave_sum(sum[i]) / (float)d;
#pragma unroll
  for (int i = 0; i < LN_SMALL_ROWS; ++i) {
    float var = 0.f;
    if (col_ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float c0 = v[i][e] - mean[i]; var += c0 * c0; }
    }
    sum[i] = var;
  }
#pragma unroll
  for (int i = 0; i < LN_SMALL_ROWS; ++i) rstd[i] = rsqrtf(wave_sum(sum[i]) / (float)d + eps);
  const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c4), bv = *reinterpret_cast<const f32x4*>(b + c4);
#pragma unroll
  for (int i = 0; i < LN_SMALL_ROWS; ++i) {
    if (col_ok && row0 + i < M) {
      const size_t off = (size_t)(row0 + i) * d + c4;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean[i]) * rstd[i] * wv[e] + bv[e];
      *reinterpret_cast<f32x4*>(z + off) = v[i];
      *reinterpret_cast<f32x4*>(y + off) = o;
    }
    if (lane == 0 && row0 + i < M) { stats[2 * (size_t)(row0 + i)] = mean[i]; stats[2 * (size_t)(row0 + i) + 1] = rstd[i]; }
  }
}

__global__ __launch_bounds__(LN_THREADS) void mt_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                               const float* __restrict__ stats,
                                                               const uint8_t* __restrict__ keep,
                                                               const float* __restrict__ w, float* __restrict__ dx,
                                                               float* __restrict__ dres, float* __restrict__ dwb_part,
                                                               int M, int d, float keep_scale,
                                                               const int64_t* __restrict__ seed, uint32_t thresh, int reps) {
  __shared__ float red[4][2 * 2048 / 4];     // per wave: dw then db for one float4 column group at a time
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = d >> 2;
  f32x4 dw[LN_MAXV], db[LN_MAXV];
#pragma unroll
  for (int t = 0; t < LN_MAXV; ++t) { dw[t] = f32x4{0.f, 0.f, 0.f, 0.f}; db[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int r = 0; r < reps * LN_ROWS_PER_WAVE; ++r) {
    const int row = ((blockIdx.x * reps + r / LN_ROWS_PER_WAVE) * 4 + wave) * LN_ROWS_PER_WAVE + r % LN_ROWS_PER_WAVE;
    if (row >= M) break;
    const size_t base = (size_t)row * d;
    const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
    f32x4 g[LN_MAXV], xh[LN_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
      const int c = lane + 64 * t;
      g[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      xh[t] = g[t];
      if (c < nv) {
        const f32x4 dyv = *reinterpret_cast<const f32x4*>(dy + base + 4 * c);
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + base + 4 * c);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[t][e] = (zv[e] - mean) * rstd;
          g[t][e] = dyv[e] * wv[e];
          s1 += g[t][e];
          s2 += g[t][e] * xh[t][e];
          dw[t][e] += dyv[e] * xh[t][e];
          db[t][e] += dyv[e];
        }
      }
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
      const int c = lane + 64 * t;
      if (c < nv) {
        f32x4 dz, dxv;
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[e] = rstd * (g[t][e] - s1 - xh[t][e] * s2);
        dxv = dz;
        if (keep != nullptr || seed != nullptr) {
          const uint32_t kw = mt_keep_word4(keep, seed, thresh, base + 4 * c);
#pragma unroll
          for (int e = 0; e < 4; ++e) dxv[e] *= ((kw >> (8 * e)) & 0xFFu) ? keep_scale : 0.f;
        }
        *reinterpret_cast<f32x4*>(dres + base + 4 * c) = dz;
        *reinterpret_cast<f32x4*>(dx + base + 4 * c) = dxv;
      }
    }
  }
  // block-level partial of dw / db: waves take turns through LDS, one float4 column group at a time
  float* out = dwb_part + (size_t)blockIdx.x * 2 * d;
#pragma unroll
  for (int t = 0; t < LN_MAXV; ++t) {
    const int c = lane + 64 * t;
    __syncthreads();
    if (c < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[wave][(lane * 4 + e)] = dw[t][e];
        red[wave][256 + (lane * 4 + e)] = db[t][e];
      }
    }
    __syncthreads();
    if (wave == 0 && c < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = lane * 4 + e;
        out[4 * c + e] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        out[d + 4 * c + e] = (red[0][256 + k] + red[1][256 + k]) + (red[2][256 + k] + red[3][256 + k]);
      }
    }
  }
}

// Narrow rows (d <= 256): the same regrouping as the narrow forward.  A wave takes LN_SMALL_ROWS rows per round with every
// load issued before the first reduction and the two reductions of the rows interleaved, instead of eight rows one
// after the other with two dependent reductions each; the weight/bias partials need one LDS exchange, not eight.
// Rows per block and the layout of the per-block partials are those of mt_ln_bwd_kernel.
__global__ __launch_bounds__(LN_THREADS) void mt_ln_bwd_small_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                                     const float* __restrict__ stats,
                                                                     const uint8_t* __restrict__ keep,
                                                                     const float* __restrict__ w, float* __restrict__ dx,
                                                                     float* __restrict__ dres, float* __restrict__ dwb_part,
                                                                     int M, int d, float keep_scale,
                                                                     const int64_t* __restrict__ seed, uint32_t thresh,
                                                                     int reps) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = d >> 2;
  const bool col_ok = lane < nv;
  const int c4 = col_ok ? 4 * lane : 0;
  const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c4);
  f32x4 dw = f32x4{0.f, 0.f, 0.f, 0.f}, db = dw;
  const float inv_d = 1.f / (float)d;
  for (int rep = 0; rep < reps; ++rep) {
  const int row0 = ((blockIdx.x * reps + rep) * 4 + wave) * LN_ROWS_PER_WAVE;
#pragma unroll
  for (int it = 0; it < LN_ROWS_PER_WAVE; it += LN_SMALL_ROWS) {
    if (row0 + it >= M) break;                                   // wave-uniform
    f32x4 g[LN_SMALL_ROWS], xh[LN_SMALL_ROWS];
    float s1[LN_SMALL_ROWS], s2[LN_SMALL_ROWS], rstd[LN_SMALL_ROWS];
#pragma unroll
    for (int i = 0; i < LN_SMALL_ROWS; ++i) {
      const int row = row0 + it + i;
      const bool ok = col_ok && row < M;
      const int rc = row < M ? row : row0 + it;                  // clamped: loads never branch
      const size_t off = (size_t)rc * d + c4;
      const f32x4 dyv = *reinterpret_cast<const f32x4*>(dy + off);
      const f32x4 zv = *reinterpret_cast<const f32x4*>(z + off);
      const float mean = stats[2 * (size_t)rc];
      rstd[i] = stats[2 * (size_t)rc + 1];
      s1[i] = 0.f; s2[i] = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dye = ok ? dyv[e] : 0.f;
        xh[i][e] = ok ? (zv[e] - mean) * rstd[i] : 0.f;
        g[i][e] = dye * wv[e];
        s1[i] += g[i][e];
        s2[i] += g[i][e] * xh[i][e];
        dw[e] += dye * xh[i][e];
        db[e] += dye;
      }
    }
#pragma unroll
    for (int i = 0; i < LN_SMALL_ROWS; ++i) s1[i] = wave_sum(s1[i]) * inv_d;
#pragma unroll
    for (int i = 0; i < LN_SMALL_ROWS; ++i) s2[i] = wave_sum(s2[i]) * inv_d;
#pragma unroll
    for (int i = 0; i < LN_SMALL_ROWS; ++i) {
      const int row = row0 + it + i;
      if (col_ok && row < M) {
        const size_t off = (size_t)row * d + c4;
        f32x4 dz, dxv;
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[e] = rstd[i] * (g[i][e] - s1[i] - xh[i][e] * s2[i]);
        dxv = dz;
        if (keep != nullptr || seed != nullptr) {
          const uint32_t kw = mt_keep_word4(keep, seed, thresh, off);
#pragma unroll
          for (int e = 0; e < 4; ++e) dxv[e] *= ((kw >> (8 * e)) & 0xFFu) ? keep_scale : 0.f;
        }
        *reinterpret_cast<f32x4*>(dres + off) = dz;
        *reinterpret_cast<f32x4*>(dx + off) = dxv;
      }
    }
  }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[wave][lane * 4 + e] = dw[e];
    red[wave][256 + lane * 4 + e] = db[e];
  }
  __syncthreads();
  float* out = dwb_part + (size_t)blockIdx.x * 2 * d;
  for (int k = threadIdx.x; k < d; k += LN_THREADS) {            // d <= 256: one pass
    out[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    out[d + k] = (red[0][256 + k] + red[1][256 + k]) + (red[2][256 + k] + red[3][256 + k]);
  }
}

// A block takes `reps` consecutive groups of 32 rows and writes one partial of the weight / bias gradients: large row
// counts are capped at about LN_BWD_MAX_BLOCKS partials, so the column sum over them that follows stays a small kernel
// (66 560 rows x 192 columns: 694 partials instead of 2 080).  Up to that many groups nothing changes.
#define LN_BWD_MAX_BLOCKS 1024
static int ln_bwd_groups(int M) { return (M + 4 * LN_ROWS_PER_WAVE - 1) / (4 * LN_ROWS_PER_WAVE); }
static int ln_bwd_reps(int M) { return (ln_bwd_groups(M) + LN_BWD_MAX_BLOCKS - 1) / LN_BWD_MAX_BLOCKS > 1
                                           ? (ln_bwd_groups(M) + LN_BWD_MAX_BLOCKS - 1) / LN_BWD_MAX_BLOCKS : 1; }
extern "C" int mt_ln_bwd_blocks(int M) { const int r = ln_bwd_reps(M); return (ln_bwd_groups(M) + r - 1) / r; }

extern "C" int mt_ln_fwd_launch(const float* x, const float* res, const void* keep, const float* w, const float* b,
                                float* y, float* z, float* stats, int M, int d, float eps, float keep_scale,
                                const int64_t* seed, float p, hipStream_t stream) {
  // dropout mask: `keep` (uint8 per element) if given, else drawn in the kernel from `seed` with drop probability p
  const uint32_t thresh = (seed != nullptr && p > 0.f) ? (uint32_t)((double)p * 4294967296.0) : 0u;
  if (thresh == 0u) seed = nullptr;
  if (M <= 0) return 0;
  if (d > 2048 || (d & 3)) return -2;
  if (d <= 256)
    hipLaunchKernelGGL(mt_ln_fwd_small_kernel, dim3((M + 4 * LN_SMALL_ROWS - 1) / (4 * LN_SMALL_ROWS)), dim3(LN_THREADS), 0,
                       stream, x, res, (const uint8_t*)keep, w, b, y, z, stats, M, d, eps, keep_scale, seed, thresh);
  else
    hipLaunchKernelGGL(mt_ln_fwd_kernel, dim3((M + 3) / 4), dim3(LN_THREADS), 0, stream, x, res, (const uint8_t*)keep, w,
                       b, y, z, stats, M, d, eps, keep_scale, seed, thresh);
  return (int)hipGetLastError();
}

extern "C" int mt_ln_bwd_launch(const float* dy, const float* z, const float* stats, const void* keep, const float* w,
                                float* dx, float* dres, float* dwb_part, int M, int d, float keep_scale,
                                const int64_t* seed, float p, hipStream_t stream) {
  const uint32_t thresh = (seed != nullptr && p > 0.f) ? (uint32_t)((double)p * 4294967296.0) : 0u;
  if (thresh == 0u) seed = nullptr;
  if (M <= 0) return 0;
  if (d > 2048 || (d & 3)) return -2;
  if (d <= 256)
    hipLaunchKernelGGL(mt_ln_bwd_small_kernel, dim3(mt_ln_bwd_blocks(M)), dim3(LN_THREADS), 0, stream, dy, z, stats,
                       (const uint8_t*)keep, w, dx, dres, dwb_part, M, d, keep_scale, seed, thresh, ln_bwd_reps(M));
  else
    hipLaunchKernelGGL(mt_ln_bwd_kernel, dim3(mt_ln_bwd_blocks(M)), dim3(LN_THREADS), 0, stream, dy, z, stats,
                       (const uint8_t*)keep, w, dx, dres, dwb_part, M, d, keep_scale, seed, thresh, ln_bwd_reps(M));
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Mixed-precision forms for bf16-autocast models (rows of d <= 256): the branch output x arrives in bf16 straight from
// the GEMM that produced it, the residual stream stays f32, and the normalised output can be written twice -- f32 for
// the residual stream, bf16 for the next GEMM's operand -- so no cast kernel runs on either side.  Backward takes the
// two gradients of those two outputs (either may be absent), adds them while loading, and writes the gradient of x in
// x's own precision.  Dropout masks come from the device seed only.
__device__ __forceinline__ f32x4 ln_ld4_bf16(const uint16_t* p) {
  const u32x2 v = *reinterpret_cast<const u32x2*>(p);
  return f32x4{bf16_bits_to_f32(v[0] & 0xFFFFu), bf16_bits_to_f32(v[0] >> 16), bf16_bits_to_f32(v[1] & 0xFFFFu),
               bf16_bits_to_f32(v[1] >> 16)};
}
__device__ __forceinline__ void ln_st4_bf16(uint16_t* p, const f32x4 v) {
  *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
}

template <bool XB>
__global__ __launch_bounds__(LN_THREADS) void mt_ln_fwd_mixed_kernel(const void* __restrict__ x_, const float* __restrict__ res,
                                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                                     float* __restrict__ y, uint16_t* __restrict__ y16,
                                                                     float* __restrict__ z, float* __restrict__ stats,
                                                                     int M, int d, float eps, float keep_scale,
                                                                     const int64_t* __restrict__ seed, uint32_t thresh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * LN_SMALL_ROWS;
  if (row0 >= M) return;
  const int nv = d >> 2;
  const bool col_ok = lane < nv;
  const int c4 = col_ok ? 4 * lane : 0;
  f32x4 v[LN_SMALL_ROWS];
  float sum[LN_SMALL_ROWS];
#pragma unroll
  for (int i = 0; i < LN_SMALL_ROWS; ++i) {
    const bool ok = col_ok && (row0 + i < M);
    const size_t off = (size_t)(row0 + i < M ? row0 + i : row0) * d + c4;
    f32x4 xv;
    if constexpr (XB) xv = ln_ld4_bf16(reinterpret_cast<const uint16_t*>(x_) + off);
    else xv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x_) + off);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(res + off);
    if (seed != nullptr) {
      const uint32_t kw = mt_keep_word4(nullptr, seed, thresh, off);
#pragma unroll
      for (int e = 0; e < 4; ++e) xv[e] *= ((kw >> (8 * e)) & 0xFFu) ? keep_scale : 0.f;
    }
    const f32x4 zv = xv + rv;
    v[i] = ok ? zv : f32x4{0.f, 0.f, 0.f, 0.f};
    sum[i] = (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  float mean[LN_SMALL_ROWS], rstd[LN_SMALL_ROWS];
#pragma unroll
  for (int i = 0; i < LN_SMALL_ROWS; ++i) mean[i] = wave_sum(sum[i]) / (float)d;
#pragma unroll
  for (int i = 0; i < LN_SMALL_ROWS; ++i) {
    float var = 0.f;
    if (col_ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float c0 = v[i][e] - mean[i]; var += c0 * c0; }
    }
    sum[i] = var;
  }
#pragma unroll
  for (int i = 0; i < LN_SMALL_ROWS; ++i) rstd[i] = rsqrtf(wave_sum(sum[i]) / (float)d + eps);
  const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c4), bv = *reinterpret_cast<const f32x4*>(b + c4);
#pragma unroll
  for (int i = 0; i < LN_SMALL_ROWS; ++i) {
    if (col_ok && row0 + i < M) {
      const size_t off = (size_t)(row0 + i) * d + c4;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean[i]) * rstd[i] * wv[e] + bv[e];
      *reinterpret_cast<f32x4*>(z + off) = v[i];
      *reinterpret_cast<f32x4*>(y + off) = o;
      if (y16 != nullptr) ln_st4_bf16(y16 + off, o);
    }
    if (lane == 0 && row0 + i < M) { stats[2 * (size_t)(row0 + i)] = mean[i]; stats[2 * (size_t)(row0 + i) + 1] = rstd[i]; }
  }
}

template <bool XB>
__global__ __launch_bounds__(LN_THREADS) void mt_ln_bwd_mixed_kernel(const float* __restrict__ dy, const uint16_t* __restrict__ dy16,
                                                                     const float* __restrict__ z, const float* __restrict__ stats,
                                                                     const float* __restrict__ w, void* __restrict__ dx_,
                                                                     float* __restrict__ dres, float* __restrict__ dwb_part,
                                                                     int M, int d, float keep_scale,
                                                                     const int64_t* __restrict__ seed, uint32_t thresh,
                                                                     int reps) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = d >> 2;
  const bool col_ok = lane < nv;
  const int c4 = col_ok ? 4 * lane : 0;
  const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c4);
  f32x4 dw = f32x4{0.f, 0.f, 0.f, 0.f}, db = dw;
  const float inv_d = 1.f / (float)d;
  for (int rep = 0; rep < reps; ++rep) {
  const int row0 = ((blockIdx.x * reps + rep) * 4 + wave) * LN_ROWS_PER_WAVE;
#pragma unroll
  for (int it = 0; it < LN_ROWS_PER_WAVE; it += LN_SMALL_ROWS) {
    if (row0 + it >= M) break;                                   // wave-uniform
    f32x4 g[LN_SMALL_ROWS], xh[LN_SMALL_ROWS];
    float s1[LN_SMALL_ROWS], s2[LN_SMALL_ROWS], rstd[LN_SMALL_ROWS];
#pragma unroll
    for (int i = 0; i < LN_SMALL_ROWS; ++i) {
      const int row = row0 + it + i;
      const bool ok = col_ok && row < M;
      const int rc = row < M ? row : row0 + it;
      const size_t off = (size_t)rc * d + c4;
      f32x4 dyv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (dy != nullptr) dyv = *reinterpret_cast<const f32x4*>(dy + off);
      if (dy16 != nullptr) dyv += ln_ld4_bf16(dy16 + off);
      const f32x4 zv = *reinterpret_cast<const f32x4*>(z + off);
      const float mean = stats[2 * (size_t)rc];
      rstd[i] = stats[2 * (size_t)rc + 1];
      s1[i] = 0.f; s2[i] = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dye = ok ? dyv[e] : 0.f;
        xh[i][e] = ok ? (zv[e] - mean) * rstd[i] : 0.f;
        g[i][e] = dye * wv[e];
        s1[i] += g[i][e];
        s2[i] += g[i][e] * xh[i][e];
        dw[e] += dye * xh[i][e];
        db[e] += dye;
      }
    }
#pragma unroll
    for (int i = 0; i < LN_SMALL_ROWS; ++i) s1[i] = wave_sum(s1[i]) * inv_d;
#pragma unroll
    for (int i = 0; i < LN_SMALL_ROWS; ++i) s2[i] = wave_sum(s2[i]) * inv_d;
#pragma unroll
    for (int i = 0; i < LN_SMALL_ROWS; ++i) {
      const int row = row0 + it + i;
      if (col_ok && row < M) {
        const size_t off = (size_t)row * d + c4;
        f32x4 dz, dxv;
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[e] = rstd[i] * (g[i][e] - s1[i] - xh[i][e] * s2[i]);
        dxv = dz;
        if (seed != nullptr) {
          const uint32_t kw = mt_keep_word4(nullptr, seed, thresh, off);
#pragma unroll
          for (int e = 0; e < 4; ++e) dxv[e] *= ((kw >> (8 * e)) & 0xFFu) ? keep_scale : 0.f;
        }
        *reinterpret_cast<f32x4*>(dres + off) = dz;
        if constexpr (XB) ln_st4_bf16(reinterpret_cast<uint16_t*>(dx_) + off, dxv);
        else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dx_) + off) = dxv;
      }
    }
  }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[wave][lane * 4 + e] = dw[e];
    red[wave][256 + lane * 4 + e] = db[e];
  }
  __syncthreads();
  float* out = dwb_part + (size_t)blockIdx.x * 2 * d;
  for (int k = threadIdx.x; k < d; k += LN_THREADS) {
    out[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    out[d + k] = (red[0][256 + k] + red[1][256 + k]) + (red[2][256 + k] + red[3][256 + k]);
  }
}

// x_bf16 != 0: x is bf16; y16 (may be null): second, bf16 copy of the output.
extern "C" int mt_ln_fwd_mixed_launch(const void* x, int x_bf16, const float* res, const float* w, const float* b, float* y,
                                      void* y16, float* z, float* stats, int M, int d, float eps, float keep_scale,
                                      const int64_t* seed, float p, hipStream_t stream) {
  const uint32_t thresh = (seed != nullptr && p > 0.f) ? (uint32_t)((double)p * 4294967296.0) : 0u;
  if (thresh == 0u) seed = nullptr;
  if (M <= 0) return 0;
  if (d > 256 || (d & 3) || !x || !res || !w || !b || !y || !z || !stats) return -2;
  const dim3 grid((M + 4 * LN_SMALL_ROWS - 1) / (4 * LN_SMALL_ROWS)), block(LN_THREADS);
  if (x_bf16)
    hipLaunchKernelGGL(mt_ln_fwd_mixed_kernel<true>, grid, block, 0, stream, x, res, w, b, y, (uint16_t*)y16, z, stats, M, d,
                       eps, keep_scale, seed, thresh);
  else
    hipLaunchKernelGGL(mt_ln_fwd_mixed_kernel<false>, grid, block, 0, stream, x, res, w, b, y, (uint16_t*)y16, z, stats, M, d,
                       eps, keep_scale, seed, thresh);
  return (int)hipGetLastError();
}

// dy (f32) and dy16 (bf16): gradients of the two outputs, at least one given; dx is written in bf16 when dx_bf16 != 0.
extern "C" int mt_ln_bwd_mixed_launch(const float* dy, const void* dy16, const float* z, const float* stats, const float* w,
                                      void* dx, int dx_bf16, float* dres, float* dwb_part, int M, int d, float keep_scale,
                                      const int64_t* seed, float p, hipStream_t stream) {
  const uint32_t thresh = (seed != nullptr && p > 0.f) ? (uint32_t)((double)p * 4294967296.0) : 0u;
  if (thresh == 0u) seed = nullptr;
  if (M <= 0) return 0;
  if (d > 256 || (d & 3) || (!dy && !dy16) || !z || !stats || !w || !dx || !dres || !dwb_part) return -2;
  const dim3 grid(mt_ln_bwd_blocks(M)), block(LN_THREADS);
  if (dx_bf16)
    hipLaunchKernelGGL(mt_ln_bwd_mixed_kernel<true>, grid, block, 0, stream, dy, (const uint16_t*)dy16, z, stats, w, dx, dres,
                       dwb_part, M, d, keep_scale, seed, thresh, ln_bwd_reps(M));
  else
    hipLaunchKernelGGL(mt_ln_bwd_mixed_kernel<false>, grid, block, 0, stream, dy, (const uint16_t*)dy16, z, stats, w, dx, dres,
                       dwb_part, M, d, keep_scale, seed, thresh, ln_bwd_reps(M));
  return (int)hipGetLastError();
}
