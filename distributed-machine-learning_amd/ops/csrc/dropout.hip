// Counter-based dropout for the Transformer path (gfx950): keep masks and fused dropout without the
// framework's random-number generator.
//
// Element i is kept iff hash(seed, i) >= p * 2^32, with the 64-bit seed read from DEVICE memory at execution
// time.  Nothing about a mask is a host-side value, so a training step captured in a hipGraph draws fresh
// masks on every replay (the caller bumps the device counter inside the captured step), and the backward pass
// recomputes the forward's mask from the same seed snapshot instead of storing it.
//
//   mt_keep_mask_launch     uint8 mask [n] (1 = keep) for kernels that take a mask operand (attention, LayerNorm)
//   mt_dropout_launch       y = keep ? x / (1 - p) : 0 for f32 or bf16 data; the same call with dy gives dx
//   mt_relu_dropout_*       y = keep ? max(x, 0) / (1 - p) : 0 (the feed-forward block's activation + dropout in one
//                           pass) and dx = (y != 0) ? dy / (1 - p) : 0: y itself records both the sign and the mask
#include "mt_common.h"

#define dr_hash mt_rng_hash       // the same hash the fused kernels use for their in-kernel masks

__global__ __launch_bounds__(256) void mt_keep_mask_kernel(uint8_t* __restrict__ out, long n, const int64_t* __restrict__ seed,
                                                           uint32_t thresh) {
  const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const uint64_t s = (uint64_t)seed[0];
  const uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
  uint32_t w = 0u;
#pragma unroll
  for (int e = 0; e < 4; ++e) w |= (dr_hash(s0, s1, (uint32_t)(i4 + e)) >= thresh ? 1u : 0u) << (8 * e);
  if (i4 + 4 <= n) *reinterpret_cast<uint32_t*>(out + i4) = w;
  else for (long e = i4; e < n; ++e) out[e] = (uint8_t)((w >> (8 * (e - i4))) & 0xFFu);
}

template <typename T> struct DrIO;
template <> struct DrIO<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct DrIO<uint16_t> {   // bf16 bits
  static __device__ __forceinline__ float ld(const uint16_t* p) { return bf16_bits_to_f32(*p); }
  static __device__ __forceinline__ void st(uint16_t* p, float v) { *p = (uint16_t)f32_to_bf16_bits(v); }
};

template <typename T>
__global__ __launch_bounds__(256) void mt_dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long n,
                                                         const int64_t* __restrict__ seed, uint32_t thresh, float scale) {
  const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const uint64_t s = (uint64_t)seed[0];
  const uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
  const int cnt = (n - i4) < 4 ? (int)(n - i4) : 4;
  for (int e = 0; e < cnt; ++e) {
    const float v = DrIO<T>::ld(x + i4 + e);
    DrIO<T>::st(y + i4 + e, dr_hash(s0, s1, (uint32_t)(i4 + e)) >= thresh ? v * scale : 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mt_relu_dropout_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n,
                                                                  const int64_t* __restrict__ seed, uint32_t thresh,
                                                                  float scale) {
  const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const uint64_t s = (uint64_t)seed[0];
  const uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
  const int cnt = (n - i4) < 4 ? (int)(n - i4) : 4;
  for (int e = 0; e < cnt; ++e) {
    const float v = fmaxf(DrIO<T>::ld(x + i4 + e), 0.f);
    DrIO<T>::st(y + i4 + e, dr_hash(s0, s1, (uint32_t)(i4 + e)) >= thresh ? v * scale : 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mt_relu_dropout_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                                  T* __restrict__ dx, long n, float scale) {
  const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const int cnt = (n - i4) < 4 ? (int)(n - i4) : 4;
  for (int e = 0; e < cnt; ++e)
    DrIO<T>::st(dx + i4 + e, DrIO<T>::ld(y + i4 + e) != 0.f ? DrIO<T>::ld(dy + i4 + e) * scale : 0.f);
}

static inline uint32_t dr_thresh(float p) { return (uint32_t)((double)p * 4294967296.0); }

extern "C" int mt_keep_mask_launch(void* out, long n, const int64_t* seed, float p, hipStream_t stream) {
  if (n <= 0) return 0;
  if (seed == nullptr || p < 0.f || p >= 1.f) return -2;
  hipLaunchKernelGGL(mt_keep_mask_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, (uint8_t*)out, n, seed,
                     dr_thresh(p));
  return (int)hipGetLastError();
}

// dtype: 0 = f32, 1 = bf16.  x and y may alias.
extern "C" int mt_dropout_launch(const void* x, void* y, long n, const int64_t* seed, float p, int dtype, hipStream_t stream) {
  if (n <= 0) return 0;
  if (seed == nullptr || p < 0.f || p >= 1.f) return -2;
  dim3 grid((unsigned)((n + 1023) / 1024)), block(256);
  const float scale = 1.f / (1.f - p);
  if (dtype == 0)
    hipLaunchKernelGGL(mt_dropout_kernel<float>, grid, block, 0, stream, (const float*)x, (float*)y, n, seed, dr_thresh(p), scale);
  else if (dtype == 1)
    hipLaunchKernelGGL(mt_dropout_kernel<uint16_t>, grid, block, 0, stream, (const uint16_t*)x, (uint16_t*)y, n, seed,
                       dr_thresh(p), scale);
  else return -2;
  return (int)hipGetLastError();
}

extern "C" int mt_relu_dropout_fwd_launch(const void* x, void* y, long n, const int64_t* seed, float p, int dtype,
                                          hipStream_t stream) {
  if (n <= 0) return 0;
  if (seed == nullptr || p < 0.f || p >= 1.f) return -2;
  dim3 grid((unsigned)((n + 1023) / 1024)), block(256);
  const float scale = 1.f / (1.f - p);
  if (dtype == 0)
    hipLaunchKernelGGL(mt_relu_dropout_fwd_kernel<float>, grid, block, 0, stream, (const float*)x, (float*)y, n, seed,
                       dr_thresh(p), scale);
  else if (dtype == 1)
    hipLaunchKernelGGL(mt_relu_dropout_fwd_kernel<uint16_t>, grid, block, 0, stream, (const uint16_t*)x, (uint16_t*)y, n, seed,
                       dr_thresh(p), scale);
  else return -2;
  return (int)hipGetLastError();
}

extern "C" int mt_relu_dropout_bwd_launch(const void* dy, const void* y, void* dx, long n, float p, int dtype,
                                          hipStream_t stream) {
  if (n <= 0) return 0;
  if (p < 0.f || p >= 1.f) return -2;
  dim3 grid((unsigned)((n + 1023) / 1024)), block(256);
  const float scale = 1.f / (1.f - p);
  if (dtype == 0)
    hipLaunchKernelGGL(mt_relu_dropout_bwd_kernel<float>, grid, block, 0, stream, (const float*)dy, (const float*)y, (float*)dx,
                       n, scale);
  else if (dtype == 1)
    hipLaunchKernelGGL(mt_relu_dropout_bwd_kernel<uint16_t>, grid, block, 0, stream, (const uint16_t*)dy, (const uint16_t*)y,
                       (uint16_t*)dx, n, scale);
  else return -2;
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// ReGLU with dropout for bf16 feed-forward blocks: u = [a | g] (row of 2h), out = a * max(g, 0) * keep / (1 - p).
// One pass per direction instead of chunk + relu + multiply + dropout (forward) and five element-wise launches plus
// a concatenation (backward); the backward writes da and dg straight into the two halves of du.  Eight elements per
// thread (16-byte loads); h % 8 == 0.  seed == nullptr or p == 0: no dropout.
__device__ __forceinline__ void rg_unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = bf16_bits_to_f32(v[k] & 0xFFFFu); f[2 * k + 1] = bf16_bits_to_f32(v[k] >> 16); }
}
__device__ __forceinline__ u32x4 rg_pack8(const float (&f)[8]) {
  return u32x4{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
}

__global__ __launch_bounds__(256) void mt_reglu_fwd_kernel(const uint16_t* __restrict__ u, uint16_t* __restrict__ out,
                                                           long rows, int h, const int64_t* __restrict__ seed,
                                                           uint32_t thresh, float scale) {
  const int hc = h >> 3;                                     // 8-element chunks per output row
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= rows * hc) return;
  const long r = q / hc;
  const int c = (int)(q - r * hc) * 8;
  float a[8], g[8], o[8];
  rg_unpack8(*reinterpret_cast<const u32x4*>(u + r * 2 * h + c), a);
  rg_unpack8(*reinterpret_cast<const u32x4*>(u + r * 2 * h + h + c), g);
  uint32_t s0 = 0u, s1 = 0u;
  const bool drop = seed != nullptr && thresh != 0u;
  if (drop) { const uint64_t s = (uint64_t)seed[0]; s0 = (uint32_t)s; s1 = (uint32_t)(s >> 32); }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float keep = (!drop || dr_hash(s0, s1, (uint32_t)(r * h + c + e)) >= thresh) ? scale : 0.f;
    o[e] = a[e] * fmaxf(g[e], 0.f) * keep;
  }
  *reinterpret_cast<u32x4*>(out + r * h + c) = rg_pack8(o);
}

__global__ __launch_bounds__(256) void mt_reglu_bwd_kernel(const uint16_t* __restrict__ u, const uint16_t* __restrict__ dout,
                                                           uint16_t* __restrict__ du, long rows, int h,
                                                           const int64_t* __restrict__ seed, uint32_t thresh, float scale) {
  const int hc = h >> 3;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= rows * hc) return;
  const long r = q / hc;
  const int c = (int)(q - r * hc) * 8;
  float a[8], g[8], d[8], da[8], dg[8];
  rg_unpack8(*reinterpret_cast<const u32x4*>(u + r * 2 * h + c), a);
  rg_unpack8(*reinterpret_cast<const u32x4*>(u + r * 2 * h + h + c), g);
  rg_unpack8(*reinterpret_cast<const u32x4*>(dout + r * h + c), d);
  uint32_t s0 = 0u, s1 = 0u;
  const bool drop = seed != nullptr && thresh != 0u;
  if (drop) { const uint64_t s = (uint64_t)seed[0]; s0 = (uint32_t)s; s1 = (uint32_t)(s >> 32); }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float keep = (!drop || dr_hash(s0, s1, (uint32_t)(r * h + c + e)) >= thresh) ? scale : 0.f;
    const float dk = d[e] * keep;
    da[e] = dk * fmaxf(g[e], 0.f);
    dg[e] = g[e] > 0.f ? dk * a[e] : 0.f;
  }
  *reinterpret_cast<u32x4*>(du + r * 2 * h + c) = rg_pack8(da);
  *reinterpret_cast<u32x4*>(du + r * 2 * h + h + c) = rg_pack8(dg);
}

static int rg_check(const void* a, const void* b, long rows, int h, float p) {
  if (!a || !b || h <= 0 || (h & 7) || p < 0.f || p >= 1.f || rows * (long)h >= (1L << 32)) return -2;
  return 0;
}

extern "C" int mt_reglu_fwd_launch(const void* u, void* out, long rows, int h, const int64_t* seed, float p,
                                   hipStream_t stream) {
  if (rows <= 0) return 0;
  if (rg_check(u, out, rows, h, p)) return -2;
  const long n = rows * (h >> 3);
  hipLaunchKernelGGL(mt_reglu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint16_t*)u,
                     (uint16_t*)out, rows, h, p > 0.f ? seed : nullptr, p > 0.f ? dr_thresh(p) : 0u, 1.f / (1.f - p));
  return (int)hipGetLastError();
}

extern "C" int mt_reglu_bwd_launch(const void* u, const void* dout, void* du, long rows, int h, const int64_t* seed,
                                   float p, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (rg_check(u, dout, rows, h, p) || !du) return -2;
  const long n = rows * (h >> 3);
  hipLaunchKernelGGL(mt_reglu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint16_t*)u,
                     (const uint16_t*)dout, (uint16_t*)du, rows, h, p > 0.f ? seed : nullptr, p > 0.f ? dr_thresh(p) : 0u,
                     1.f / (1.f - p));
  return (int)hipGetLastError();
}
