// Whole-model optimiser step over one flat f32 parameter slab, with every hyper-parameter and the step
// counter read from DEVICE memory (gfx950).
//
// A training step captured in a hipGraph replays the exact kernel arguments it was captured with, so a
// host-side step number or learning rate would be frozen at capture time.  Here the kernel arguments are
// only pointers: lr / betas / schedule / step live in one MtFlatOpt record on the device, the update kernel
// reads them, and a one-thread kernel advances the counter after the update.  Changing the learning rate
// between replays is a 4-byte copy into that record; no re-capture.
//
// The same pass zeroes the gradient slab after reading it (autograd accumulates into views of that slab on
// the next backward), so a separate zero_grad pass never runs.  Global-norm clipping is a two-launch
// reduction with per-block partials added in index order (bitwise reproducible, no atomics).
#include "mt_common.h"

#define FOPT_THREADS 256
#define FOPT_PER_BLOCK 2048

struct MtFlatOpt {
  float lr, beta1, beta2, eps, wd, momentum, clip;
  int32_t opt_type;        // 0 adam, 1 adamw, 2 sgd, 3 rmsprop
  int32_t warmup, total;   // linear warm-up then linear decay over optimiser steps (0 = off)
  int32_t step;            // optimiser steps taken so far
  int32_t pad;
};

__global__ __launch_bounds__(FOPT_THREADS) void mt_flat_sqnorm_kernel(const float* __restrict__ G, long n,
                                                                      float* __restrict__ partials) {
  const long base = (long)blockIdx.x * FOPT_PER_BLOCK + threadIdx.x * 8;
  float acc = 0.f;
  if (base + 8 <= n) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(G + base), b = *reinterpret_cast<const f32x4*>(G + base + 4);
    acc = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
  } else {
    for (long e = base; e < n; ++e) acc += G[e] * G[e];
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float flat_lr_factor(int t, int warmup, int total) {
  if (warmup > 0 && t < warmup) return (float)(t + 1) / (float)warmup;
  if (total > 0) {
    const int span = (total - warmup) > 1 ? (total - warmup) : 1;
    const float f = (float)(total - t) / (float)span;
    return f > 0.f ? f : 0.f;
  }
  return 1.f;
}

__global__ __launch_bounds__(FOPT_THREADS) void mt_flat_optim_kernel(float* __restrict__ P, float* __restrict__ G,
                                                                     float* __restrict__ M, float* __restrict__ V, long n,
                                                                     const MtFlatOpt* __restrict__ st,
                                                                     const float* __restrict__ partials, int nb,
                                                                     int zero_grad) {
  const MtFlatOpt h = *st;
  float gscale = 1.f;
  if (h.clip > 0.f && partials != nullptr) {
    float tot = 0.f;
    for (int i = 0; i < nb; ++i) tot += partials[i];
    const float c = h.clip / (sqrtf(tot) + 1e-6f);
    gscale = c < 1.f ? c : 1.f;
  }
  const long base = (long)blockIdx.x * FOPT_PER_BLOCK + threadIdx.x * 8;
  if (base >= n) return;
  const int cnt = (n - base) < 8 ? (int)(n - base) : 8;
  const int t = h.step;
  const float lr = h.lr * flat_lr_factor(t, h.warmup, h.total);
  const float bc1 = 1.f - powf(h.beta1, (float)(t + 1));
  const float inv_bc2_sqrt = 1.f / sqrtf(1.f - powf(h.beta2, (float)(t + 1)));
  float p[8], g[8], m[8], v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool ok = e < cnt;
    p[e] = ok ? P[base + e] : 0.f;
    g[e] = ok ? G[base + e] : 0.f;
    m[e] = ok ? M[base + e] : 0.f;
    v[e] = ok ? V[base + e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float gg = g[e] * gscale;
    switch (h.opt_type) {
      case 0:
      case 1: {
        if (h.opt_type == 0) gg += h.wd * p[e];
        else p[e] *= (1.f - lr * h.wd);
        m[e] = h.beta1 * m[e] + (1.f - h.beta1) * gg;
        v[e] = h.beta2 * v[e] + (1.f - h.beta2) * gg * gg;
        p[e] -= (lr / bc1) * (m[e] / (sqrtf(v[e]) * inv_bc2_sqrt + h.eps));
      } break;
      case 2: {
        gg += h.wd * p[e];
        m[e] = (t == 0 || h.momentum <= 0.f) ? gg : h.momentum * m[e] + gg;   // first step seeds the buffer
        p[e] -= lr * m[e];
      } break;
      default: {
        gg += h.wd * p[e];
        v[e] = 0.99f * v[e] + 0.01f * gg * gg;
        const float avg = sqrtf(v[e]) + h.eps;
        if (h.momentum > 0.f) { m[e] = h.momentum * m[e] + gg / avg; p[e] -= lr * m[e]; }
        else p[e] -= lr * gg / avg;
      } break;
    }
  }
  for (int e = 0; e < cnt; ++e) {
    P[base + e] = p[e]; M[base + e] = m[e]; V[base + e] = v[e];
    if (zero_grad) G[base + e] = 0.f;
  }
}

__global__ void mt_flat_advance_kernel(MtFlatOpt* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) st->step = st->step + 1;
}

// ------------------------------------------------------------------------------------------------------------
// Scattered gradients.  With gradients pre-bound to views of a slab, autograd ADDS into them: one extra kernel per
// parameter per step.  Left unbound, autograd hands each parameter the tensor its backward op produced -- inside a
// captured graph always at the same address -- and these kernels read the gradients where they are, through a
// table of segments (one per block of up to 2048 elements): {offset in the parameter slab, gradient pointer, count}.
struct MtGradSeg {
  long off;
  const float* g;      // null: the parameter received no gradient this step (treated as zero)
  int32_t cnt;
  int32_t pad;
};

__global__ __launch_bounds__(FOPT_THREADS) void mt_flat_sqnorm_seg_kernel(const MtGradSeg* __restrict__ segs,
                                                                          float* __restrict__ partials) {
  const MtGradSeg sg = segs[blockIdx.x];
  const int local = threadIdx.x * 8;
  float acc = 0.f;
  if (sg.g != nullptr)
    for (int e = local; e < local + 8 && e < sg.cnt; ++e) acc += sg.g[e] * sg.g[e];
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(FOPT_THREADS) void mt_flat_optim_seg_kernel(float* __restrict__ P, float* __restrict__ M,
                                                                         float* __restrict__ V,
                                                                         const MtGradSeg* __restrict__ segs,
                                                                         const MtFlatOpt* __restrict__ st,
                                                                         const float* __restrict__ partials, int nb) {
  const MtFlatOpt h = *st;
  const MtGradSeg sg = segs[blockIdx.x];
  float gscale = 1.f;
  if (h.clip > 0.f && partials != nullptr) {
    float tot = 0.f;
    for (int i = 0; i < nb; ++i) tot += partials[i];
    const float c = h.clip / (sqrtf(tot) + 1e-6f);
    gscale = c < 1.f ? c : 1.f;
  }
  const int local = threadIdx.x * 8;
  if (local >= sg.cnt) return;
  const int cnt = (sg.cnt - local) < 8 ? (sg.cnt - local) : 8;
  const int t = h.step;
  const float lr = h.lr * flat_lr_factor(t, h.warmup, h.total);
  const float bc1 = 1.f - powf(h.beta1, (float)(t + 1));
  const float inv_bc2_sqrt = 1.f / sqrtf(1.f - powf(h.beta2, (float)(t + 1)));
  const long base = sg.off + local;
  for (int e = 0; e < cnt; ++e) {
    float p = P[base + e], m = M[base + e], v = V[base + e];
    float gg = (sg.g != nullptr ? sg.g[local + e] : 0.f) * gscale;
    switch (h.opt_type) {
      case 0:
      case 1: {
        if (h.opt_type == 0) gg += h.wd * p;
        else p *= (1.f - lr * h.wd);
        m = h.beta1 * m + (1.f - h.beta1) * gg;
        v = h.beta2 * v + (1.f - h.beta2) * gg * gg;
        p -= (lr / bc1) * (m / (sqrtf(v) * inv_bc2_sqrt + h.eps));
      } break;
      case 2: {
        gg += h.wd * p;
        m = (t == 0 || h.momentum <= 0.f) ? gg : h.momentum * m + gg;
        p -= lr * m;
      } break;
      default: {
        gg += h.wd * p;
        v = 0.99f * v + 0.01f * gg * gg;
        const float avg = sqrtf(v) + h.eps;
        if (h.momentum > 0.f) { m = h.momentum * m + gg / avg; p -= lr * m; }
        else p -= lr * gg / avg;
      } break;
    }
    P[base + e] = p; M[base + e] = m; V[base + e] = v;
  }
}

extern "C" int mt_grad_seg_sizeof() { return (int)sizeof(MtGradSeg); }

// segs: n_segs device-resident MtGradSeg; partials: f32[n_segs] when use_clip.
extern "C" int mt_flat_optim_seg_launch(float* P, float* M, float* V, const MtGradSeg* segs, int n_segs, MtFlatOpt* st,
                                        float* partials, int use_clip, hipStream_t stream) {
  if (n_segs <= 0) return 0;
  if (use_clip) {
    if (partials == nullptr) return -3;
    hipLaunchKernelGGL(mt_flat_sqnorm_seg_kernel, dim3(n_segs), dim3(FOPT_THREADS), 0, stream, segs, partials);
  }
  hipLaunchKernelGGL(mt_flat_optim_seg_kernel, dim3(n_segs), dim3(FOPT_THREADS), 0, stream, P, M, V, segs, st,
                     use_clip ? partials : nullptr, n_segs);
  hipLaunchKernelGGL(mt_flat_advance_kernel, dim3(1), dim3(64), 0, stream, st);
  return (int)hipGetLastError();
}

extern "C" int mt_flat_opt_sizeof() { return (int)sizeof(MtFlatOpt); }

extern "C" int mt_flat_optim_blocks(long n) { return (int)((n + FOPT_PER_BLOCK - 1) / FOPT_PER_BLOCK); }

// partials: f32[mt_flat_optim_blocks(n)] when use_clip, else may be null.
extern "C" int mt_flat_optim_launch(float* P, float* G, float* M, float* V, long n, MtFlatOpt* st, float* partials,
                                    int use_clip, int zero_grad, hipStream_t stream) {
  if (n <= 0) return 0;
  const int nb = (int)((n + FOPT_PER_BLOCK - 1) / FOPT_PER_BLOCK);
  if (use_clip) {
    if (partials == nullptr) return -3;
    hipLaunchKernelGGL(mt_flat_sqnorm_kernel, dim3(nb), dim3(FOPT_THREADS), 0, stream, G, n, partials);
  }
  hipLaunchKernelGGL(mt_flat_optim_kernel, dim3(nb), dim3(FOPT_THREADS), 0, stream, P, G, M, V, n, st,
                     use_clip ? partials : nullptr, nb, zero_grad);
  hipLaunchKernelGGL(mt_flat_advance_kernel, dim3(1), dim3(64), 0, stream, st);
  return (int)hipGetLastError();
}
