// Tile body of the grouped exact-f32 GEMM (gfx950): LDS images, LDS-DMA staging, the MFMA stage and one T x T output tile
// with its three epilogues.  Shared by the grouped launches (gemm_f32.hip) and by the resident epoch kernel
// (resident.hip), which walks a slot's whole training epoch inside one launch and calls the same tile code -- a
// trial's numbers are bit for bit the same on either path.
#pragma once
#include "mt_bodies.h"
#define F32_BK 32
#define F32_THREADS 256

static __device__ __attribute__((aligned(16))) uint32_t mt_zero_chunk_f32[4];     // source of every out-of-range chunk

__device__ __forceinline__ int f32_kmaj_off(int row, int chunk) {
  return row * 128 + ((chunk ^ (row & 7)) << 4);
}
template <int T>
__device__ __forceinline__ int f32_mnmaj_off(int row, int chunk) {
  return row * (T * 4) + ((chunk ^ (((row >> 2) & 1) << 2)) << 4);
}

// ---- global -> LDS by LDS-DMA: T/8 pieces of 1 KiB per image, T/32 of them per wave ------------------------------
template <int T, bool KMAJOR>
__device__ __forceinline__ void f32_stage_tile(uint8_t* img, const float* __restrict__ base, int ld, int mn0, int mn_lim,
                                               int k0, int k_lim, int wave, int lane) {
  constexpr int PER_WAVE = T / 32;
#pragma unroll
  for (int t = 0; t < PER_WAVE; ++t) {
    const int j = wave * PER_WAVE + t;
    int row, col;
    bool ok;
    if (KMAJOR) {      // 8 image rows per piece, 8 chunks per row
      const int r = j * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (r & 7);
      row = mn0 + r;
      col = k0 + c * 4;
      ok = (row < mn_lim) && (col < k_lim);
    } else {           // 256/T image rows per piece, T/4 chunks per row
      constexpr int CPR = T / 4;
      const int r = j * (256 / T) + lane / CPR;
      const int c = (lane % CPR) ^ (((r >> 2) & 1) << 2);
      row = k0 + r;
      col = mn0 + c * 4;
      ok = (row < k_lim) && (col < mn_lim);
    }
    const float* src = ok ? base + (size_t)row * ld + col : reinterpret_cast<const float*>(mt_zero_chunk_f32);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(img + j * 1024), 16, 0, 0);
  }
}

// All MFMAs of one 32-deep stage for this wave's (T/2) x (T/2) sub-tile.
template <int T, bool A_KMAJ, bool B_KMAJ>
__device__ __forceinline__ void f32_mma_stage(f32x4 (&acc)[T / 32][T / 32], const uint8_t* ldsA, const uint8_t* ldsB,
                                              int wr, int wc, int lane) {
  constexpr int R = T / 32, W = T / 2;
  const int g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    f32x4 a_[R], b_[R];
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const int mn = wr * W + m * 16 + i;
      if constexpr (A_KMAJ) {
        a_[m] = *reinterpret_cast<const f32x4*>(ldsA + f32_kmaj_off(mn, ks * 4 + g));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          a_[m][e] = *reinterpret_cast<const float*>(ldsA + f32_mnmaj_off<T>(ks * 16 + g * 4 + e, mn >> 2) + ((mn & 3) << 2));
      }
    }
#pragma unroll
    for (int n = 0; n < R; ++n) {
      const int mn = wc * W + n * 16 + i;
      if constexpr (B_KMAJ) {
        b_[n] = *reinterpret_cast<const f32x4*>(ldsB + f32_kmaj_off(mn, ks * 4 + g));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          b_[n][e] = *reinterpret_cast<const float*>(ldsB + f32_mnmaj_off<T>(ks * 16 + g * 4 + e, mn >> 2) + ((mn & 3) << 2));
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < R; ++m)
#pragma unroll
        for (int n = 0; n < R; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(b_[n][e], a_[m][e], acc[m][n], 0, 0, 0);
  }
}

// The same stage with bf16 PRODUCTS (f32 operands in memory and LDS, f32 accumulation): the eight values a lane holds
// per 16-row block and stage (kk = 4 g + e and 16 + 4 g + e) are rounded to bf16 and go through ONE
// v_mfma_f32_16x16x32_bf16 -- the instruction contracts 32 kk in whatever order its two operands agree on -- in place of
// eight v_mfma_f32_16x16x4_f32, at a sixteenth of their issue cycles.  The arithmetic of bf16 autocast with f32 outputs.
template <int T, bool A_KMAJ, bool B_KMAJ>
__device__ __forceinline__ void f32_mma_stage_bf16(f32x4 (&acc)[T / 32][T / 32], const uint8_t* ldsA, const uint8_t* ldsB,
                                                   int wr, int wc, int lane) {
  constexpr int R = T / 32, W = T / 2;
  const int g = lane >> 4, i = lane & 15;
  bf16x8 af[R], bfr[R];
#pragma unroll
  for (int m = 0; m < R; ++m) {
    const int mn = wr * W + m * 16 + i;
    f32x4 v[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (A_KMAJ) {
        v[ks] = *reinterpret_cast<const f32x4*>(ldsA + f32_kmaj_off(mn, ks * 4 + g));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[ks][e] = *reinterpret_cast<const float*>(ldsA + f32_mnmaj_off<T>(ks * 16 + g * 4 + e, mn >> 2) + ((mn & 3) << 2));
      }
    }
    const u32x4 pk = {pack_bf16x2(v[0][0], v[0][1]), pack_bf16x2(v[0][2], v[0][3]), pack_bf16x2(v[1][0], v[1][1]),
                      pack_bf16x2(v[1][2], v[1][3])};
    af[m] = __builtin_bit_cast(bf16x8, pk);
  }
#pragma unroll
  for (int n = 0; n < R; ++n) {
    const int mn = wc * W + n * 16 + i;
    f32x4 v[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (B_KMAJ) {
        v[ks] = *reinterpret_cast<const f32x4*>(ldsB + f32_kmaj_off(mn, ks * 4 + g));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[ks][e] = *reinterpret_cast<const float*>(ldsB + f32_mnmaj_off<T>(ks * 16 + g * 4 + e, mn >> 2) + ((mn & 3) << 2));
      }
    }
    const u32x4 pk = {pack_bf16x2(v[0][0], v[0][1]), pack_bf16x2(v[0][2], v[0][3]), pack_bf16x2(v[1][0], v[1][1]),
                      pack_bf16x2(v[1][2], v[1][3])};
    bfr[n] = __builtin_bit_cast(bf16x8, pk);
  }
#pragma unroll
  for (int m = 0; m < R; ++m)
#pragma unroll
    for (int n = 0; n < R; ++n)
      acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[n], af[m], acc[m][n], 0, 0, 0);
}

// Column sums of the MN-major A image (bias gradient): thread -> column (tid % T); TWO threads share a column (16 kk
// rows each) for either tile edge, so the summation order -- and with it a trial's bias gradient, bit for bit -- does
// not depend on which tile edge the size of its pack selected.  Threads >= 2T sit this out.
#define F32_DB_PARTS 2
template <int T>
__device__ __forceinline__ float f32_colsum_stage(const uint8_t* ldsA, int tid) {
  constexpr int ROWS = F32_BK / F32_DB_PARTS;
  if (tid >= F32_DB_PARTS * T) return 0.f;
  const int col = tid % T, part = tid / T;
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int row = part * ROWS + r;
    s += *reinterpret_cast<const float*>(ldsA + f32_mnmaj_off<T>(row, col >> 2) + ((col & 3) << 2));
  }
  return s;
}

template <int N>
__device__ __forceinline__ void f32_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a six-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Wait until at most `ahead` K-tiles (of DMA LDS-DMA instructions each; loads return in order) are still in flight;
// `ahead` is wave-uniform and at most MAXA.
template <int DMA, int MAXA>
__device__ __forceinline__ void f32_wait_ahead(int ahead) {
  if constexpr (MAXA == 0) {
    f32_wait_vm<0>();
  } else {
    if (ahead >= MAXA) f32_wait_vm<DMA * MAXA>();
    else f32_wait_ahead<DMA, MAXA - 1>(ahead);
  }
}

// One T x T output tile.  ksplits > 1 (wgrad only): tile.ti carries the split index in bits 8..14.
// PROD: 0 = every product an f32 fused multiply-add (v_mfma_f32_16x16x4_f32); 1 = operands rounded to bf16 for the matrix
// instruction (f32_mma_stage_bf16).
template <int MODE, int NSTAGE, int T, int KU = 1, int PROD = 0>
__device__ __forceinline__ void f32_gemm_tile(const MtGemmGroup* __restrict__ groups, const MtTile tile, int step,
                                              int flags, int arow0, int i_override, int ksplits, size_t split_stride,
                                              uint8_t* lds) {
  constexpr bool A_KMAJ = (MODE != 2);
  constexpr bool B_KMAJ = (MODE == 0);
  constexpr int R = T / 32, W = T / 2;
  constexpr int IMG = T * 128, STAGE = 2 * IMG;
  constexpr int DMA_PER_TILE = 2 * (T / 32);       // LDS-DMA instructions a wave issues per K-step
  if (tile.group < 0) return;   // padding entry of an XCD-balanced tile list
  MtGemmGroup g = groups[tile.group];
  if (step >= g.active_steps) return;
  if (i_override > 0) {   // evaluation: every group reads rows [arow0, arow0 + i_override) of a shared A
    g.I = i_override;
    g.A = reinterpret_cast<const float*>(g.A) + (size_t)arow0 * g.lda;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int split = (MODE == 2) ? (tile.ti >> 8) : 0;
  const int i0 = (MODE == 2 ? (tile.ti & 0xFF) : (int)tile.ti) * T, j0 = tile.tj * T;   // only wgrad tiles carry a split index
  if (i0 >= g.I || j0 >= g.J) return;
  const float* A = reinterpret_cast<const float*>(g.A);
  const float* B = reinterpret_cast<const float*>(g.B);

  f32x4 acc[R][R];
#pragma unroll
  for (int m = 0; m < R; ++m)
#pragma unroll
    for (int n = 0; n < R; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool want_db = (MODE == 2) && (flags & MT_GEMM_FLAG_DBIAS) && tile.tj == 0 && g.aux_out != nullptr;
  float db_acc = 0.f;

  const int nk_all = (g.KK + F32_BK - 1) / F32_BK;
  int kt_lo = 0, kt_hi = nk_all;
  if (MODE == 2 && g.site > 0) ksplits = g.site;      // the group's own split count (launches that mix layers)
  if (MODE == 2 && ksplits > 1) {
    const int per = (nk_all + ksplits - 1) / ksplits;
    kt_lo = split * per;
    kt_hi = (kt_lo + per < nk_all) ? kt_lo + per : nk_all;
    if (kt_lo > kt_hi) kt_lo = kt_hi;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int n_it = kt_hi - kt_lo;
  auto issue = [&](int slot, int kt) {
    f32_stage_tile<T, A_KMAJ>(lds + slot * STAGE, A, g.lda, i0, g.I, kt * F32_BK, g.KK, wave_u, lane);
    f32_stage_tile<T, B_KMAJ>(lds + slot * STAGE + IMG, B, g.ldb, j0, g.J, kt * F32_BK, g.KK, wave_u, lane);
  };

  if constexpr (NSTAGE == 2) {
    // Two stages, two barriers per K-step: tile it+2 refills the stage tile `it` just vacated.
    if (n_it > 0) issue(0, kt_lo);
    if (n_it > 1) issue(1, kt_lo + 1);
    for (int it = 0; it < n_it; ++it) {
      const int cur = it & 1;
      const uint8_t* ldsA = lds + cur * STAGE;
      if (it + 1 < n_it) f32_wait_vm<DMA_PER_TILE>();
      else f32_wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if constexpr (PROD == 1) f32_mma_stage_bf16<T, A_KMAJ, B_KMAJ>(acc, ldsA, ldsA + IMG, wr, wc, lane);
      else f32_mma_stage<T, A_KMAJ, B_KMAJ>(acc, ldsA, ldsA + IMG, wr, wc, lane);
      if (want_db) db_acc += f32_colsum_stage<T>(ldsA, tid);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (it + 2 < n_it) issue(cur, kt_lo + it + 2);
    }
  } else {
    // NSTAGE (3 or more) stages, one barrier per K-step: it publishes tile `it` and proves everybody is done with tile
    // it-1, whose stage is refilled right away with tile it + NSTAGE - 1.  Deep forms (6-8 stages of 32-tiles, 4 of
    // 64-tiles) are for launches of a few tiles per CU: with nothing else resident to hide it, a K-step costs one memory
    // latency unless its operands were asked for that many steps earlier.
    // KU > 1: KU K-steps share one barrier (they are waited for, published and computed together), which divides the
    // barrier and LDS round trips of a short tile by KU.
    static_assert(NSTAGE >= 3 && NSTAGE >= 2 * KU && DMA_PER_TILE * (NSTAGE - 2 * KU) < 64,
                  "one-barrier form: three stages or more, room for two groups of K-steps, vmcnt in range");
#pragma unroll
    for (int p = 0; p < NSTAGE - KU; ++p)
      if (p < n_it) issue(p, kt_lo + p);
    for (int it = 0; it < n_it; it += KU) {
      const int issued = (it + NSTAGE - KU < n_it) ? it + NSTAGE - KU : n_it;
      const int need = (it + KU < n_it) ? it + KU : n_it;
      f32_wait_ahead<DMA_PER_TILE, NSTAGE - 2 * KU>(issued - need);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS reads of the previous K-steps are back
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int nxt = it + NSTAGE - KU + u;
        if (nxt < n_it) issue(nxt % NSTAGE, kt_lo + nxt);
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        if (it + u < n_it) {
          const uint8_t* ldsA = lds + ((it + u) % NSTAGE) * STAGE;
          f32_mma_stage<T, A_KMAJ, B_KMAJ>(acc, ldsA, ldsA + IMG, wr, wc, lane);
          if (want_db) db_acc += f32_colsum_stage<T>(ldsA, tid);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // the bias-gradient reduction reuses stage 0
  }

  // ------------------------------------------------------------------------------ epilogues
  const int fq = lane >> 4, fr = lane & 15;
  if (MODE == 2) {
    float* C = reinterpret_cast<float*>(g.C) + (size_t)split * split_stride;
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const int row = i0 + wr * W + m * 16 + fr;
#pragma unroll
      for (int n = 0; n < R; ++n) {
        const int col = j0 + wc * W + n * 16 + fq * 4;
        if (row < g.I) {
          float* dst = C + (size_t)row * g.ldc + col;
          if (col + 3 < g.J) *reinterpret_cast<f32x4*>(dst) = acc[m][n];
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (col + j < g.J) dst[j] = acc[m][n][j];
          }
        }
      }
    }
    if (want_db) {
      float* red = reinterpret_cast<float*>(lds);    // all waves are past the last K-step barrier
      red[tid] = db_acc;
      __syncthreads();
      if (tid < T && i0 + tid < g.I)
        (reinterpret_cast<float*>(g.aux_out) + (size_t)split * split_stride)[i0 + tid] = red[tid] + red[tid + T];
    }
    return;
  }

  const bool relu = (MODE == 0) && ((flags & MT_GEMM_FLAG_RELU) || (g.gflags & MT_GROUP_ACT));
  const bool add = (MODE == 1) && (g.gflags & MT_GROUP_ADD) && g.aux != nullptr;
  const bool masked = (MODE == 1) && !add && ((flags & MT_GEMM_FLAG_MASK) || (g.gflags & MT_GROUP_ACT)) && g.aux != nullptr;
  const float mscale = (g.scale != 0.f) ? g.scale : 1.f;
  const bool drop = (MODE == 0) && g.thresh != 0u && g.seed != nullptr;
  uint32_t s0 = 0u, s1 = 0u;
  if (drop) mt_site_seed(g.seed, g.site, step, s0, s1);
  float* C = reinterpret_cast<float*>(g.C);
#pragma unroll
  for (int n = 0; n < R; ++n) {
    const int col = j0 + wc * W + n * 16 + fq * 4;
    f32x4 bb = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0 && g.aux != nullptr) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (col + j < g.J) bb[j] = reinterpret_cast<const float*>(g.aux)[col + j];
    }
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const int row = i0 + wr * W + m * 16 + fr;
      if (row >= g.I || col >= g.J) continue;
      f32x4 v = acc[m][n] + bb;
      if (relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (drop) {        // dropout behind the activation: element index = position in the [I][ldc] output
        const uint32_t kw = mt_keep_word4s(s0, s1, g.thresh, (uint32_t)row * (uint32_t)g.ldc + (uint32_t)col);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((kw >> (8 * j)) & 0xFFu) ? v[j] * mscale : 0.f;
      }
      if (masked) {      // ReLU (+ dropout) backward: zero where the forward output was not positive
        const f32x4 h = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.aux) + (size_t)row * g.ldaux + col);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = h[j] > 0.f ? v[j] * mscale : 0.f;
      }
      if (add) v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.aux) + (size_t)row * g.ldaux + col);
      // Columns >= J inside the last 4-chunk are products with zero-filled operand rows (and zero bias): they are
      // written as zeros, which keeps the pad columns of activation rows zero.
      *reinterpret_cast<f32x4*>(C + (size_t)row * g.ldc + col) = v;
    }
  }
}
