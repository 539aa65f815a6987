// Grouped exact-f32 GEMM on MFMA for packed multi-trial MLP training (gfx950).
//
// The f32 twin of gemm.hip: same descriptors, same three storage modes (NT forward, NN input gradient, TN weight
// gradient), same tile lists, but every operand, activation and gradient is f32 and the products run on
// v_mfma_f32_16x16x4_f32, which is bit-for-bit a k-ordered chain of f32 fused multiply-adds (no reduced-precision
// input path exists on this chip).  This is the engine behind PackedTrainable(dtype="f32"): the same arithmetic as an
// eager f32 PyTorch training loop (/root/reference/ray-tune-hpo-regression-sample.py:104-119), only batched over the
// trial slots of a pack.
//
// The f32 matrix instruction runs at 1/16 of the bf16 rate, so these kernels are bound by the matrix pipe, not by
// staging: per 32-deep K-step a wave issues 128 MFMAs (4096 cycles) for 32 KB of LDS-DMA per workgroup.  Two things
// follow.  (1) Tiles can be small without losing anything: the launcher offers a 64x64 tile (four waves of 32x32)
// besides the 128x128 one, and the engine picks 64 whenever a launch would otherwise have too few tiles to put
// two workgroups on every CU -- the late rounds of a search, where a dozen survivors are left.  (2) Two workgroups
// per CU (two waves per SIMD) matter more than pipeline depth: the partner wave's MFMAs fill the bubbles of LDS
// reads and barriers.
//
// Tile T x T (T = 128 or 64), BK = 32, 256 threads = 2 x 2 waves, each wave (T/2) x (T/2) as (T/32)^2 accumulators.
// Operand images in LDS (T*128 bytes each):
//   K-major  [T rows][32 kk] f32: 128-B rows, 16-B chunk index XORed with (row & 7); a lane reads ONE ds_read_b128
//            = 4 consecutive kk of its row, which feed four successive MFMAs;
//   MN-major [32 kk][T cols] f32: rows of T*4 bytes, chunk index XORed with ((row >> 2) & 1) << 2; a lane reads four
//            ds_read_b32 (one per MFMA), 16 lanes covering 16 consecutive columns of one kk row.
// MFMA number (ks, e), ks = 0..1, e = 0..3, of lane group g = lane >> 4 contracts kk = 16 ks + 4 g + e for BOTH
// operands, whatever their storage form -- any fixed permutation of kk inside a K-step is legal as long as the two
// operands agree on it.  Operand roles are swapped as in gemm.hip so that a lane's accumulator holds four consecutive
// output columns of one row (16-byte epilogue stores).
#include "mt_bodies.h"

#define F32_BK 32
#define F32_THREADS 256

__device__ __attribute__((aligned(16))) uint32_t mt_zero_chunk_f32[4];     // source of every out-of-range chunk

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
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "unsupported vmcnt");
}

// One T x T output tile.  ksplits > 1 (wgrad only): tile.ti carries the split index in bits 8..14.
template <int MODE, int NSTAGE, int T>
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
      f32_mma_stage<T, A_KMAJ, B_KMAJ>(acc, ldsA, ldsA + IMG, wr, wc, lane);
      if (want_db) db_acc += f32_colsum_stage<T>(ldsA, tid);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (it + 2 < n_it) issue(cur, kt_lo + it + 2);
    }
  } else {
    // NSTAGE (3) stages, one barrier per K-step: it publishes tile `it` and proves everybody is done with tile
    // it-1, whose stage is refilled right away with tile it + NSTAGE - 1.
    static_assert(NSTAGE == 3, "one-barrier form is built for three stages");
#pragma unroll
    for (int p = 0; p < NSTAGE - 1; ++p)
      if (p < n_it) issue(p, kt_lo + p);
    for (int it = 0; it < n_it; ++it) {
      if (n_it - 1 - it >= 1) f32_wait_vm<DMA_PER_TILE>();
      else f32_wait_vm<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS reads of tile it-1 are back
      __builtin_amdgcn_s_barrier();
      const int nxt = it + NSTAGE - 1;
      if (nxt < n_it) issue(nxt % NSTAGE, kt_lo + nxt);
      const uint8_t* ldsA = lds + (it % NSTAGE) * STAGE;
      f32_mma_stage<T, A_KMAJ, B_KMAJ>(acc, ldsA, ldsA + IMG, wr, wc, lane);
      if (want_db) db_acc += f32_colsum_stage<T>(ldsA, tid);
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

template <int MODE, int NSTAGE, int T>
__global__ __launch_bounds__(F32_THREADS) void mt_gemm_f32_kernel(const MtGemmGroup* __restrict__ groups,
                                                                  const MtTile* __restrict__ tiles, int step, int flags,
                                                                  int arow0, int i_override, int ksplits,
                                                                  size_t split_stride, const int* __restrict__ step_base) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[NSTAGE * 2 * T * 128];
  f32_gemm_tile<MODE, NSTAGE, T>(groups, tiles[blockIdx.x], mt_step(step, step_base), flags, arow0, i_override, ksplits,
                                 split_stride, lds);
}

// Backward of one layer in ONE launch (weight-gradient tiles, input-gradient tiles, and optionally the head
// kernel's partial-record reduction as extra workgroups), as in gemm.hip.
template <int NSTAGE, int T>
__global__ __launch_bounds__(F32_THREADS) void mt_gemm_f32_bwd_kernel(const MtGemmGroup* __restrict__ wgroups,
                                                                      const MtTile* __restrict__ wtiles, int n_wtiles,
                                                                      const MtGemmGroup* __restrict__ dgroups,
                                                                      const MtTile* __restrict__ dtiles, int n_dtiles,
                                                                      int step_imm, int ksplits, size_t split_stride,
                                                                      const int* __restrict__ step_base,
                                                                      MtHeadReduceArgs hr) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[NSTAGE * 2 * T * 128];
  const int step = mt_step(step_imm, step_base);
  const int b = (int)blockIdx.x;
  if (b < n_wtiles)
    f32_gemm_tile<2, NSTAGE, T>(wgroups, wtiles[b], step, MT_GEMM_FLAG_DBIAS, 0, 0, ksplits, split_stride, lds);
  else if (b < n_wtiles + n_dtiles)
    f32_gemm_tile<1, NSTAGE, T>(dgroups, dtiles[b - n_wtiles], step, MT_GEMM_FLAG_MASK, 0, 0, 1, 0, lds);
  else
    mt_head_reduce_body(hr.slots, hr.ids, step, hr.scratch, hr.max_chunks, hr.rec, b - n_wtiles - n_dtiles);
}

// tile_edge: 128 or 64 (both on the two-stage, two-barrier pipeline: 32 KB of LDS per 64-tile workgroup, so five of
// them fit a CU and a launch of four tiles per CU has no second round; measured 7-22 % faster than the three-stage,
// one-barrier form, which 65 still selects for comparison); the tile list must have been built with the same edge.
extern "C" int mt_gemm_f32_launch(int mode, const MtGemmGroup* groups, const MtTile* tiles, int ntiles, int step,
                                  int flags, int arow0, int i_override, int ksplits, size_t split_stride, int tile_edge,
                                  hipStream_t stream) {
  if (ntiles <= 0) return 0;
  if (ksplits < 1 || ksplits > 127 || (mode != 2 && ksplits != 1)) return -5;
  if (tile_edge != 128 && tile_edge != 64 && tile_edge != 65 && tile_edge != 32) return -7;      // 65: 64-tiles, three stages
  dim3 grid(ntiles), block(F32_THREADS);
#define MT_F32_GO(M, N, T_, KS, SS)                                                                                       \
  hipLaunchKernelGGL((mt_gemm_f32_kernel<M, N, T_>), grid, block, 0, stream, groups, tiles, step, flags, arow0, i_override, \
                     KS, SS, mt_step_base_tls)
  const bool big = tile_edge == 128, two = tile_edge == 64;
  if (tile_edge == 32) {        // latency form: launches of a few hundred 64-tiles fill the chip four times better
    switch (mode) {
      case 0: MT_F32_GO(0, 2, 32, 1, (size_t)0); break;
      case 1: MT_F32_GO(1, 2, 32, 1, (size_t)0); break;
      case 2: MT_F32_GO(2, 2, 32, ksplits, split_stride); break;
      default: return -1;
    }
    return (int)hipGetLastError();
  }
  switch (mode) {
    case 0: if (big) MT_F32_GO(0, 2, 128, 1, (size_t)0); else if (two) MT_F32_GO(0, 2, 64, 1, (size_t)0); else MT_F32_GO(0, 3, 64, 1, (size_t)0); break;
    case 1: if (big) MT_F32_GO(1, 2, 128, 1, (size_t)0); else if (two) MT_F32_GO(1, 2, 64, 1, (size_t)0); else MT_F32_GO(1, 3, 64, 1, (size_t)0); break;
    case 2: if (big) MT_F32_GO(2, 2, 128, ksplits, split_stride); else if (two) MT_F32_GO(2, 2, 64, ksplits, split_stride); else MT_F32_GO(2, 3, 64, ksplits, split_stride); break;
    default: return -1;
  }
#undef MT_F32_GO
  return (int)hipGetLastError();
}

extern "C" int mt_gemm_f32_bwd_head_launch(const MtGemmGroup* wgroups, const MtTile* wtiles, int n_wtiles,
                                           const MtGemmGroup* dgroups, const MtTile* dtiles, int n_dtiles, int step,
                                           int ksplits, size_t split_stride, int tile_edge, MtHeadReduceArgs hr,
                                           hipStream_t stream) {
  if (n_wtiles + n_dtiles + hr.n_slots <= 0) return 0;
  if (ksplits < 1 || ksplits > 127) return -5;
  if (tile_edge != 128 && tile_edge != 64 && tile_edge != 65 && tile_edge != 32) return -7;
  dim3 grid(n_wtiles + n_dtiles + hr.n_slots), block(F32_THREADS);
  if (tile_edge == 32)
    hipLaunchKernelGGL((mt_gemm_f32_bwd_kernel<2, 32>), grid, block, 0, stream, wgroups, wtiles, n_wtiles, dgroups, dtiles,
                       n_dtiles, step, ksplits, split_stride, mt_step_base_tls, hr);
  else if (tile_edge == 64)
    hipLaunchKernelGGL((mt_gemm_f32_bwd_kernel<2, 64>), grid, block, 0, stream, wgroups, wtiles, n_wtiles, dgroups, dtiles,
                       n_dtiles, step, ksplits, split_stride, mt_step_base_tls, hr);
  else if (tile_edge == 128)
    hipLaunchKernelGGL((mt_gemm_f32_bwd_kernel<2, 128>), grid, block, 0, stream, wgroups, wtiles, n_wtiles, dgroups, dtiles,
                       n_dtiles, step, ksplits, split_stride, mt_step_base_tls, hr);
  else
    hipLaunchKernelGGL((mt_gemm_f32_bwd_kernel<3, 64>), grid, block, 0, stream, wgroups, wtiles, n_wtiles, dgroups, dtiles,
                       n_dtiles, step, ksplits, split_stride, mt_step_base_tls, hr);
  return (int)hipGetLastError();
}

extern "C" int mt_gemm_f32_bwd_launch(const MtGemmGroup* wgroups, const MtTile* wtiles, int n_wtiles,
                                      const MtGemmGroup* dgroups, const MtTile* dtiles, int n_dtiles, int step,
                                      int ksplits, size_t split_stride, int tile_edge, hipStream_t stream) {
  MtHeadReduceArgs none = {nullptr, nullptr, nullptr, 0, 0, 0};
  return mt_gemm_f32_bwd_head_launch(wgroups, wtiles, n_wtiles, dgroups, dtiles, n_dtiles, step, ksplits, split_stride,
                                     tile_edge, none, stream);
}

// ----------------------------------------------------------------------------------------------------
// Stand-alone f32 dense-layer GEMMs (one group by value, tile from blockIdx), the f32 twin of mt_gemm_single_launch in
// gemm.hip: forward (mode 0, bias + optional ReLU), input gradient (mode 1) and weight gradient (mode 2, bias gradient
// = column sums of dY when db is given; contraction over the rows, no split).  64-tiles unless the launch has enough
// 128-tiles to fill the chip twice.
template <int MODE, int NSTAGE, int T>
__global__ __launch_bounds__(F32_THREADS) void mt_gemm_f32_single_kernel(MtGemmGroup g, int tiles_i, int tiles_j, int flags) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[NSTAGE * 2 * T * 128];
  const int b = blockIdx.x;
  const int n = tiles_i * tiles_j;
  const int per = (n + 7) / 8;
  const int t = (b % 8) * per + b / 8;
  if (t >= n) return;
  MtTile tile;
  tile.group = 0;
  tile.ti = (int16_t)(t / tiles_j);
  tile.tj = (int16_t)(t % tiles_j);
  f32_gemm_tile<MODE, NSTAGE, T>(&g, tile, 0, flags, 0, 0, 1, 0, lds);
}

extern "C" int mt_gemm_f32_single_launch(int mode, const void* A, const void* B, void* C, const void* aux, void* aux_out,
                                         int I, int J, int KK, int lda, int ldb, int ldc, int flags, hipStream_t stream) {
  if (I <= 0 || J <= 0 || KK <= 0) return 0;
  if ((lda & 3) || (ldb & 3) || (ldc & 3) || mode < 0 || mode > 2) return -2;
  const int n128 = ((I + 127) / 128) * ((J + 127) / 128);
  const int T = n128 >= 512 ? 128 : 64;
  const int tiles_i = (I + T - 1) / T, tiles_j = (J + T - 1) / T;
  if (tiles_i > (mode == 2 ? 255 : 32767) || tiles_j > 32767) return -2;
  MtGemmGroup g = {};
  g.A = A; g.B = B; g.C = C; g.aux = aux; g.aux_out = aux_out;
  g.I = I; g.J = J; g.KK = KK;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = 0;
  g.active_steps = 1;
  const int nblocks = ((tiles_i * tiles_j + 7) / 8) * 8;
  dim3 grid(nblocks), block(F32_THREADS);
#define MT_F32_SINGLE(M_) \
  do { \
    if (T == 128) hipLaunchKernelGGL((mt_gemm_f32_single_kernel<M_, 2, 128>), grid, block, 0, stream, g, tiles_i, tiles_j, flags); \
    else hipLaunchKernelGGL((mt_gemm_f32_single_kernel<M_, 2, 64>), grid, block, 0, stream, g, tiles_i, tiles_j, flags); \
  } while (0)
  if (mode == 0) MT_F32_SINGLE(0);
  else if (mode == 1) MT_F32_SINGLE(1);
  else MT_F32_SINGLE(2);
#undef MT_F32_SINGLE
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------------------------------
// Weight (+ bias) gradient of ONE f32 dense layer with the contraction over the rows split into `ksplits` slices:
// dW [I x J] = dY^T X is a handful of output tiles with a contraction as long as the batch has token rows, so without a
// split a few workgroups would each walk hundreds of K-steps while the rest of the chip idles.  Slice s writes its
// partial products to arena + s * slab (dW at offset 0, db at offset I * J); mt_slab_sum_kernel adds the slices in
// index order (fixed order: bitwise repeatable, no atomics).
template <int NSTAGE, int T>
__global__ __launch_bounds__(F32_THREADS) void mt_wgrad_f32_single_kernel(MtGemmGroup g, int tiles_i, int tiles_j,
                                                                          int ksplits, size_t slab, int flags) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[NSTAGE * 2 * T * 128];
  const int b = blockIdx.x;
  const int n = tiles_i * tiles_j * ksplits;
  const int per = (n + 7) / 8;
  const int t = (b % 8) * per + b / 8;        // consecutive work items of one XCD share the operands' column blocks
  if (t >= n) return;
  const int split = t % ksplits, tt = t / ksplits;
  MtTile tile;
  tile.group = 0;
  tile.ti = (int16_t)((tt / tiles_j) | (split << 8));
  tile.tj = (int16_t)(tt % tiles_j);
  f32_gemm_tile<2, NSTAGE, T>(&g, tile, 0, flags, 0, 0, ksplits, slab, lds);
}

// Thread -> one float4 column group; the slices are walked with four running sums (slice s goes to sum s % 4) so that
// eight loads are in flight instead of one dependent chain, and the four sums are combined as (0 + 1) + (2 + 3): a fixed
// order, independent of the launch geometry.
__global__ __launch_bounds__(256) void mt_slab_sum_kernel(const float* __restrict__ arena, float* __restrict__ out, long n,
                                                          int ksplits, size_t slab) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 3 < n) {
    f32x4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= ksplits; s += 8) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(arena + (size_t)(s + k) * slab + i);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k & 3] += v[k];
    }
    for (; s < ksplits; ++s) acc[s & 3] += *reinterpret_cast<const f32x4*>(arena + (size_t)s * slab + i);
    *reinterpret_cast<f32x4*>(out + i) = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  } else {
    for (long k = i; k < n; ++k) {
      float a4[4] = {0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < ksplits; ++s) a4[s & 3] += arena[(size_t)s * slab + k];
      out[k] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
  }
}

// out[0 .. n) = sum over s of arena[s * slab + (0 .. n)), slices added in index order (slab and out 16-byte aligned).
extern "C" int mt_slab_sum_launch(const float* arena, float* out, long n, int ksplits, size_t slab, hipStream_t stream) {
  if (n <= 0) return 0;
  if (ksplits < 1 || !arena || !out || (slab & 3) || (((uintptr_t)arena | (uintptr_t)out) & 15)) return -2;
  hipLaunchKernelGGL(mt_slab_sum_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, arena, out, n, ksplits,
                     slab);
  return (int)hipGetLastError();
}

// dY: [M][I] (row stride I), X: [M][J]; arena: ksplits slices of `slab` floats (slab >= I * J + I, a multiple of 4).
// with_bias != 0: column sums of dY go to offset I * J of every slice.
extern "C" int mt_wgrad_f32_single_launch(const void* dY, const void* X, float* arena, int M, int I, int J, int ksplits,
                                          size_t slab, int with_bias, hipStream_t stream) {
  if (M <= 0 || I <= 0 || J <= 0) return 0;
  if ((I & 3) || (J & 3) || ksplits < 1 || ksplits > 127 || (slab & 3) || slab < (size_t)I * J + (with_bias ? I : 0)) return -2;
  constexpr int T = 64;
  const int tiles_i = (I + T - 1) / T, tiles_j = (J + T - 1) / T;
  if (tiles_i > 255 || tiles_j > 32767) return -2;
  MtGemmGroup g = {};
  g.A = dY; g.B = X; g.C = arena; g.aux = nullptr; g.aux_out = with_bias ? arena + (size_t)I * J : nullptr;
  g.I = I; g.J = J; g.KK = M;
  g.lda = I; g.ldb = J; g.ldc = J; g.ldaux = 0;
  g.active_steps = 1;
  const int nblocks = ((tiles_i * tiles_j * ksplits + 7) / 8) * 8;
  hipLaunchKernelGGL((mt_wgrad_f32_single_kernel<2, T>), dim3(nblocks), dim3(F32_THREADS), 0, stream, g, tiles_i, tiles_j,
                     ksplits, slab, with_bias ? MT_GEMM_FLAG_DBIAS : 0);
  return (int)hipGetLastError();
}
