// Grouped bf16 GEMM on MFMA for packed multi-trial MLP training (gfx950).
//
// One launch serves every trial slot of a pack: block b takes tiles[b] = (group, ti, tj) and
// computes one 128x128 output tile of that group's C = A x B with BK = 64.  Four waves (2 x 2),
// each owning a 64x64 sub-tile as 4 x 4 accumulators of v_mfma_f32_16x16x32_bf16.
//
// Operands come in two storage forms and each has its own LDS image and fragment read:
//   K-major  ([row][kk], kk contiguous; X in forward, W in forward, dY in dgrad):
//       image [128 rows][64 kk] bf16, 128-B rows, 16-B chunk index XORed with (row & 7);
//       fragment = one ds_read_b128 per 16x32 block (conflict-free, see chunk math below).
//   MN-major ([kk][col], col contiguous; W in dgrad, dY and X in wgrad):
//       image [64 kk][128 cols] bf16, 256-B rows, chunk index XORed with ((row&3)<<2 | (row>>2)&3);
//       fragment = two ds_read_b64_tr_b16 (hardware transpose: each lane receives 4 consecutive
//       kk of ONE column), so no operand ever needs a transposed copy in HBM.
// Global -> LDS staging goes through registers (16-B loads issued one tile ahead of the MFMAs that
// consume them) because both images are swizzled per row.
//
// Epilogues: forward adds bias and applies ReLU in f32, converts to bf16, transposes the tile
// through LDS and stores 16 B per lane; dgrad does the same and zeroes elements whose forward
// activation was <= 0 (ReLU backward) while storing; wgrad stores f32 and, on the tj == 0 tiles,
// also reduces the bias gradient (column sums of dY) from the LDS image it already holds.
#include "mt_common.h"

#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256

typedef __attribute__((address_space(3))) uint8_t lds_u8;

__device__ __forceinline__ int kmaj_off(int row, int chunk) {       // [128][64] bf16 image
  return row * 128 + ((chunk ^ (row & 7)) << 4);
}
__device__ __forceinline__ int mnmaj_off(int row, int chunk) {      // [64][128] bf16 image
  return row * 256 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}
__device__ __forceinline__ int ctile_off(int row, int chunk) {      // [128][128] bf16 epilogue image
  return row * 256 + ((chunk ^ (((row >> 2) & 3) << 1)) << 4);
}

// ---- global -> register staging (4 x 16 B per thread per operand) ---------------------------------
template <bool KMAJOR>
__device__ __forceinline__ void load_tile(u32x4 (&r)[4], const uint16_t* __restrict__ base, int ld,
                                          int mn0, int mn_lim, int k0, int k_lim, int tid) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int q = tid + NTHREADS * t;
    int row, col;      // row/col in the GLOBAL matrix as stored
    bool ok;
    if (KMAJOR) {      // stored [mn][kk]
      const int rr = q >> 3, c = q & 7;
      row = mn0 + rr;
      col = k0 + c * 8;
      ok = (row < mn_lim) && (col < k_lim);
    } else {           // stored [kk][mn]
      const int rr = q >> 4, c = q & 15;
      row = k0 + rr;
      col = mn0 + c * 8;
      ok = (row < k_lim) && (col < mn_lim);
    }
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ok) v = *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + col);
    r[t] = v;
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void store_tile(const u32x4 (&r)[4], uint8_t* lds, int tid) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int q = tid + NTHREADS * t;
    const int off = KMAJOR ? kmaj_off(q >> 3, q & 7) : mnmaj_off(q >> 4, q & 15);
    *reinterpret_cast<u32x4*>(lds + off) = r[t];
  }
}

// ---- LDS -> MFMA fragments -----------------------------------------------------------------------
// K-major: lane l holds row (l & 15), kk = 32*ks + 8*(l >> 4) + 0..7  -> one 16-B chunk.
__device__ __forceinline__ bf16x8 frag_kmajor(const uint8_t* lds, int row0, int ks, int lane) {
  const int row = row0 + (lane & 15);
  const int chunk = ks * 4 + (lane >> 4);
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds + kmaj_off(row, chunk)));
}
// MN-major: per 16-lane group g the two transposed reads cover kk rows 32*ks + 8g + {0..3} and
// {4..7}; lane 4q+p of the group addresses row q, columns 4p..4p+3 of the 16-column block and
// receives column (l & 15) of the four rows.
__device__ __forceinline__ bf16x8 frag_mnmajor(const uint8_t* lds, int col0, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int c0 = col0 >> 3;
  const int r0 = ks * 32 + g * 8 + q;
  const lds_u8* a0 = (const lds_u8*)(lds + mnmaj_off(r0, c0 + (p >> 1)) + 8 * (p & 1));
  const lds_u8* a1 = (const lds_u8*)(lds + mnmaj_off(r0 + 4, c0 + (p >> 1)) + 8 * (p & 1));
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int MODE>
__global__ __launch_bounds__(NTHREADS) void mt_gemm_kernel(const MtGemmGroup* __restrict__ groups,
                                                           const MtTile* __restrict__ tiles, int step,
                                                           int flags, int arow0, int i_override) {
  constexpr bool A_KMAJ = (MODE != 2);
  constexpr bool B_KMAJ = (MODE == 0);
  __shared__ __attribute__((aligned(16))) uint8_t lds[32768];
  uint8_t* ldsA = lds;
  uint8_t* ldsB = lds + 16384;

  const MtTile tile = tiles[blockIdx.x];
  if (tile.group < 0) return;   // padding entry of an XCD-balanced tile list
  MtGemmGroup g = groups[tile.group];
  if (step >= g.active_steps) return;
  if (i_override > 0) {   // evaluation: every group reads rows [arow0, arow0 + i_override) of a shared A
    g.I = i_override;
    g.A = reinterpret_cast<const uint16_t*>(g.A) + (size_t)arow0 * g.lda;
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i0 = tile.ti * BM, j0 = tile.tj * BN;
  if (i0 >= g.I || j0 >= g.J) return;
  const uint16_t* A = reinterpret_cast<const uint16_t*>(g.A);
  const uint16_t* B = reinterpret_cast<const uint16_t*>(g.B);

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool want_db = (MODE == 2) && (flags & MT_GEMM_FLAG_DBIAS) && tile.tj == 0 && g.aux_out != nullptr;
  float db_acc = 0.f;

  u32x4 ra[4], rb[4];
  const int nk = (g.KK + BK - 1) / BK;
  load_tile<A_KMAJ>(ra, A, g.lda, i0, g.I, 0, g.KK, tid);
  load_tile<B_KMAJ>(rb, B, g.ldb, j0, g.J, 0, g.KK, tid);
  store_tile<A_KMAJ>(ra, ldsA, tid);
  store_tile<B_KMAJ>(rb, ldsB, tid);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) {   // next tile's loads fly while this tile is multiplied
      load_tile<A_KMAJ>(ra, A, g.lda, i0, g.I, (kt + 1) * BK, g.KK, tid);
      load_tile<B_KMAJ>(rb, B, g.ldb, j0, g.J, (kt + 1) * BK, g.KK, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if constexpr (A_KMAJ) af[m] = frag_kmajor(ldsA, wr * 64 + m * 16, ks, lane);
        else af[m] = frag_mnmajor(ldsA, wr * 64 + m * 16, ks, lane);
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if constexpr (B_KMAJ) bfr[n] = frag_kmajor(ldsB, wc * 64 + n * 16, ks, lane);
        else bfr[n] = frag_mnmajor(ldsB, wc * 64 + n * 16, ks, lane);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
    }
    if (want_db) {   // column sums of the A image: thread -> column (tid & 127), half of the 64 rows
      const int col = tid & 127, half = tid >> 7;
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) {
        const int row = half * 32 + r;
        const uint16_t v = *reinterpret_cast<const uint16_t*>(ldsA + mnmaj_off(row, col >> 3) + (col & 7) * 2);
        s += bf16_bits_to_f32(v);
      }
      db_acc += s;
    }
    __syncthreads();
    if (more) {
      store_tile<A_KMAJ>(ra, ldsA, tid);
      store_tile<B_KMAJ>(rb, ldsB, tid);
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------------------ epilogues
  // accumulator map: acc[m][n][j] = C[wr*64 + m*16 + (lane>>4)*4 + j][wc*64 + n*16 + (lane&15)]
  const int fq = lane >> 4, fr = lane & 15;
  if (MODE == 2) {
    float* C = reinterpret_cast<float*>(g.C);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int col = j0 + wc * 64 + n * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = i0 + wr * 64 + m * 16 + fq * 4 + j;
          if (row < g.I && col < g.J) C[(size_t)row * g.ldc + col] = acc[m][n][j];
        }
      }
    if (want_db) {
      float* red = reinterpret_cast<float*>(lds);    // all waves are past the last tile barrier
      red[tid] = db_acc;
      __syncthreads();
      if (tid < 128 && i0 + tid < g.I)
        reinterpret_cast<float*>(g.aux_out)[i0 + tid] = red[tid] + red[tid + 128];
    }
    return;
  }

  const float* bias = (MODE == 0) ? reinterpret_cast<const float*>(g.aux) : nullptr;
  const bool relu = (MODE == 0) && (flags & MT_GEMM_FLAG_RELU);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int lcol = wc * 64 + n * 16 + fr;
      float bv = 0.f;
      if (bias != nullptr && j0 + lcol < g.J) bv = bias[j0 + lcol];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lrow = wr * 64 + m * 16 + fq * 4 + j;
        float v = acc[m][n][j] + bv;
        if (relu) v = fmaxf(v, 0.f);
        *reinterpret_cast<uint16_t*>(lds + ctile_off(lrow, lcol >> 3) + (lcol & 7) * 2) =
            (uint16_t)f32_to_bf16_bits(v);
      }
    }
  __syncthreads();
  uint16_t* C = reinterpret_cast<uint16_t*>(g.C);
  const uint16_t* msk = (MODE == 1 && (flags & MT_GEMM_FLAG_MASK)) ? reinterpret_cast<const uint16_t*>(g.aux)
                                                                  : nullptr;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int q = tid + NTHREADS * t;
    const int lrow = q >> 4, ch = q & 15;
    const int row = i0 + lrow, col = j0 + ch * 8;
    if (row < g.I && col < g.J) {
      u32x4 v = *reinterpret_cast<const u32x4*>(lds + ctile_off(lrow, ch));
      if (msk != nullptr) {
        const u32x4 h = *reinterpret_cast<const u32x4*>(msk + (size_t)row * g.ldaux + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // bf16 > 0  <=>  sign bit clear and magnitude non-zero (NaN never appears post-ReLU)
          const uint32_t hw = h[e];
          const uint32_t lo_ok = ((hw & 0x8000u) == 0u && (hw & 0x7FFFu) != 0u) ? 0x0000FFFFu : 0u;
          const uint32_t hi_ok = ((hw & 0x80000000u) == 0u && (hw & 0x7FFF0000u) != 0u) ? 0xFFFF0000u : 0u;
          v[e] &= (lo_ok | hi_ok);
        }
      }
      *reinterpret_cast<u32x4*>(C + (size_t)row * g.ldc + col) = v;
    }
  }
}

extern "C" int mt_gemm_launch(int mode, const MtGemmGroup* groups, const MtTile* tiles, int ntiles, int step,
                              int flags, int arow0, int i_override, hipStream_t stream) {
  if (ntiles <= 0) return 0;
  dim3 grid(ntiles), block(NTHREADS);
  switch (mode) {
    case 0: hipLaunchKernelGGL(mt_gemm_kernel<0>, grid, block, 0, stream, groups, tiles, step, flags, arow0, i_override); break;
    case 1: hipLaunchKernelGGL(mt_gemm_kernel<1>, grid, block, 0, stream, groups, tiles, step, flags, arow0, i_override); break;
    case 2: hipLaunchKernelGGL(mt_gemm_kernel<2>, grid, block, 0, stream, groups, tiles, step, flags, arow0, i_override); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
