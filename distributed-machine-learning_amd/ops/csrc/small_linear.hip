// Backward of a small f32 dense layer in ONE launch: dX = dY W, dW = dY^T X, db = colsum(dY).
//
// A Transformer trial of the sequence-regression searches has R = batch x tokens ~ 10^3 rows against 10..256
// features: each of the three products is a few workgroups of a BLAS kernel plus a separate column reduction, and a
// training step spends its time on launch latency, not on arithmetic.  Here a workgroup of sixteen waves drives the
// exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 straight from global memory (operands of a whole round are
// requested before the first instruction, so a round costs one memory latency) and adds the waves' accumulators
// through LDS.  No workspace, no counters, no inter-workgroup traffic: every output element is written once by the
// workgroup that owns it, in a fixed summation order (bitwise repeatable).
//
//   * dW tile (o0, i0) [32x32], contraction over rows: the sixteen waves take R/16 consecutive rows each; lane l
//     feeds A = dY[r + l/32][o0 + l%32], B = X[r + l/32][i0 + l%32] -- both operands are 128-byte coalesced row
//     segments, no transposition needed.  Column tile 0 also adds up its dY operand: the bias gradient.
//   * dX block (r0, i0) [128x32], contraction over `out`: four row sub-tiles x four contraction quarters; lane l reads
//     four consecutive dY[row][k..k+3] (one 16-byte load feeds four instructions) against coalesced rows of W.
//
// Numerics: f32 fma chains, like a plain f32 GEMM; no reduced-precision operand.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SL_WAVES 16
#define SL_THREADS 1024

struct MtSmallLinBwd {
  const float* X;    // [R, in]
  const float* W;    // [out, in]
  const float* dY;   // [R, out]
  float* dX;         // [R, in] or null
  float* dW;         // [out, in]
  float* db;         // [out] or null
  int R, in, out;
  int n_dx;          // workgroups that compute dX blocks (0 when dX is null)
  int nti;           // 32-column tiles along `in`
  int rpw;           // rows per wave of a dW tile (even); SL_WAVES * rpw >= R
  int kq;            // contraction columns per wave of a dX sub-tile (multiple of 8 / 16); 4 * kq >= out
  int t16;           // 1: 16x16 tiles on v_mfma_f32_16x16x4_f32 (few weight tiles: more, smaller workgroups' worth of work)
};

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// D layout of the 32x32 forms: register i of lane l holds row (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), column l & 31.
__device__ __forceinline__ int d_row(int i, int l) { return (i & 3) + 8 * (i >> 2) + 4 * (l >> 5); }

__device__ void dx_block(const MtSmallLinBwd& p, int blk, float (*red)[1024]) {
  const int t = threadIdx.x, wave = t >> 6, l = t & 63, n = l & 31, kh = l >> 5;
  const int sub = wave >> 2, quarter = wave & 3;
  const int r0 = (blk / p.nti) * 128 + sub * 32, i0 = (blk % p.nti) * 32;
  const int row = r0 + n, col = i0 + n;
  const bool row_ok = row < p.R, col_ok = col < p.in;
  const float* a_row = p.dY + (size_t)(row_ok ? row : 0) * p.out;
  const float* b_col = p.W + (col_ok ? col : 0);
  const bool vec = (p.out & 3) == 0;
  const int k_begin = quarter * p.kq, k_end = min(p.out, k_begin + p.kq);
  f32x16 acc = {0};
  if (r0 < p.R) {
    for (int k0 = k_begin; k0 < k_end; k0 += 64) {     // a round: 8 x (one 16-byte A load + four B loads), then <= 32 MFMAs
      float a[8][4], b[8][4];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int kb = k0 + it * 8 + kh * 4;
        if (row_ok && vec && kb + 3 < k_end) {
          const float4 v = *reinterpret_cast<const float4*>(a_row + kb);
          a[it][0] = v.x; a[it][1] = v.y; a[it][2] = v.z; a[it][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[it][j] = (row_ok && kb + j < k_end) ? a_row[kb + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) b[it][j] = (col_ok && kb + j < k_end) ? b_col[(size_t)(kb + j) * p.in] : 0.f;
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        if (k0 + it * 8 < k_end) {                     // wave-uniform: skip instructions that would only add zeros
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = mfma2(a[it][j], b[it][j], acc);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][i * 64 + l] = acc[i];
  __syncthreads();
  const int g = t >> 8, tt = t & 255;                  // 256 threads finish each 32-row sub-tile
  const int rb = (blk / p.nti) * 128 + g * 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tt + 256 * j, i = e >> 6, ll = e & 63;
    const float v = (red[g * 4][e] + red[g * 4 + 1][e]) + (red[g * 4 + 2][e] + red[g * 4 + 3][e]);
    const int r = rb + d_row(i, ll), c = i0 + (ll & 31);
    if (r < p.R && c < p.in) p.dX[(size_t)r * p.in + c] = v;
  }
}

__device__ void dw_tile(const MtSmallLinBwd& p, int tile, float (*red)[1024], float (*redb)[32]) {
  const int t = threadIdx.x, wave = t >> 6, l = t & 63, n = l & 31, kh = l >> 5;
  const int to = tile / p.nti, ti = tile % p.nti;
  const int o0 = to * 32, i0 = ti * 32;
  const bool o_ok = o0 + n < p.out, i_ok = i0 + n < p.in;
  const int w_begin = wave * p.rpw;
  const int w_end = min(p.R, w_begin + p.rpw);
  const float* ap = p.dY + (o_ok ? o0 + n : 0);
  const float* bp = p.X + (i_ok ? i0 + n : 0);
  const bool want_db = p.db != nullptr && ti == 0;
  f32x16 acc = {0};
  float bsum = 0.f;
  for (int r = w_begin; r < w_end; r += 64) {          // a round: 64 rows requested, then <= 32 MFMAs
    float a[32], b[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int rr = r + 2 * j + kh;
      const bool ok = rr < w_end;
      a[j] = (ok && o_ok) ? ap[(size_t)rr * p.out] : 0.f;
      b[j] = (ok && i_ok) ? bp[(size_t)rr * p.in] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (r + 2 * j < w_end) {                         // wave-uniform
        acc = mfma2(a[j], b[j], acc);
        bsum += a[j];
      }
    }
  }
  bsum += __shfl_xor(bsum, 32);                        // both row parities of a column
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][i * 64 + l] = acc[i];
  if (kh == 0) redb[wave][n] = bsum;
  __syncthreads();
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < SL_WAVES; ++w) v += red[w][t];   // wave order: fixed
  const int i = t >> 6, ll = t & 63;
  const int o = o0 + d_row(i, ll), c = i0 + (ll & 31);
  if (o < p.out && c < p.in) p.dW[(size_t)o * p.in + c] = v;
  if (want_db && t < 32 && o0 + t < p.out) {
    float bs = 0.f;
#pragma unroll
    for (int w = 0; w < SL_WAVES; ++w) bs += redb[w][t];
    p.db[o0 + t] = bs;
  }
}

// ---- 16x16 tiles (v_mfma_f32_16x16x4_f32): A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15]; register i of lane l
// holds row 4 * (l >> 4) + i, column l & 15.  A 32x32 tile keeps all sixteen waves of its contraction on ONE CU, and
// the f32 matrix rate of a CU (256 flop/clk) then bounds the launch when a layer has only a handful of tiles; quarter
// tiles put four times as many CUs to work and need one operand round per wave instead of two at R = 1600.
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4v mfma4(float a, float b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ void dx_block16(const MtSmallLinBwd& p, int blk, float (*red)[256]) {     // 64 rows x 16 columns
  const int t = threadIdx.x, wave = t >> 6, l = t & 63, n = l & 15, kg = l >> 4;
  const int sub = wave >> 2, quarter = wave & 3;
  const int r0 = (blk / p.nti) * 64 + sub * 16, i0 = (blk % p.nti) * 16;
  const int row = r0 + n, col = i0 + n;
  const bool row_ok = row < p.R, col_ok = col < p.in;
  const float* a_row = p.dY + (size_t)(row_ok ? row : 0) * p.out;
  const float* b_col = p.W + (col_ok ? col : 0);
  const bool vec = (p.out & 3) == 0;
  const int k_begin = quarter * p.kq, k_end = min(p.out, k_begin + p.kq);
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  if (r0 < p.R) {
    for (int k0 = k_begin; k0 < k_end; k0 += 128) {    // a round: 8 x (one 16-byte A load + four B loads), then <= 32 MFMAs
      float a[8][4], b[8][4];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int kb = k0 + it * 16 + kg * 4;
        if (row_ok && vec && kb + 3 < k_end) {
          const float4 v = *reinterpret_cast<const float4*>(a_row + kb);
          a[it][0] = v.x; a[it][1] = v.y; a[it][2] = v.z; a[it][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[it][j] = (row_ok && kb + j < k_end) ? a_row[kb + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) b[it][j] = (col_ok && kb + j < k_end) ? b_col[(size_t)(kb + j) * p.in] : 0.f;
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        if (k0 + it * 16 < k_end) {                    // wave-uniform
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = mfma4(a[it][j], b[it][j], acc);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][i * 64 + l] = acc[i];
  __syncthreads();
  const int g = t >> 8, e = t & 255;                   // 256 threads finish each 16-row sub-tile, one element each
  const int i = e >> 6, ll = e & 63;
  const float v = (red[g * 4][e] + red[g * 4 + 1][e]) + (red[g * 4 + 2][e] + red[g * 4 + 3][e]);
  const int r = (blk / p.nti) * 64 + g * 16 + 4 * (ll >> 4) + i, c = i0 + (ll & 15);
  if (r < p.R && c < p.in) p.dX[(size_t)r * p.in + c] = v;
}

__device__ void dw_tile16(const MtSmallLinBwd& p, int tile, float (*red)[256], float (*redb)[16]) {
  const int t = threadIdx.x, wave = t >> 6, l = t & 63, n = l & 15, kg = l >> 4;
  const int to = tile / p.nti, ti = tile % p.nti;
  const int o0 = to * 16, i0 = ti * 16;
  const bool o_ok = o0 + n < p.out, i_ok = i0 + n < p.in;
  const int w_begin = wave * p.rpw;
  const int w_end = min(p.R, w_begin + p.rpw);
  const float* ap = p.dY + (o_ok ? o0 + n : 0);
  const float* bp = p.X + (i_ok ? i0 + n : 0);
  const bool want_db = p.db != nullptr && ti == 0;
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int r = w_begin; r < w_end; r += 128) {         // a round: 128 rows requested, then <= 32 MFMAs
    float a[32], b[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int rr = r + 4 * j + kg;
      const bool ok = rr < w_end;
      a[j] = (ok && o_ok) ? ap[(size_t)rr * p.out] : 0.f;
      b[j] = (ok && i_ok) ? bp[(size_t)rr * p.in] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (r + 4 * j < w_end) {                         // wave-uniform
        acc = mfma4(a[j], b[j], acc);
        bsum += a[j];
      }
    }
  }
  bsum += __shfl_xor(bsum, 16);                        // the four row residues of a column
  bsum += __shfl_xor(bsum, 32);
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][i * 64 + l] = acc[i];
  if (kg == 0) redb[wave][n] = bsum;
  __syncthreads();
  if (t < 256) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SL_WAVES; ++w) v += red[w][t];  // wave order: fixed
    const int i = t >> 6, ll = t & 63;
    const int o = o0 + 4 * (ll >> 4) + i, c = i0 + (ll & 15);
    if (o < p.out && c < p.in) p.dW[(size_t)o * p.in + c] = v;
  }
  if (want_db && t < 16 && o0 + t < p.out) {
    float bs = 0.f;
#pragma unroll
    for (int w = 0; w < SL_WAVES; ++w) bs += redb[w][t];
    p.db[o0 + t] = bs;
  }
}

__global__ __launch_bounds__(SL_THREADS) void mt_small_linear_bwd_kernel(const MtSmallLinBwd p) {
  __shared__ float red[SL_WAVES][1024];
  __shared__ float redb[SL_WAVES][32];
  const int b = blockIdx.x;
  if (b < p.n_dx) dx_block(p, b, red);
  else dw_tile(p, b - p.n_dx, red, redb);
}

__global__ __launch_bounds__(SL_THREADS) void mt_small_linear_bwd16_kernel(const MtSmallLinBwd p) {
  __shared__ float red[SL_WAVES][256];
  __shared__ float redb[SL_WAVES][16];
  const int b = blockIdx.x;
  if (b < p.n_dx) dx_block16(p, b, red);
  else dw_tile16(p, b - p.n_dx, red, redb);
}

// tile: 0 = choose (quarter tiles while the layer has at most SL_T16_MAX_TILES 32x32 tiles or at most 128 input
// features -- measured at R = 1600: 64 -> 256 6.9 vs 15.8 us, 128 -> 512 15.8 vs 19.4 us, but 512 -> 128 24.3 vs 21.0 us
// and 256 -> 1024 55.6 vs 40.2 us, where the quarter tiles re-read too much of a wide X); 16 or 32 = force.
#define SL_T16_MAX_TILES 32
extern "C" int mt_small_linear_bwd_tiled_launch(const float* X, const float* W, const float* dY, float* dX, float* dW,
                                                float* db, int R, int in, int out, int tile, hipStream_t stream);

extern "C" int mt_small_linear_bwd_launch(const float* X, const float* W, const float* dY, float* dX, float* dW, float* db,
                                          int R, int in, int out, hipStream_t stream) {
  return mt_small_linear_bwd_tiled_launch(X, W, dY, dX, dW, db, R, in, out, 0, stream);
}

extern "C" int mt_small_linear_bwd_tiled_launch(const float* X, const float* W, const float* dY, float* dX, float* dW,
                                                float* db, int R, int in, int out, int tile, hipStream_t stream) {
  if (R <= 0 || in <= 0 || out <= 0) return -1;
  if (!X || !W || !dY || !dW) return -2;
  if (((uintptr_t)dY & 15) != 0) return -3;            // 16-byte loads of dY rows
  MtSmallLinBwd p;
  p.X = X; p.W = W; p.dY = dY; p.dX = dX; p.dW = dW; p.db = db;
  p.R = R; p.in = in; p.out = out;
  if (tile != 0 && tile != 16 && tile != 32) return -7;
  if (tile == 0) tile = (in <= 128 || ((in + 31) / 32) * ((out + 31) / 32) <= SL_T16_MAX_TILES) ? 16 : 32;
  p.t16 = tile == 16;
  const int ts = p.t16 ? 16 : 32, rows_per_dx_block = p.t16 ? 64 : 128, kgran = p.t16 ? 4 : 2;
  p.nti = (in + ts - 1) / ts;
  const int nto = (out + ts - 1) / ts;
  p.n_dx = dX ? ((R + rows_per_dx_block - 1) / rows_per_dx_block) * p.nti : 0;
  p.rpw = ((R + SL_WAVES - 1) / SL_WAVES + kgran - 1) / kgran * kgran;
  p.kq = p.t16 ? ((out + 3) / 4 + 15) / 16 * 16 : ((out + 3) / 4 + 7) / 8 * 8;
  if ((long)SL_WAVES * p.rpw < R || 4L * p.kq < out) return -5;
  const long blocks = (long)p.n_dx + (long)nto * p.nti;
  if (blocks <= 0 || blocks > 0x7fffffffL) return -6;
  if (p.t16)
    hipLaunchKernelGGL(mt_small_linear_bwd16_kernel, dim3((unsigned)blocks), dim3(SL_THREADS), 0, stream, p);
  else
    hipLaunchKernelGGL(mt_small_linear_bwd_kernel, dim3((unsigned)blocks), dim3(SL_THREADS), 0, stream, p);
  return (int)hipGetLastError();
}
