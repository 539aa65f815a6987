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
#include "mt_gemm_f32_tile.h"

template <int MODE, int NSTAGE, int T, int KU = 1, int PROD = 0>
__global__ __launch_bounds__(F32_THREADS) void mt_gemm_f32_kernel(const MtGemmGroup* __restrict__ groups,
                                                                  const MtTile* __restrict__ tiles, int step, int flags,
                                                                  int arow0, int i_override, int ksplits,
                                                                  size_t split_stride, const int* __restrict__ step_base) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[NSTAGE * 2 * T * 128];
  f32_gemm_tile<MODE, NSTAGE, T, KU, PROD>(groups, tiles[blockIdx.x], mt_step(step, step_base), flags, arow0, i_override,
                                           ksplits, split_stride, lds);
}

// Backward of one layer in ONE launch (weight-gradient tiles, input-gradient tiles, and optionally the head
// kernel's partial-record reduction as extra workgroups), as in gemm.hip.
template <int NSTAGE, int T, int KU = 1>
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
    f32_gemm_tile<2, NSTAGE, T, KU>(wgroups, wtiles[b], step, MT_GEMM_FLAG_DBIAS, 0, 0, ksplits, split_stride, lds);
  else if (b < n_wtiles + n_dtiles)
    f32_gemm_tile<1, NSTAGE, T, KU>(dgroups, dtiles[b - n_wtiles], step, MT_GEMM_FLAG_MASK, 0, 0, 1, 0, lds);
  else
    mt_head_reduce_body(hr.slots, hr.ids, step, hr.scratch, hr.max_chunks, hr.rec, b - n_wtiles - n_dtiles);
}

// tile_edge selects the tile: 128, 64 or 32, all on the two-stage, two-barrier pipeline (a 64-tile workgroup holds 32 KB
// of LDS, so five fit a CU).  Deeper pipelines were built and measured for launches of a few tiles per CU and dropped:
// 64-tiles with three stages and one barrier were 7-22 % slower; 64-tiles with four stages, 32-tiles with eight, and the
// latter with four K-steps per barrier changed nothing that matters on 13-slot launches (weight gradient of 16 tiles over
// 1024 rows: 19.4 us with two stages, 20.0 with eight, 17.4 with four K-steps per barrier) -- such a tile is bound by the
// dependent chain inside a K-step (LDS fragment reads -> MFMAs on one accumulator), not by memory latency or barriers.
// The tile list must have been built with the same edge.
#define MT_F32_FORMS(CALL)                    \
  switch (tile_edge) {                        \
    case 128: { CALL(2, 128); } break;        \
    case 64: { CALL(2, 64); } break;          \
    default: { CALL(2, 32); } break;          \
  }

static bool f32_edge_ok(int e) { return e == 128 || e == 64 || e == 32; }

extern "C" int mt_gemm_f32_launch(int mode, const MtGemmGroup* groups, const MtTile* tiles, int ntiles, int step,
                                  int flags, int arow0, int i_override, int ksplits, size_t split_stride, int tile_edge,
                                  hipStream_t stream) {
  if (ntiles <= 0) return 0;
  if (ksplits < 1 || ksplits > 127 || (mode != 2 && ksplits != 1)) return -5;
  if (!f32_edge_ok(tile_edge)) return -7;
  if (mode < 0 || mode > 2) return -1;
  dim3 grid(ntiles), block(F32_THREADS);
#define MT_F32_GO(...)                                                                                                    \
  do {                                                                                                                    \
    if (mode == 0)                                                                                                        \
      hipLaunchKernelGGL((mt_gemm_f32_kernel<0, __VA_ARGS__>), grid, block, 0, stream, groups, tiles, step, flags, arow0,       \
                         i_override, 1, (size_t)0, mt_step_base_tls);                                                     \
    else if (mode == 1)                                                                                                   \
      hipLaunchKernelGGL((mt_gemm_f32_kernel<1, __VA_ARGS__>), grid, block, 0, stream, groups, tiles, step, flags, arow0,       \
                         i_override, 1, (size_t)0, mt_step_base_tls);                                                     \
    else                                                                                                                  \
      hipLaunchKernelGGL((mt_gemm_f32_kernel<2, __VA_ARGS__>), grid, block, 0, stream, groups, tiles, step, flags, arow0,       \
                         i_override, ksplits, split_stride, mt_step_base_tls);                                            \
  } while (0)
  if (flags & MT_GEMM_FLAG_BF16_PRODUCTS) {       // f32 operands, bf16 products (packs that train at autocast precision)
    switch (tile_edge) {
      case 128: { MT_F32_GO(2, 128, 1, 1); } break;
      case 64: { MT_F32_GO(2, 64, 1, 1); } break;
      default: { MT_F32_GO(2, 32, 1, 1); } break;
    }
  } else {
    MT_F32_FORMS(MT_F32_GO)
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
  if (!f32_edge_ok(tile_edge)) return -7;
  dim3 grid(n_wtiles + n_dtiles + hr.n_slots), block(F32_THREADS);
#define MT_F32_BWD(...)                                                                                                 \
  hipLaunchKernelGGL((mt_gemm_f32_bwd_kernel<__VA_ARGS__>), grid, block, 0, stream, wgroups, wtiles, n_wtiles, dgroups, dtiles, \
                     n_dtiles, step, ksplits, split_stride, mt_step_base_tls, hr)
  MT_F32_FORMS(MT_F32_BWD)
#undef MT_F32_BWD
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
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.ldaux = mode == 1 ? ldc : 0;      // input gradient: aux is the layer input the ReLU mask is read from, shaped like C
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
