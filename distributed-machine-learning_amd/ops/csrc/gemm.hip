// Grouped bf16 GEMM on MFMA for packed multi-trial MLP training (gfx950).
//
// One launch serves every trial slot of a pack: block b takes tiles[b] = (group, ti, tj) and
// computes one 128x128 output tile of that group's C = A x B with BK = 64.  Four waves (2 x 2),
// each owning a 64x64 sub-tile as 4 x 4 accumulators of v_mfma_f32_16x16x32_bf16.
//
// Operands come in two storage forms and each has its own LDS image and fragment read:
//   K-major  ([row][kk], kk contiguous; X in forward, W in forward, dY in dgrad):
//       image [128 rows][64 kk] bf16, 128-B rows, 16-B chunk index XORed with (row & 7);
//       fragment = one ds_read_b128 per 16x32 block.
//   MN-major ([kk][col], col contiguous; W in dgrad, dY and X in wgrad):
//       image [64 kk][128 cols] bf16, 256-B rows, chunk index XORed with ((row&3)<<2 | (row>>2)&3);
//       fragment = two ds_read_b64_tr_b16 (hardware transpose: each lane receives 4 consecutive
//       kk of ONE column), so no operand ever needs a transposed copy in HBM.
//
// Pipeline: LDS stages of 32 KB (two stages and two workgroups per CU for big packs, four stages and one
// workgroup per CU when a pack has fewer tiles than the chip has CUs) filled by LDS-DMA (global_load_lds_dwordx4):
// no staging registers and no ds_write pass -- with K <= 512 these GEMMs are bound by how many bytes a
// CU keeps in flight, and register staging spent a third of every K-step moving 32 KB through VGPRs
// into LDS.  An LDS-DMA instruction writes wave-uniform-base + lane*16, so the images are filled in
// 1-KiB pieces and the row swizzle is applied to the per-lane SOURCE address; out-of-range chunks read
// a 16-byte block of zeros instead of being skipped.  Tile k+2 is issued into the stage tile k just
// vacated (after the barrier that ends its MFMAs); a counted s_waitcnt vmcnt(8) (= the 8 DMA pieces a
// wave issues per tile) retires tile k+1 while tile k+2 stays in flight across the barrier.
//
// The MFMAs are issued with the operand roles swapped (B fragment first): the accumulator then holds,
// per lane, four CONSECUTIVE output columns of one output row, so the epilogues move 8 bytes (bf16)
// or 16 bytes (f32) per lane instead of single elements.
//
// Epilogues: forward adds bias and applies ReLU in f32, converts to bf16, stages the tile through LDS
// (8-byte writes) and stores whole 256-B rows, 16 bytes per lane -- measured 7-20 % faster end to end than
// 8-byte stores straight from the accumulators; dgrad does the same and zeroes elements whose forward
// activation was <= 0 (ReLU backward) while storing; wgrad stores f32 (16 bytes per lane) and, on the
// tj == 0 tiles, also reduces the bias gradient (column sums of dY) from the LDS image it already holds.
// wgrad can split its contraction (the batch) over several workgroups: split s writes its partial sums to
// slab s of the gradient buffer and the optimiser adds the slabs in order (deterministic, no atomics);
// that trades bytes for latency when a pack has too few tiles to fill 256 CUs.
//
// One large GEMM (hundreds of 256 x 256 tiles; the layers of a data-parallel trial) takes the eight-wave 256-tile kernel
// in the second half of this file: same images and fragment reads, half the staged bytes per flop, fragment reads issued a
// group of MFMAs ahead, and a banded tile order that keeps an XCD's resident tiles on few operand panels.
#include "mt_bodies.h"

#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256
#define STAGE_BYTES 32768

typedef __attribute__((address_space(3))) uint8_t lds_u8;

__device__ __forceinline__ int kmaj_off(int row, int chunk) {       // [128][64] bf16 image
  return row * 128 + ((chunk ^ (row & 7)) << 4);
}
__device__ __forceinline__ int mnmaj_off(int row, int chunk) {      // [64][128] bf16 image
  return row * 256 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}
__device__ __forceinline__ int ctile_off(int row, int chunk) {      // [128][128] bf16 epilogue image
  return row * 256 + ((chunk ^ (row & 15)) << 4);
}

// ---- global -> LDS by LDS-DMA (4 x 1 KiB pieces per wave per operand image) ------------------------
__device__ __attribute__((aligned(16))) uint32_t mt_zero_chunk[4];     // source of every out-of-range chunk

template <bool KMAJOR, int PIECES = 4>      // PIECES per wave: 4 with four waves, 2 with eight
__device__ __forceinline__ void stage_tile(uint8_t* img, const uint16_t* __restrict__ base, int ld, int mn0, int mn_lim,
                                           int k0, int k_lim, int wave, int lane) {
#pragma unroll
  for (int t = 0; t < PIECES; ++t) {
    const int j = wave * PIECES + t;   // which 1-KiB piece of the 16-KiB image this instruction fills
    int row, col;
    bool ok;
    if (KMAJOR) {      // image row r = 128 B = 8 chunks; slot s of row r holds chunk s ^ (r & 7)
      const int r = j * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (r & 7);
      row = mn0 + r;
      col = k0 + c * 8;
      ok = (row < mn_lim) && (col < k_lim);
    } else {           // image row r = 256 B = 16 chunks; slot s holds chunk s ^ ((r&3)<<2 | (r>>2)&3)
      const int r = j * 4 + (lane >> 4);
      const int c = (lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
      row = k0 + r;
      col = mn0 + c * 8;
      ok = (row < k_lim) && (col < mn_lim);
    }
    const uint16_t* src = ok ? base + (size_t)row * ld + col : reinterpret_cast<const uint16_t*>(mt_zero_chunk);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(img + j * 1024), 16, 0, 0);
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

// All MFMAs of one 64-deep stage for this wave's 64x64 sub-tile (operands swapped, see header).
template <bool A_KMAJ, bool B_KMAJ>
__device__ __forceinline__ void mma_stage(f32x4 (&acc)[4][4], const uint8_t* ldsA, const uint8_t* ldsB, int wr, int wc,
                                          int lane, int flags) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 a_[4], b_[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if constexpr (A_KMAJ) a_[m] = frag_kmajor(ldsA, wr * 64 + m * 16, ks, lane);
      else a_[m] = frag_mnmajor(ldsA, wr * 64 + m * 16, ks, lane);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      if constexpr (B_KMAJ) b_[n] = frag_kmajor(ldsB, wc * 64 + n * 16, ks, lane);
      else b_[n] = frag_mnmajor(ldsB, wc * 64 + n * 16, ks, lane);
    }
    if (!(flags & 256)) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_[n], a_[m], acc[m][n], 0, 0, 0);
    } else {       // timing-only ablation: fragments stay live, matrix pipe idle
#pragma unroll
      for (int m = 0; m < 4; ++m) asm volatile("" ::"v"(a_[m]), "v"(b_[m]));
    }
  }
}

// Column sums of the MN-major A image (bias gradient): thread -> column (tid & 127), half of the 64 rows.
__device__ __forceinline__ float colsum_stage(const uint8_t* ldsA, int tid) {
  const int col = tid & 127, half = tid >> 7;
  float s = 0.f;
#pragma unroll 8
  for (int r = 0; r < 32; ++r) {
    const int row = half * 32 + r;
    const uint16_t v = *reinterpret_cast<const uint16_t*>(ldsA + mnmaj_off(row, col >> 3) + (col & 7) * 2);
    s += bf16_bits_to_f32(v);
  }
  return s;
}

// One 128x128 output tile.  ksplits > 1 (wgrad only): tile.ti (int16) carries the split index in bits 8..14,
// so at most 127 splits.
template <int MODE, int NSTAGE>
__device__ __forceinline__ void gemm_tile(const MtGemmGroup* __restrict__ groups, const MtTile tile, int step,
                                          int flags, int arow0, int i_override, int ksplits, size_t split_stride,
                                          uint8_t* lds) {
  constexpr bool A_KMAJ = (MODE != 2);
  constexpr bool B_KMAJ = (MODE == 0);
  if (tile.group < 0) return;   // padding entry of an XCD-balanced tile list
  MtGemmGroup g = groups[tile.group];
  if (step >= g.active_steps) return;
  if (i_override > 0) {   // evaluation: every group reads rows [arow0, arow0 + i_override) of a shared A
    g.I = i_override;
    g.A = reinterpret_cast<const uint16_t*>(g.A) + (size_t)arow0 * g.lda;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int split = (MODE == 2) ? (tile.ti >> 8) : 0;
  const int i0 = (MODE == 2 ? (tile.ti & 0xFF) : (int)tile.ti) * BM, j0 = tile.tj * BN;   // only wgrad tiles carry a split index
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

  const int nk_all = (g.KK + BK - 1) / BK;
  int kt_lo = 0, kt_hi = nk_all;
  if (MODE == 2 && ksplits > 1) {
    const int per = (nk_all + ksplits - 1) / ksplits;
    kt_lo = split * per;
    kt_hi = (kt_lo + per < nk_all) ? kt_lo + per : nk_all;
  }

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int n_it = kt_hi - kt_lo;
  auto issue = [&](int slot, int kt) {      // 8 LDS-DMA pieces per wave: tile kt -> stage `slot`
    stage_tile<A_KMAJ>(lds + slot * STAGE_BYTES, A, g.lda, i0, g.I, kt * BK, g.KK, wave_u, lane);
    stage_tile<B_KMAJ>(lds + slot * STAGE_BYTES + 16384, B, g.ldb, j0, g.J, kt * BK, g.KK, wave_u, lane);
  };

  if constexpr (NSTAGE == 2) {
    // Two stages, two workgroups per CU (throughput form).  Tile it+2 refills the stage tile `it` just
    // vacated, so two barriers per K-step; the other workgroup on the CU covers the bubbles.
    if (n_it > 0) issue(0, kt_lo);
    if (n_it > 1) issue(1, kt_lo + 1);
    for (int it = 0; it < n_it; ++it) {
      const int cur = it & 1;
      const uint8_t* ldsA = lds + cur * STAGE_BYTES;
      // this wave's 8 pieces of tile `it` have landed once at most the 8 pieces of tile it+1 are outstanding
      if (it + 1 < n_it) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // ... and after the barrier so have everybody else's
      mma_stage<A_KMAJ, B_KMAJ>(acc, ldsA, ldsA + 16384, wr, wc, lane, flags);
      if (want_db) db_acc += colsum_stage(ldsA, tid);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // every wave is done reading this stage: it may be refilled
      if (it + 2 < n_it && !(flags & 512)) issue(cur, kt_lo + it + 2);
    }
  } else {
    // NSTAGE stages, one workgroup per CU (latency form, for packs with fewer tiles than CUs): NSTAGE-1
    // tiles stay in flight and ONE barrier per K-step both publishes tile `it` and proves that everybody
    // has finished tile it-1, whose stage is refilled right away.
#pragma unroll
    for (int p = 0; p < NSTAGE - 1; ++p)
      if (p < n_it) issue(p, kt_lo + p);
    for (int it = 0; it < n_it; ++it) {
      const int ahead = (n_it - 1 - it) < (NSTAGE - 2) ? (n_it - 1 - it) : (NSTAGE - 2);
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int nxt = it + NSTAGE - 1;
      if (nxt < n_it && !(flags & 512)) issue(nxt % NSTAGE, kt_lo + nxt);
      const uint8_t* ldsA = lds + (it % NSTAGE) * STAGE_BYTES;
      mma_stage<A_KMAJ, B_KMAJ>(acc, ldsA, ldsA + 16384, wr, wc, lane, flags);
      if (want_db) db_acc += colsum_stage(ldsA, tid);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // the epilogues reuse stage 0
  }

  // ------------------------------------------------------------------------------ epilogues
  const int fq = lane >> 4, fr = lane & 15;
  if (flags & 1024) {       // timing-only: keep the accumulators alive, store nothing
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) asm volatile("" ::"v"(acc[m][n]));
    return;
  }
  if (MODE == 2) {
    float* C = reinterpret_cast<float*>(g.C) + (size_t)split * split_stride;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = i0 + wr * 64 + m * 16 + fr;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int col = j0 + wc * 64 + n * 16 + fq * 4;
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
      if (tid < 128 && i0 + tid < g.I)
        (reinterpret_cast<float*>(g.aux_out) + (size_t)split * split_stride)[i0 + tid] = red[tid] + red[tid + 128];
    }
    return;
  }

  {
    const bool relu = (MODE == 0) && (flags & MT_GEMM_FLAG_RELU);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int lcol = wc * 64 + n * 16 + fq * 4;
      float bb[4] = {0.f, 0.f, 0.f, 0.f};
      if (MODE == 0 && g.aux != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j0 + lcol + j < g.J) bb[j] = reinterpret_cast<const float*>(g.aux)[j0 + lcol + j];
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int lrow = wr * 64 + m * 16 + fr;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[m][n][j] + bb[j];
          if (relu) v[j] = fmaxf(v[j], 0.f);
        }
        u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(lds + ctile_off(lrow, lcol >> 3) + (lcol & 7) * 2) = pk;
      }
    }
    __syncthreads();
    uint16_t* C = reinterpret_cast<uint16_t*>(g.C);
    const uint16_t* msk = (MODE == 1 && (flags & MT_GEMM_FLAG_MASK)) ? reinterpret_cast<const uint16_t*>(g.aux)
                                                                    : nullptr;
    u32x4 hv[8];
    if (msk != nullptr) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int q = tid + NTHREADS * t;
        const int row = i0 + (q >> 4), col = j0 + (q & 15) * 8;
        const bool ok = (row < g.I && col < g.J);
        hv[t] = *reinterpret_cast<const u32x4*>(msk + (ok ? ((size_t)row * g.ldaux + col) : (size_t)0));
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int q = tid + NTHREADS * t;
      const int lrow = q >> 4, ch = q & 15;
      const int row = i0 + lrow, col = j0 + ch * 8;
      u32x4 v = *reinterpret_cast<const u32x4*>(lds + ctile_off(lrow, ch));
      if (msk != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t hw = hv[t][e];
          const uint32_t lo_ok = ((hw & 0x8000u) == 0u && (hw & 0x7FFFu) != 0u) ? 0x0000FFFFu : 0u;
          const uint32_t hi_ok = ((hw & 0x80000000u) == 0u && (hw & 0x7FFF0000u) != 0u) ? 0xFFFF0000u : 0u;
          v[e] &= (lo_ok | hi_ok);
        }
      }
      if (row < g.I && col < g.J) *reinterpret_cast<u32x4*>(C + (size_t)row * g.ldc + col) = v;
    }
  }
}

// ====================================================================================================
// 256 x 256 output tiles (experimental; selected only when a caller passes nstage = 256).
//
// Measured slower than the 128-tile forms with this simple two-barrier K-loop and one wave per SIMD (256 slots of
// 512^3: forward 166 us vs 122 us): a 256-tile only pays with ds_reads, LDS-DMA and MFMAs interleaved across two waves per
// SIMD -- the eight-wave kernel further down, which serves single large GEMMs.  This grouped four-wave form is kept for
// packs because it is numerically verified and selectable per launch.
//
// With K <= 512 a tile is a chain of at most 8 K-steps, each waiting on one global->LDS round trip, so a CU's
// rate is (output area per resident tile) / (round-trip time).  A 128x128 tile moves 32 KB per K-step for 2*128*128*64
// flop; a 256x256 tile moves 64 KB for four times that, i.e. half the bytes per flop through the same latency-bound
// pipe.  Layout: a stage holds four 16-KiB images [A rows 0-127 | A rows 128-255 | B cols 0-127 | B cols 128-255] in
// exactly the image formats of the 128-tile path (same staging and fragment code); wave (wr, wc) owns output
// quadrant (wr, wc) = 128x128 = 8 x 8 accumulators (256 accumulator registers per lane, one wave per SIMD), and
// two 64-KiB stages fill 128 KiB of the CU's 160 KiB LDS (one workgroup per CU).  Waves whose quadrant lies outside
// the group's matrix (edge tiles of narrow layers) skip their MFMAs but still stage and synchronise.
#define BIGT 256
#define BIG_STAGE_BYTES 65536

template <bool A_KMAJ, bool B_KMAJ>
__device__ __forceinline__ void mma_stage_big(f32x4 (&acc)[8][8], const uint8_t* imgA, const uint8_t* imgB, int lane) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 b_[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      if constexpr (B_KMAJ) b_[n] = frag_kmajor(imgB, n * 16, ks, lane);
      else b_[n] = frag_mnmajor(imgB, n * 16, ks, lane);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      bf16x8 a_;
      if constexpr (A_KMAJ) a_ = frag_kmajor(imgA, m * 16, ks, lane);
      else a_ = frag_mnmajor(imgA, m * 16, ks, lane);
#pragma unroll
      for (int n = 0; n < 8; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_[n], a_, acc[m][n], 0, 0, 0);
    }
  }
}

template <int MODE>
__device__ __forceinline__ void gemm_tile_big(const MtGemmGroup* __restrict__ groups, const MtTile tile, int step,
                                              int flags, int arow0, int i_override, int ksplits, size_t split_stride,
                                              uint8_t* lds) {
  constexpr bool A_KMAJ = (MODE != 2);
  constexpr bool B_KMAJ = (MODE == 0);
  if (tile.group < 0) return;
  MtGemmGroup g = groups[tile.group];
  if (step >= g.active_steps) return;
  if (i_override > 0) {
    g.I = i_override;
    g.A = reinterpret_cast<const uint16_t*>(g.A) + (size_t)arow0 * g.lda;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int split = (MODE == 2) ? (tile.ti >> 8) : 0;
  const int i0 = (tile.ti & 0xFF) * BIGT, j0 = tile.tj * BIGT;
  if (i0 >= g.I || j0 >= g.J) return;
  const uint16_t* A = reinterpret_cast<const uint16_t*>(g.A);
  const uint16_t* B = reinterpret_cast<const uint16_t*>(g.B);
  const bool live = (i0 + wr * 128 < g.I) && (j0 + wc * 128 < g.J);      // wave-uniform

  f32x4 acc[8][8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool want_db = (MODE == 2) && (flags & MT_GEMM_FLAG_DBIAS) && tile.tj == 0 && g.aux_out != nullptr;
  float db0 = 0.f, db1 = 0.f;

  const int nk_all = (g.KK + BK - 1) / BK;
  int kt_lo = 0, kt_hi = nk_all;
  if (MODE == 2 && ksplits > 1) {
    const int per = (nk_all + ksplits - 1) / ksplits;
    kt_lo = split * per;
    kt_hi = (kt_lo + per < nk_all) ? kt_lo + per : nk_all;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int n_it = kt_hi - kt_lo;
  auto issue = [&](int slot, int kt) {      // 16 LDS-DMA pieces per wave: tile kt -> the four images of stage `slot`
    uint8_t* st = lds + slot * BIG_STAGE_BYTES;
    // The 16 per-lane source addresses are recomputed for every K-tile (a few dozen integer ops next to 128 MFMAs):
    // hoisted out of the loop they would pin ~40 registers for its whole length and spill the accumulator file.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    stage_tile<A_KMAJ>(st, A, g.lda, i0, g.I, kt * BK, g.KK, wave_u, ln);
    stage_tile<A_KMAJ>(st + 16384, A, g.lda, i0 + 128, g.I, kt * BK, g.KK, wave_u, ln);
    stage_tile<B_KMAJ>(st + 32768, B, g.ldb, j0, g.J, kt * BK, g.KK, wave_u, ln);
    stage_tile<B_KMAJ>(st + 49152, B, g.ldb, j0 + 128, g.J, kt * BK, g.KK, wave_u, ln);
  };

  if (n_it > 0) issue(0, kt_lo);
  if (n_it > 1) issue(1, kt_lo + 1);
  for (int it = 0; it < n_it; ++it) {
    const int cur = it & 1;
    const uint8_t* st = lds + cur * BIG_STAGE_BYTES;
    if (it + 1 < n_it) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (live) mma_stage_big<A_KMAJ, B_KMAJ>(acc, st + wr * 16384, st + 32768 + wc * 16384, lane);
    if (want_db) { db0 += colsum_stage(st, tid); db1 += colsum_stage(st + 16384, tid); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 2 < n_it) issue(cur, kt_lo + it + 2);
  }

  // ------------------------------------------------------------------------------ epilogues
  const int fq = lane >> 4, fr = lane & 15;
  const int qi0 = i0 + wr * 128, qj0 = j0 + wc * 128;      // this wave's quadrant
  if (MODE == 2) {
    float* C = reinterpret_cast<float*>(g.C) + (size_t)split * split_stride;
    if (live) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int row = qi0 + m * 16 + fr;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          const int col = qj0 + n * 16 + fq * 4;
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
    }
    if (want_db) {
      float* red = reinterpret_cast<float*>(lds);    // all waves are past the last K-step barrier
      red[tid] = db0;
      red[256 + tid] = db1;
      __syncthreads();
      float* out = reinterpret_cast<float*>(g.aux_out) + (size_t)split * split_stride;
      if (tid < 128) {
        if (i0 + tid < g.I) out[i0 + tid] = red[tid] + red[tid + 128];
        if (i0 + 128 + tid < g.I) out[i0 + 128 + tid] = red[256 + tid] + red[256 + tid + 128];
      }
    }
    return;
  }
  if (!live) return;
  {
    uint8_t* img = lds + wave * 32768;       // this wave's own 128x128 bf16 staging image (wave-private: no barrier)
    const bool relu = (MODE == 0) && (flags & MT_GEMM_FLAG_RELU);
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int lcol = n * 16 + fq * 4;
      float bb[4] = {0.f, 0.f, 0.f, 0.f};
      if (MODE == 0 && g.aux != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (qj0 + lcol + j < g.J) bb[j] = reinterpret_cast<const float*>(g.aux)[qj0 + lcol + j];
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int lrow = m * 16 + fr;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[m][n][j] + bb[j];
          if (relu) v[j] = fmaxf(v[j], 0.f);
        }
        u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(img + ctile_off(lrow, lcol >> 3) + (lcol & 7) * 2) = pk;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    uint16_t* C = reinterpret_cast<uint16_t*>(g.C);
    const uint16_t* msk = (MODE == 1 && (flags & MT_GEMM_FLAG_MASK)) ? reinterpret_cast<const uint16_t*>(g.aux)
                                                                    : nullptr;
#pragma unroll 1
    for (int t0 = 0; t0 < 32; t0 += 8) {
      u32x4 hv[8];
      if (msk != nullptr) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int q = lane + 64 * (t0 + t);
          const int row = qi0 + (q >> 4), col = qj0 + (q & 15) * 8;
          const bool ok = (row < g.I && col < g.J);
          hv[t] = *reinterpret_cast<const u32x4*>(msk + (ok ? ((size_t)row * g.ldaux + col) : (size_t)0));
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int q = lane + 64 * (t0 + t);
        const int lrow = q >> 4, ch = q & 15;
        const int row = qi0 + lrow, col = qj0 + ch * 8;
        u32x4 v = *reinterpret_cast<const u32x4*>(img + ctile_off(lrow, ch));
        if (msk != nullptr) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t hw = hv[t][e];
            const uint32_t lo_ok = ((hw & 0x8000u) == 0u && (hw & 0x7FFFu) != 0u) ? 0x0000FFFFu : 0u;
            const uint32_t hi_ok = ((hw & 0x80000000u) == 0u && (hw & 0x7FFF0000u) != 0u) ? 0xFFFF0000u : 0u;
            v[e] &= (lo_ok | hi_ok);
          }
        }
        if (row < g.I && col < g.J) *reinterpret_cast<u32x4*>(C + (size_t)row * g.ldc + col) = v;
      }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(NTHREADS) void mt_gemm256_kernel(const MtGemmGroup* __restrict__ groups,
                                                              const MtTile* __restrict__ tiles, int step, int flags,
                                                              int arow0, int i_override, int ksplits,
                                                              size_t split_stride, const int* __restrict__ step_base) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * BIG_STAGE_BYTES];
  gemm_tile_big<MODE>(groups, tiles[blockIdx.x], mt_step(step, step_base), flags, arow0, i_override, ksplits,
                      split_stride, lds);
}

__global__ __launch_bounds__(NTHREADS) void mt_gemm256_bwd_kernel(const MtGemmGroup* __restrict__ wgroups,
                                                                  const MtTile* __restrict__ wtiles, int n_wtiles,
                                                                  const MtGemmGroup* __restrict__ dgroups,
                                                                  const MtTile* __restrict__ dtiles, int step_imm,
                                                                  int ksplits, size_t split_stride,
                                                                  const int* __restrict__ step_base) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * BIG_STAGE_BYTES];
  const int step = mt_step(step_imm, step_base);
  if ((int)blockIdx.x < n_wtiles)
    gemm_tile_big<2>(wgroups, wtiles[blockIdx.x], step, MT_GEMM_FLAG_DBIAS, 0, 0, ksplits, split_stride, lds);
  else
    gemm_tile_big<1>(dgroups, dtiles[blockIdx.x - n_wtiles], step, MT_GEMM_FLAG_MASK, 0, 0, 1, 0, lds);
}

// ====================================================================================================
// 256 x 256 output tiles on EIGHT waves, fragment reads software-pipelined under the MFMAs: the form for one large GEMM
// (layers thousands wide, a data-parallel trial).  A 128-tile stages 32 KB per 2*128*128*64 flop and, with its two
// barriers per K-step, tops out near a third of the matrix peak; this tile stages half the bytes per flop, and each
// SIMD holds two waves, so one wave's ds_reads and barrier waits sit under the other's MFMAs.
//
// Waves 2 (rows) x 4 (columns): wave (wr, wc) owns rows wr*128..+128, columns wc*64..+64 = 8 x 4 accumulators (128
// registers).  A stage is the same four 16-KiB images as above [A rows 0-127 | A rows 128-255 | B cols 0-127 | B cols
// 128-255], two stages = 128 KiB, one workgroup per CU.  One K-tile (64 deep) is four groups of 16 MFMAs -- (ks, row
// half) -- and the reads of group q+1 are issued before the MFMAs of group q:
//     read A(ks0, lo), B(ks0) | MFMA(ks1, hi) of tile k-1 | read A(ks0, hi) | MFMA(ks0, lo) | read A(ks1, lo), B(ks1) |
//     MFMA(ks0, hi) | read A(ks1, hi) | MFMA(ks1, lo) | barrier | LDS-DMA of tile k+2 into the stage just read
// so every batch of ds_reads has 16 MFMAs (512 matrix-pipe cycles) to come back, and a K-tile costs ONE barrier.
// An element's contraction is still summed in K order, so results equal the 128-tile kernels' bit for bit.
#define P8_THREADS 512

template <bool A_KMAJ>
__device__ __forceinline__ void p8_read_a(bf16x8 (&a)[4], const uint8_t* imgA, int mh, int ks, int lane) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if constexpr (A_KMAJ) a[m] = frag_kmajor(imgA, (mh * 4 + m) * 16, ks, lane);
    else a[m] = frag_mnmajor(imgA, (mh * 4 + m) * 16, ks, lane);
  }
}
template <bool B_KMAJ>
__device__ __forceinline__ void p8_read_b(bf16x8 (&b)[4], const uint8_t* imgB, int col0, int ks, int lane) {
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    if constexpr (B_KMAJ) b[n] = frag_kmajor(imgB, col0 + n * 16, ks, lane);
    else b[n] = frag_mnmajor(imgB, col0 + n * 16, ks, lane);
  }
}
template <bool DRY = false>
__device__ __forceinline__ void p8_mma(f32x4 (&acc)[8][4], int mh, const bf16x8 (&a)[4], const bf16x8 (&b)[4]) {
  if constexpr (DRY) {       // timing-only ablation: fragments stay live, matrix pipe idle
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" ::"v"(a[m]), "v"(b[m]));
    return;
  }
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
      acc[mh * 4 + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n], a[m], acc[mh * 4 + m][n], 0, 0, 0);
}

// One 256 x 256 tile (ti, tj) of the group; split: which part of the contraction (weight gradient only).
template <int MODE, int SCHED>
__device__ __forceinline__ void gemm_tile_p8(const MtGemmGroup& g, int ti, int tj, int split, int flags, int ksplits,
                                             size_t split_stride, uint8_t* lds) {
  constexpr bool A_KMAJ = (MODE != 2);
  constexpr bool B_KMAJ = (MODE == 0);
  constexpr int BASE = SCHED & 1;
  constexpr bool DRY = (SCHED & 8) != 0, NOFILL = (SCHED & 16) != 0;      // timing-only ablations
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int i0 = ti * BIGT, j0 = tj * BIGT;
  if (i0 >= g.I || j0 >= g.J) return;
  const uint16_t* A = reinterpret_cast<const uint16_t*>(g.A);
  const uint16_t* B = reinterpret_cast<const uint16_t*>(g.B);
  const int qi0 = i0 + wr * 128, qj0 = j0 + wc * 64;                     // this wave's block of the output
  const bool live = (qi0 < g.I) && (qj0 < g.J);      // wave-uniform; a wave outside the matrix multiplies the zeros staged for it
  const int offA = wr * 16384, offB = 32768 + (wc >> 1) * 16384, bcol0 = (wc & 1) * 64;

  f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool want_db = (MODE == 2) && (flags & MT_GEMM_FLAG_DBIAS) && tj == 0 && g.aux_out != nullptr;
  float db_acc = 0.f;

  const int nk_all = (g.KK + BK - 1) / BK;
  int kt_lo = 0, kt_hi = nk_all;
  if (MODE == 2 && ksplits > 1) {
    const int per = (nk_all + ksplits - 1) / ksplits;
    kt_lo = split * per;
    kt_hi = (kt_lo + per < nk_all) ? kt_lo + per : nk_all;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int n_it = kt_hi - kt_lo;
  auto issue = [&](int slot, int kt) {      // 8 LDS-DMA pieces per wave: tile kt -> the four images of stage `slot`
    uint8_t* st = lds + slot * BIG_STAGE_BYTES;
    int ln = lane;
    asm volatile("" : "+v"(ln));            // addresses are recomputed per tile instead of pinned across the loop
    stage_tile<A_KMAJ, 2>(st, A, g.lda, i0, g.I, kt * BK, g.KK, wave_u, ln);
    stage_tile<A_KMAJ, 2>(st + 16384, A, g.lda, i0 + 128, g.I, kt * BK, g.KK, wave_u, ln);
    stage_tile<B_KMAJ, 2>(st + 32768, B, g.ldb, j0, g.J, kt * BK, g.KK, wave_u, ln);
    stage_tile<B_KMAJ, 2>(st + 49152, B, g.ldb, j0 + 128, g.J, kt * BK, g.KK, wave_u, ln);
  };

  if constexpr (BASE == 0) {
  // Two barriers per K-tile: B1 publishes tile it, B2 (after this wave's last read of it) frees the stage, whose refill
  // then has the second half of this tile and the whole next one to land.
  if (n_it > 0) issue(0, kt_lo);
  if (n_it > 1) issue(1, kt_lo + 1);
  for (int it = 0; it < n_it; ++it) {
    const int cur = it & 1;
    const uint8_t* st = lds + cur * BIG_STAGE_BYTES;
    const uint8_t* imgA = st + offA;
    const uint8_t* imgB = st + offB;
    if (it + 1 < n_it) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 a0[4], a1[4], a2[4], a3[4], b0[4], b1[4];
    p8_read_a<A_KMAJ>(a0, imgA, 0, 0, lane);
    p8_read_b<B_KMAJ>(b0, imgB, bcol0, 0, lane);
    __builtin_amdgcn_sched_barrier(0);
    p8_read_a<A_KMAJ>(a1, imgA, 1, 0, lane);
    __builtin_amdgcn_sched_barrier(0);
    p8_mma<DRY>(acc, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    p8_read_a<A_KMAJ>(a2, imgA, 0, 1, lane);
    __builtin_amdgcn_sched_barrier(0);
    p8_mma<DRY>(acc, 1, a1, b0);
    __builtin_amdgcn_sched_barrier(0);
    p8_read_b<B_KMAJ>(b1, imgB, bcol0, 1, lane);
    p8_read_a<A_KMAJ>(a3, imgA, 1, 1, lane);
    if (want_db) db_acc += colsum_stage(st + (tid >> 8) * 16384, tid & 255);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // every wave holds its fragments: the stage may be refilled
    if (!NOFILL && it + 2 < n_it) issue(cur, kt_lo + it + 2);
    __builtin_amdgcn_sched_barrier(0);
    p8_mma<DRY>(acc, 0, a2, b1);
    p8_mma<DRY>(acc, 1, a3, b1);
  }
  } else {
  // One barrier per K-tile: it publishes tile it+1 (this wave's pieces of it are back: vmcnt(0), nothing younger is in
  // flight yet) and proves that every wave has read the last fragment of tile it, whose stage is refilled right after.
  // The fourth MFMA group of a tile is held back until the first reads of the next tile have been issued.
  bf16x8 a3[4], b1[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) { a3[m] = bf16x8{}; b1[m] = bf16x8{}; }      // tile "-1": adds +0 to the zeroed accumulators
  if (n_it > 0) issue(0, kt_lo);
  if (n_it > 1) issue(1, kt_lo + 1);
  if (n_it > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int it = 0; it < n_it; ++it) {
    const int cur = it & 1;
    const uint8_t* st = lds + cur * BIG_STAGE_BYTES;
    const uint8_t* imgA = st + offA;
    const uint8_t* imgB = st + offB;
    {
      bf16x8 a0[4], a1[4], a2[4], b0[4];
      p8_read_a<A_KMAJ>(a0, imgA, 0, 0, lane);
      p8_read_b<B_KMAJ>(b0, imgB, bcol0, 0, lane);
      __builtin_amdgcn_sched_barrier(0);
      p8_mma<DRY>(acc, 1, a3, b1);                      // (ks1, rows hi) of the previous tile
      __builtin_amdgcn_sched_barrier(0);
      p8_read_a<A_KMAJ>(a1, imgA, 1, 0, lane);
      __builtin_amdgcn_sched_barrier(0);
      p8_mma<DRY>(acc, 0, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      p8_read_a<A_KMAJ>(a2, imgA, 0, 1, lane);
      p8_read_b<B_KMAJ>(b1, imgB, bcol0, 1, lane);
      __builtin_amdgcn_sched_barrier(0);
      p8_mma<DRY>(acc, 1, a1, b0);
      __builtin_amdgcn_sched_barrier(0);
      p8_read_a<A_KMAJ>(a3, imgA, 1, 1, lane);
      __builtin_amdgcn_sched_barrier(0);
      p8_mma<DRY>(acc, 0, a2, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (want_db) db_acc += colsum_stage(st + (tid >> 8) * 16384, tid & 255);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!NOFILL && it + 2 < n_it) issue(cur, kt_lo + it + 2);
  }
  p8_mma<DRY>(acc, 1, a3, b1);

  }
  // ------------------------------------------------------------------------------ epilogues
  const int fq = lane >> 4, fr = lane & 15;
  if (MODE == 2) {
    float* C = reinterpret_cast<float*>(g.C) + (size_t)split * split_stride;
    if (live) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int row = qi0 + m * 16 + fr;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const int col = qj0 + n * 16 + fq * 4;
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
    }
    if (want_db) {
      float* red = reinterpret_cast<float*>(lds);    // all waves are past the last K-step barrier
      red[tid] = db_acc;
      __syncthreads();
      float* out = reinterpret_cast<float*>(g.aux_out) + (size_t)split * split_stride;
      const int t = tid & 255, row = i0 + (tid >> 8) * 128 + t;
      if (t < 128 && row < g.I) out[row] = red[tid] + red[tid + 128];
    }
    return;
  }
  if (!live) return;
  {
    uint8_t* img = lds + wave * 16384;       // this wave's own [128][64] bf16 staging image (wave-private: no barrier)
    const bool relu = (MODE == 0) && (flags & MT_GEMM_FLAG_RELU);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int lcol = n * 16 + fq * 4;
      float bb[4] = {0.f, 0.f, 0.f, 0.f};
      if (MODE == 0 && g.aux != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (qj0 + lcol + j < g.J) bb[j] = reinterpret_cast<const float*>(g.aux)[qj0 + lcol + j];
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int lrow = m * 16 + fr;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[m][n][j] + bb[j];
          if (relu) v[j] = fmaxf(v[j], 0.f);
        }
        u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(img + kmaj_off(lrow, lcol >> 3) + (lcol & 7) * 2) = pk;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    uint16_t* C = reinterpret_cast<uint16_t*>(g.C);
    const uint16_t* msk = (MODE == 1 && (flags & MT_GEMM_FLAG_MASK)) ? reinterpret_cast<const uint16_t*>(g.aux)
                                                                    : nullptr;
#pragma unroll 1
    for (int t0 = 0; t0 < 16; t0 += 8) {
      u32x4 hv[8];
      if (msk != nullptr) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int q = lane + 64 * (t0 + t);
          const int row = qi0 + (q >> 3), col = qj0 + (q & 7) * 8;
          const bool ok = (row < g.I && col < g.J);
          hv[t] = *reinterpret_cast<const u32x4*>(msk + (ok ? ((size_t)row * g.ldaux + col) : (size_t)0));
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int q = lane + 64 * (t0 + t);
        const int lrow = q >> 3, ch = q & 7;
        const int row = qi0 + lrow, col = qj0 + ch * 8;
        u32x4 v = *reinterpret_cast<const u32x4*>(img + kmaj_off(lrow, ch));
        if (msk != nullptr) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t hw = hv[t][e];
            const uint32_t lo_ok = ((hw & 0x8000u) == 0u && (hw & 0x7FFFu) != 0u) ? 0x0000FFFFu : 0u;
            const uint32_t hi_ok = ((hw & 0x80000000u) == 0u && (hw & 0x7FFF0000u) != 0u) ? 0xFFFF0000u : 0u;
            v[e] &= (lo_ok | hi_ok);
          }
        }
        if (row < g.I && col < g.J) *reinterpret_cast<u32x4*>(C + (size_t)row * g.ldc + col) = v;
      }
    }
  }
}

// Block -> tile as in mt_gemm_single_kernel (whole row panels per XCD); the contraction split is the fastest index.
// Block -> tile: block b runs on XCD b % 8, which takes a contiguous run of the tile sequence; the sequence walks the
// grid in bands of `band` row tiles, row tile fastest inside a band, so the ~32 tiles an XCD has in flight form a
// band x (32 / band) rectangle that shares band + 32 / band operand panels in its L2 instead of 33.
template <int MODE, int SCHED>
__global__ __launch_bounds__(P8_THREADS) void mt_gemm_p8_kernel(MtGemmGroup g, int tiles_i, int tiles_j, int ksplits,
                                                                size_t split_stride, int flags, int band) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * BIG_STAGE_BYTES];
  const int b = blockIdx.x;
  const int n = tiles_i * tiles_j * ksplits;
  const int per = (n + 7) / 8;
  const int u = (b % 8) * per + b / 8;
  if (u >= n) return;
  const int sp = u % ksplits, t = u / ksplits;
  const int in_band = band * tiles_j;
  const int first = (t / in_band) * band;
  const int rows = (tiles_i - first < band) ? tiles_i - first : band;      // the last band may be narrower
  const int r = t - (t / in_band) * in_band;
  gemm_tile_p8<MODE, SCHED>(g, first + r % rows, r / rows, sp, flags, ksplits, split_stride, lds);
}

// The same tile for PACKS (tile lists of 256-tiles, one workgroup of eight waves per entry): a pack's layers of 512 columns
// and more are eight or more K-tiles long, enough for the pipelined loop to pay (nstage 258 of the grouped launchers).
template <int MODE, int SCHED>
__device__ __forceinline__ void gemm_tile_p8_grouped(const MtGemmGroup* __restrict__ groups, const MtTile tile, int step,
                                                     int flags, int arow0, int i_override, int ksplits, size_t split_stride,
                                                     uint8_t* lds) {
  if (tile.group < 0) return;
  MtGemmGroup g = groups[tile.group];
  if (step >= g.active_steps) return;
  if (i_override > 0) {
    g.I = i_override;
    g.A = reinterpret_cast<const uint16_t*>(g.A) + (size_t)arow0 * g.lda;
  }
  const int split = (MODE == 2) ? (tile.ti >> 8) : 0;
  const int ti = (MODE == 2) ? (tile.ti & 0xFF) : (int)tile.ti;
  gemm_tile_p8<MODE, SCHED>(g, ti, tile.tj, split, flags, ksplits, split_stride, lds);
}

template <int MODE, int SCHED>
__global__ __launch_bounds__(P8_THREADS) void mt_gemm_p8_grouped_kernel(const MtGemmGroup* __restrict__ groups,
                                                                        const MtTile* __restrict__ tiles, int step, int flags,
                                                                        int arow0, int i_override, int ksplits,
                                                                        size_t split_stride,
                                                                        const int* __restrict__ step_base) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * BIG_STAGE_BYTES];
  gemm_tile_p8_grouped<MODE, SCHED>(groups, tiles[blockIdx.x], mt_step(step, step_base), flags, arow0, i_override, ksplits,
                                    split_stride, lds);
}

__global__ __launch_bounds__(P8_THREADS) void mt_gemm_p8_grouped_bwd_kernel(const MtGemmGroup* __restrict__ wgroups,
                                                                            const MtTile* __restrict__ wtiles, int n_wtiles,
                                                                            const MtGemmGroup* __restrict__ dgroups,
                                                                            const MtTile* __restrict__ dtiles, int step_imm,
                                                                            int ksplits, size_t split_stride,
                                                                            const int* __restrict__ step_base) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * BIG_STAGE_BYTES];
  const int step = mt_step(step_imm, step_base);
  if ((int)blockIdx.x < n_wtiles)
    gemm_tile_p8_grouped<2, 0>(wgroups, wtiles[blockIdx.x], step, MT_GEMM_FLAG_DBIAS, 0, 0, ksplits, split_stride, lds);
  else
    gemm_tile_p8_grouped<1, 0>(dgroups, dtiles[blockIdx.x - n_wtiles], step, MT_GEMM_FLAG_MASK, 0, 0, 1, 0, lds);
}

// Large single GEMMs take the eight-wave 256-tile form once they have this many 256-tiles (0: never).
static int p8_limit = 256;
static int p8_min_tiles() { return p8_limit; }
// schedule (-1: by mode -- one barrier per K-tile when both operands are K-major, two otherwise: the forward GEMM gains
// 4-8 % from the single barrier, the input- and weight-gradient GEMMs lose 10-15 % to the shorter time their refills get;
// 0 / 1 force the two-barrier / one-barrier form, +8 / +16 are timing-only ablations without MFMAs / without refills)
// and the band height in tiles.  Tried on 8192^3 and 16384 x 8192^2 and dropped: a raised wave priority around the MFMA
// groups (no change), the refill's address arithmetic woven between MFMAs with scheduling-group directives (no change),
// and five 32-deep stages filling all 160 KiB of LDS so that four tiles instead of one are in flight (forward 7-10 %
// slower: 64-byte operand rows request every 128-byte line twice; weight gradient 30 % slower).  Timing-only runs put
// the loop without refills at 690 us and the refills without MFMAs at 540 us for the 8192^3 forward GEMM that takes 880.
static int p8_opts[2] = {-1, 4};
extern "C" int mt_gemm_p8_option(int which, int value) {      // experiments: returns the value in force before
  if (which < 0 || which > 1) return -2;
  const int was = p8_opts[which];
  if (value >= -1) p8_opts[which] = which == 1 && value < 1 ? 1 : value;
  return was;
}
extern "C" int mt_gemm_p8_min_tiles(int set) {      // set >= 0 replaces the threshold; returns the one in force before
  const int was = p8_min_tiles();
  if (set >= 0) p8_limit = set;
  return was;
}
// Also needed: a contraction of at least four K-tiles per workgroup (66560 x 576 x 192, an FT-Transformer projection: 48 us
// against 43 us on 128-tiles; 4096 x 4096 x 256: 18 us against 21 us) and an output at least two tiles wide and high.  With
// fewer tiles than CUs the 128-tiles win by up to 1.5x (16384 x 512 x 2048: 128 tiles, 68 us against 45 us).
static bool p8_wanted(int rows, int cols, int depth, int ksplits) {
  const int lim = p8_min_tiles();
  if (lim <= 0) return false;
  const long n = (long)((rows + BIGT - 1) / BIGT) * ((cols + BIGT - 1) / BIGT) * ksplits;
  return rows >= 2 * BIGT && cols >= 2 * BIGT && depth / ksplits >= 4 * BK && n >= lim;
}

template <int MODE, int NSTAGE>
__global__ __launch_bounds__(NTHREADS) void mt_gemm_kernel(const MtGemmGroup* __restrict__ groups,
                                                           const MtTile* __restrict__ tiles, int step, int flags,
                                                           int arow0, int i_override, int ksplits,
                                                           size_t split_stride, const int* __restrict__ step_base) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[NSTAGE * STAGE_BYTES];
  gemm_tile<MODE, NSTAGE>(groups, tiles[blockIdx.x], mt_step(step, step_base), flags, arow0, i_override, ksplits, split_stride, lds);
}

// Backward of one layer in ONE launch: the weight-gradient tiles and the input-gradient tiles both
// consume dZ of that layer and do not depend on each other.
template <int NSTAGE>
__global__ __launch_bounds__(NTHREADS) void mt_gemm_bwd_kernel(const MtGemmGroup* __restrict__ wgroups,
                                                               const MtTile* __restrict__ wtiles, int n_wtiles,
                                                               const MtGemmGroup* __restrict__ dgroups,
                                                               const MtTile* __restrict__ dtiles, int n_dtiles,
                                                               int step_imm, int ksplits, size_t split_stride,
                                                               const int* __restrict__ step_base, MtHeadReduceArgs hr) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[NSTAGE * STAGE_BYTES];
  const int step = mt_step(step_imm, step_base);
  const int b = (int)blockIdx.x;
  if (b < n_wtiles)
    gemm_tile<2, NSTAGE>(wgroups, wtiles[b], step, MT_GEMM_FLAG_DBIAS, 0, 0, ksplits, split_stride, lds);
  else if (b < n_wtiles + n_dtiles)
    gemm_tile<1, NSTAGE>(dgroups, dtiles[b - n_wtiles], step, MT_GEMM_FLAG_MASK, 0, 0, 1, 0, lds);
  else      // riders of the last layer's launch: add up the head kernel's partial records, one workgroup per slot
    mt_head_reduce_body(hr.slots, hr.ids, step, hr.scratch, hr.max_chunks, hr.rec, b - n_wtiles - n_dtiles);
}

// nstage: 2 = throughput form (two workgroups per CU), 4 = latency form (one workgroup per CU, three
// tiles in flight); 0 lets the launcher choose from the tile count.  256 / 258 = the tile list holds 256x256 tiles (four
// waves and the simple two-barrier loop / eight waves with pipelined fragment reads).
static int pick_nstage(int nstage, int ntiles) {
  if (nstage == 2 || nstage == 4) return nstage;
  return ntiles <= 320 ? 4 : 2;
}

extern "C" int mt_gemm_launch(int mode, const MtGemmGroup* groups, const MtTile* tiles, int ntiles, int step,
                              int flags, int arow0, int i_override, int ksplits, size_t split_stride, int nstage,
                              hipStream_t stream) {
  if (ntiles <= 0) return 0;
  if (ksplits < 1 || ksplits > 127 || (mode != 2 && ksplits != 1)) return -5;
  dim3 grid(ntiles), block(NTHREADS);
  if (nstage == 258) {      // 256-tiles on eight waves (one barrier per K-tile when both operands are K-major, else two)
    dim3 block8(P8_THREADS);
    switch (mode) {
      case 0: hipLaunchKernelGGL((mt_gemm_p8_grouped_kernel<0, 1>), grid, block8, 0, stream, groups, tiles, step, flags, arow0,
                                 i_override, 1, (size_t)0, mt_step_base_tls); break;
      case 1: hipLaunchKernelGGL((mt_gemm_p8_grouped_kernel<1, 0>), grid, block8, 0, stream, groups, tiles, step, flags, arow0,
                                 i_override, 1, (size_t)0, mt_step_base_tls); break;
      case 2: hipLaunchKernelGGL((mt_gemm_p8_grouped_kernel<2, 0>), grid, block8, 0, stream, groups, tiles, step, flags, arow0,
                                 i_override, ksplits, split_stride, mt_step_base_tls); break;
      default: return -1;
    }
    return (int)hipGetLastError();
  }
  if (nstage == 256) {
#define MT_GEMM_BIG(M, KS, SS) \
  hipLaunchKernelGGL((mt_gemm256_kernel<M>), grid, block, 0, stream, groups, tiles, step, flags, arow0, i_override, KS, SS, \
                     mt_step_base_tls)
    switch (mode) {
      case 0: MT_GEMM_BIG(0, 1, (size_t)0); break;
      case 1: MT_GEMM_BIG(1, 1, (size_t)0); break;
      case 2: MT_GEMM_BIG(2, ksplits, split_stride); break;
      default: return -1;
    }
#undef MT_GEMM_BIG
    return (int)hipGetLastError();
  }
  const bool deep = pick_nstage(nstage, ntiles) == 4;
#define MT_GEMM_GO(M, N, KS, SS) \
  hipLaunchKernelGGL((mt_gemm_kernel<M, N>), grid, block, 0, stream, groups, tiles, step, flags, arow0, i_override, KS, SS, \
                     mt_step_base_tls)
  switch (mode) {
    case 0: if (deep) MT_GEMM_GO(0, 4, 1, (size_t)0); else MT_GEMM_GO(0, 2, 1, (size_t)0); break;
    case 1: if (deep) MT_GEMM_GO(1, 4, 1, (size_t)0); else MT_GEMM_GO(1, 2, 1, (size_t)0); break;
    case 2: if (deep) MT_GEMM_GO(2, 4, ksplits, split_stride); else MT_GEMM_GO(2, 2, ksplits, split_stride); break;
    default: return -1;
  }
#undef MT_GEMM_GO
  return (int)hipGetLastError();
}

// hr.n_slots > 0 appends that many workgroups which reduce the head kernel's partial records (see mt_bodies.h).
extern "C" int mt_gemm_bwd_head_launch(const MtGemmGroup* wgroups, const MtTile* wtiles, int n_wtiles,
                                       const MtGemmGroup* dgroups, const MtTile* dtiles, int n_dtiles, int step,
                                       int ksplits, size_t split_stride, int nstage, MtHeadReduceArgs hr,
                                       hipStream_t stream) {
  if (n_wtiles + n_dtiles + hr.n_slots <= 0) return 0;
  if (ksplits < 1 || ksplits > 127) return -5;
  if ((nstage == 256 || nstage == 258) && hr.n_slots > 0) return -6;
  dim3 grid(n_wtiles + n_dtiles + hr.n_slots), block(NTHREADS);
  if (nstage == 258) {
    hipLaunchKernelGGL(mt_gemm_p8_grouped_bwd_kernel, grid, dim3(P8_THREADS), 0, stream, wgroups, wtiles, n_wtiles, dgroups,
                       dtiles, step, ksplits, split_stride, mt_step_base_tls);
    return (int)hipGetLastError();
  }
  if (nstage == 256) {
    hipLaunchKernelGGL(mt_gemm256_bwd_kernel, grid, block, 0, stream, wgroups, wtiles, n_wtiles, dgroups, dtiles, step,
                       ksplits, split_stride, mt_step_base_tls);
    return (int)hipGetLastError();
  }
  if (pick_nstage(nstage, n_wtiles + n_dtiles) == 4)
    hipLaunchKernelGGL(mt_gemm_bwd_kernel<4>, grid, block, 0, stream, wgroups, wtiles, n_wtiles, dgroups, dtiles, n_dtiles,
                       step, ksplits, split_stride, mt_step_base_tls, hr);
  else
    hipLaunchKernelGGL(mt_gemm_bwd_kernel<2>, grid, block, 0, stream, wgroups, wtiles, n_wtiles, dgroups, dtiles, n_dtiles,
                       step, ksplits, split_stride, mt_step_base_tls, hr);
  return (int)hipGetLastError();
}

extern "C" int mt_gemm_bwd_launch(const MtGemmGroup* wgroups, const MtTile* wtiles, int n_wtiles,
                                  const MtGemmGroup* dgroups, const MtTile* dtiles, int n_dtiles, int step,
                                  int ksplits, size_t split_stride, int nstage, hipStream_t stream) {
  MtHeadReduceArgs none = {nullptr, nullptr, nullptr, 0, 0, 0};
  return mt_gemm_bwd_head_launch(wgroups, wtiles, n_wtiles, dgroups, dtiles, n_dtiles, step, ksplits, split_stride,
                                 nstage, none, stream);
}

// ----------------------------------------------------------------------------------------------------
// Stand-alone weight gradient of one dense layer:  dW[out][in] = dY[M][out]^T . X[M][in]  (+ db = column sums of dY),
// contraction over M tokens split `ksplits` ways into f32 slabs that the caller adds.  Same tile code as the grouped
// path (mode TN), but the single group and the tile coordinates come from kernel arguments and blockIdx, so no
// descriptor table has to be uploaded per call (this runs inside autograd, under hipGraph capture).  For the long,
// skinny shapes of token-major Transformer layers (M ~ 10^4..10^5, out/in a few hundred) the BLAS library picks a
// 64x64 tile without splitting the contraction and leaves most CUs idle.
template <int NSTAGE>
__global__ __launch_bounds__(NTHREADS) void mt_wgrad_single_kernel(MtGemmGroup g, int tiles_i, int ksplits,
                                                                   size_t split_stride, int flags) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[NSTAGE * STAGE_BYTES];
  const int b = blockIdx.x;
  const int sp = b % ksplits, t = b / ksplits;
  MtTile tile;
  tile.group = 0;
  tile.ti = (int16_t)((t % tiles_i) | (sp << 8));
  tile.tj = (int16_t)(t / tiles_i);
  gemm_tile<2, NSTAGE>(&g, tile, 0, flags, 0, 0, ksplits, split_stride, lds);
}

// C: f32 [ksplits][split_stride] slabs, slab s holds dW partial at offset 0 (row stride `in`); db (may be null): f32,
// slab s of it at db + s * split_stride.  M, out, in: out % 8 == 0 and in % 8 == 0 (16-byte operand rows).
extern "C" int mt_wgrad_single_launch(const void* dY, const void* X, float* C, float* db, int M, int out, int in,
                                      int ksplits, size_t split_stride, hipStream_t stream) {
  if (M <= 0 || out <= 0 || in <= 0) return 0;
  if ((out & 7) || (in & 7) || ksplits < 1 || ksplits > 127) return -2;
  const int tiles_i = (out + BM - 1) / BM, tiles_j = (in + BN - 1) / BN;
  if (tiles_i > 255) return -2;
  MtGemmGroup g = {};
  g.A = dY; g.B = X; g.C = C; g.aux = nullptr; g.aux_out = db;
  g.I = out; g.J = in; g.KK = M;
  g.lda = out; g.ldb = in; g.ldc = in; g.ldaux = 0;
  g.active_steps = 1;
  const int nblocks = tiles_i * tiles_j * ksplits;
  const int flags = db != nullptr ? MT_GEMM_FLAG_DBIAS : 0;
  if (p8_wanted(out, in, M, ksplits)) {
    const int t_i = (out + BIGT - 1) / BIGT, t_j = (in + BIGT - 1) / BIGT;
    const int nb = ((t_i * t_j * ksplits + 7) / 8) * 8;
#define MT_P8_GO(M, S) \
  hipLaunchKernelGGL((mt_gemm_p8_kernel<M, S>), dim3(nb), dim3(P8_THREADS), 0, stream, g, t_i, t_j, ksplits, split_stride, flags, \
                     p8_opts[1])
    switch (p8_opts[0] < 0 ? 0 : p8_opts[0]) {
      case 1: MT_P8_GO(2, 1); break;
      default: MT_P8_GO(2, 0); break;
    }
#undef MT_P8_GO
    return (int)hipGetLastError();
  }
  if (nblocks <= 320)
    hipLaunchKernelGGL(mt_wgrad_single_kernel<4>, dim3(nblocks), dim3(NTHREADS), 0, stream, g, tiles_i, ksplits,
                       split_stride, flags);
  else
    hipLaunchKernelGGL(mt_wgrad_single_kernel<2>, dim3(nblocks), dim3(NTHREADS), 0, stream, g, tiles_i, ksplits,
                       split_stride, flags);
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------------------------------
// Stand-alone dense-layer GEMMs for modules that are not part of a grouped pack (Transformer projections under bf16
// autocast): the same tile code, the single group passed by value and the tile coordinates taken from blockIdx, so no
// descriptor table is uploaded per call (these run inside autograd, under hipGraph capture).  The BLAS library's
// kernels for these shapes are "stream-K" kernels whose workgroups wait on each other inside one launch; two of them
// launched from different HIP streams share the CUs and can starve each other for good.  These kernels have no
// workgroup-to-workgroup dependency, so any number of streams may run them side by side.
//   mode 0: C[M][N] = A[M][K] . B[N][K]^T (+ bias f32[N]) (ReLU if flags & 1)        forward
//   mode 1: C[M][N] = A[M][K] . B[K][N]  (zeroed where aux[M][ldaux] <= 0 if flags & 1)  input gradient (A = dY, B = W)
// Block -> tile: consecutive blocks walk the column tiles of one row panel (they share the A panel in L2), and with
// the number of column tiles padded to a multiple of 8 a row panel's tiles stay on one XCD.
template <int MODE, int NSTAGE>
__global__ __launch_bounds__(NTHREADS) void mt_gemm_single_kernel(MtGemmGroup g, int tiles_i, int tiles_j, int flags) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[NSTAGE * STAGE_BYTES];
  const int b = blockIdx.x;
  // XCD-aware order: block b runs on XCD b % 8; give every XCD whole row panels (tiles_j consecutive column tiles)
  const int n = tiles_i * tiles_j;
  const int per = (n + 7) / 8;
  const int t = (b % 8) * per + b / 8;
  if (t >= n) return;
  MtTile tile;
  tile.group = 0;
  tile.ti = (int16_t)(t / tiles_j);
  tile.tj = (int16_t)(t % tiles_j);
  gemm_tile<MODE, NSTAGE>(&g, tile, 0, flags, 0, 0, 1, 0, lds);
}

extern "C" int mt_gemm_single_launch(int mode, const void* A, const void* B, void* C, const void* aux, int M, int N, int K,
                                     int lda, int ldb, int ldc, int ldaux, int flags, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if ((lda & 7) || (ldb & 7) || (ldc & 7) || (mode != 0 && mode != 1)) return -2;
  if (mode == 1 && (flags & MT_GEMM_FLAG_MASK) && (aux == nullptr || (ldaux & 7))) return -2;
  const int tiles_i = (M + BM - 1) / BM, tiles_j = (N + BN - 1) / BN;
  if (tiles_i > 32767 || tiles_j > 32767) return -2;
  MtGemmGroup g = {};
  g.A = A; g.B = B; g.C = C; g.aux = aux; g.aux_out = nullptr;
  g.I = M; g.J = N; g.KK = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
  g.active_steps = 1;
  if (p8_wanted(M, N, K, 1)) {
    const int t_i = (M + BIGT - 1) / BIGT, t_j = (N + BIGT - 1) / BIGT;
    const int nb = ((t_i * t_j + 7) / 8) * 8;
#define MT_P8_GO(M, S) \
  hipLaunchKernelGGL((mt_gemm_p8_kernel<M, S>), dim3(nb), dim3(P8_THREADS), 0, stream, g, t_i, t_j, 1, (size_t)0, flags, p8_opts[1])
    const int sc = p8_opts[0] < 0 ? (mode == 0 ? 1 : 0) : p8_opts[0];
#define MT_P8_PICK(M) \
  switch (sc) { case 1: MT_P8_GO(M, 1); break; case 8: MT_P8_GO(M, 8); break; case 9: MT_P8_GO(M, 9); break; \
                case 16: MT_P8_GO(M, 16); break; case 17: MT_P8_GO(M, 17); break; default: MT_P8_GO(M, 0); break; }
    if (mode == 0) { MT_P8_PICK(0) } else { MT_P8_PICK(1) }
#undef MT_P8_PICK
#undef MT_P8_GO
    return (int)hipGetLastError();
  }
  const int n = tiles_i * tiles_j;
  const int nblocks = ((n + 7) / 8) * 8;
  dim3 grid(nblocks), block(NTHREADS);
  const bool deep = n <= 320;
  if (mode == 0) {
    if (deep) hipLaunchKernelGGL((mt_gemm_single_kernel<0, 4>), grid, block, 0, stream, g, tiles_i, tiles_j, flags);
    else hipLaunchKernelGGL((mt_gemm_single_kernel<0, 2>), grid, block, 0, stream, g, tiles_i, tiles_j, flags);
  } else {
    if (deep) hipLaunchKernelGGL((mt_gemm_single_kernel<1, 4>), grid, block, 0, stream, g, tiles_i, tiles_j, flags);
    else hipLaunchKernelGGL((mt_gemm_single_kernel<1, 2>), grid, block, 0, stream, g, tiles_i, tiles_j, flags);
  }
  return (int)hipGetLastError();
}
