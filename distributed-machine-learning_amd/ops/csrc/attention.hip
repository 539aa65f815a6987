// Fused short-sequence self-attention, forward and backward (gfx950, bf16 MFMA, f32 softmax/accumulate).
//
// One workgroup (4 waves) owns one (batch, head): Q, K, V (and dO in backward) are staged once into LDS
// as [row][32 | 64 | 128] bf16 images; the whole S x S score tile lives in registers.  S <= 128, head widths 1..64 (any
// value; widths that are not multiples of 8 are staged element by element) and 65..128 for S <= 96.
//
// Register-level dataflow (no LDS round trip for the probabilities in forward):
//   * scores are computed TRANSPOSED, S^T[key][query] = K . Q^T, with v_mfma_f32_16x16x32_bf16.  In the
//     accumulator layout a lane then holds, for ONE query (lane & 15), four consecutive keys per 16-key
//     block -- so the row-wise softmax is a register reduction plus two lane shuffles (xor 16, 32);
//   * those same registers, packed to bf16, are directly the A operand of O = P . V: the contraction
//     index (key) may be enumerated in any order as long as both operands agree, so V's B fragment is
//     fetched with ds_read_b64_tr_b16 from exactly the key rows the P registers hold
//     (32*step + 4*g + {0..3} and 32*step + 16 + 4*g + {0..3} for lane group g).
// Backward recomputes P from Q, K and the saved row statistic (log-sum-exp, or the ReLU row sum),
// forms dP^T = V . dO^T in the same layout, reduces delta = rowsum(dP * P) in registers, turns both into
// dS^T, gets dQ^T = K^T . dS with the same register trick, and only then writes P^T and dS^T (bf16) to LDS
// for the two products that contract over queries: dV = P^T . dO and dK = dS^T . Q.
//
// kind 0: softmax(QK^T * scale); kind 1: relu(QK^T * scale) / (rowsum + eps).
// keep: optional uint8 [BH][S][S] attention-dropout mask (1 = keep), applied with scale 1/(1-p).
#include "mt_common.h"
#include <stdlib.h>

#define AT_THREADS 256
// Q/K/V/dO images are [rows_pad][64] bf16 (8 chunks of 16 B per row) or, when both head dims are <= 32, [rows_pad][32]
// (4 chunks): half the LDS, so twice the workgroups per CU for the small heads of narrow models (d_model/h = 24..32).
#define AT_RELU_EPS 1e-6f

typedef __attribute__((address_space(3))) uint8_t at_lds_u8;
typedef __attribute__((ext_vector_type(8))) short at_s16x8;

__device__ __forceinline__ int img_off(int row, int chunk, int cpr) {
  // 64-byte rows put rows r and r+4 on the same banks, so their swizzle also folds in bits 2..3 of the row
  const int swz = (cpr == 4) ? ((row ^ (row >> 2)) & 3) : (row & (cpr - 1));
  return row * cpr * 16 + ((chunk ^ swz) << 4);
}

// global [rows][cols] (row stride ld elements; bf16, or f32 converted on the way in) -> zero-padded LDS image
// [rows_pad][PCHQ * 8] bf16.  Head widths that are multiples of 8 on rows that keep every 8-element chunk 16-byte
// aligned are loaded as vectors; any other width (heads of 2, 4, 6, 10, 12, 20 ... elements, which a search space with
// free head counts produces) element by element -- a head is at most S x 128 values, the MFMA work is the same.
template <int PCHQ, bool F32>
__device__ __forceinline__ void stage_image(uint8_t* lds, const void* __restrict__ src_, int rows, int cols, long ld,
                                            int rows_pad, int tid) {
  const bool vec = ((cols & 7) == 0) && ((ld & 7) == 0) &&
                   ((reinterpret_cast<uintptr_t>(src_) & 15) == 0);
  for (int q = tid; q < rows_pad * PCHQ; q += AT_THREADS) {
    const int r = q / PCHQ, c = q % PCHQ;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < rows && c * 8 < cols && !vec) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = c * 8 + e;
        float x = 0.f;
        if (col < cols) {
          if constexpr (F32) x = reinterpret_cast<const float*>(src_)[(size_t)r * ld + col];
          else x = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(src_)[(size_t)r * ld + col]);
        }
        f[e] = x;
      }
      v = u32x4{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
    } else if (r < rows && c * 8 < cols) {
      if constexpr (F32) {
        const float* src = reinterpret_cast<const float*>(src_) + (size_t)r * ld + c * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        v = u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
      } else {
        v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(src_) + (size_t)r * ld + c * 8);
      }
    }
    *reinterpret_cast<u32x4*>(lds + img_off(r, c, PCHQ)) = v;
  }
}

// Issue-then-commit staging.  stage_image above waits for each chunk before it asks for the next one, so a workgroup
// that stages four images of two chunks per thread pays eight global-memory latencies in a row before its first MFMA;
// these problems are small enough that this wait is most of the workgroup's life.  Here every 16-byte chunk of every
// image a thread owns is requested first (stage_issue, into registers), and only then converted and written to LDS
// (stage_commit): one latency per workgroup.  Images whose rows cannot be read as aligned vectors keep the
// element-by-element path, taken inside stage_commit.
template <int PCHQ, bool F32, int SP>
struct StageRegs {
  static constexpr int NIT = (SP * PCHQ + AT_THREADS - 1) / AT_THREADS;
  u32x4 lo[NIT];
  u32x4 hi[F32 ? NIT : 1];        // f32 operands: elements 4..7 of the chunk
  bool vec;
};

template <int PCHQ, bool F32, int SP>
__device__ __forceinline__ void stage_issue(StageRegs<PCHQ, F32, SP>& b, const void* __restrict__ src_, int rows, int cols,
                                            long ld, int tid) {
  b.vec = ((cols & 7) == 0) && ((ld & 7) == 0) && ((reinterpret_cast<uintptr_t>(src_) & 15) == 0);
  if (!b.vec) return;
#pragma unroll
  for (int it = 0; it < StageRegs<PCHQ, F32, SP>::NIT; ++it) {
    const int q = tid + it * AT_THREADS, r = q / PCHQ, c = q % PCHQ;
    b.lo[it] = u32x4{0u, 0u, 0u, 0u};
    if constexpr (F32) b.hi[it] = u32x4{0u, 0u, 0u, 0u};
    if (q < SP * PCHQ && r < rows && c * 8 < cols) {
      if constexpr (F32) {
        const float* src = reinterpret_cast<const float*>(src_) + (size_t)r * ld + c * 8;
        b.lo[it] = *reinterpret_cast<const u32x4*>(src);
        b.hi[it] = *reinterpret_cast<const u32x4*>(src + 4);
      } else {
        b.lo[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(src_) + (size_t)r * ld + c * 8);
      }
    }
  }
}

template <int PCHQ, bool F32, int SP>
__device__ __forceinline__ void stage_commit(const StageRegs<PCHQ, F32, SP>& b, uint8_t* lds, const void* __restrict__ src_,
                                             int rows, int cols, long ld, int tid) {
  if (!b.vec) { stage_image<PCHQ, F32>(lds, src_, rows, cols, ld, SP, tid); return; }
#pragma unroll
  for (int it = 0; it < StageRegs<PCHQ, F32, SP>::NIT; ++it) {
    const int q = tid + it * AT_THREADS, r = q / PCHQ, c = q % PCHQ;
    if (q < SP * PCHQ) {
      u32x4 v = b.lo[it];
      if constexpr (F32) {
        const f32x4 x = __builtin_bit_cast(f32x4, b.lo[it]), y = __builtin_bit_cast(f32x4, b.hi[it]);
        v = u32x4{pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
      }
      *reinterpret_cast<u32x4*>(lds + img_off(r, c, PCHQ)) = v;
    }
  }
}

// How many images are in flight at once: all of them while that is at most 24 registers quads per thread, else two.
template <int PCHQ, bool F32, int SP>
constexpr bool stage_all_at_once() { return StageRegs<PCHQ, F32, SP>::NIT * (F32 ? 2 : 1) * 4 <= 24; }

// Where problem bh = (b, head) of a [B, S, H * d] operand with row stride ld starts: element offset.
__device__ __forceinline__ size_t at_base(int bh, int H, int S, long ld, int d) {
  const int b = bh / H, hh = bh - b * H;
  return (size_t)b * S * ld + (size_t)hh * d;
}

// rows = M or N index (16 of them from row0), contraction contiguous: one ds_read_b128
__device__ __forceinline__ bf16x8 frag_rows(const uint8_t* lds, int row0, int ks, int lane, int cpr) {
  const int row = row0 + (lane & 15);
  const int chunk = ks * 4 + (lane >> 4);
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds + img_off(row, chunk, cpr)));
}

// contraction index = image ROW, M/N index = image column (16 of them from col0): two transposed reads.
// The lane group g = lane >> 4 takes contraction rows rbase_lo + {0..3} and rbase_hi + {0..3}.
__device__ __forceinline__ bf16x8 frag_cols(const uint8_t* lds, int col0, int rbase_lo, int rbase_hi, int lane, int cpr) {
  const int i = lane & 15, q = i >> 2, p = i & 3;
  const int c0 = (col0 >> 3) + (p >> 1);
  const at_lds_u8* a0 = (const at_lds_u8*)(lds + img_off(rbase_lo + q, c0, cpr) + 8 * (p & 1));
  const at_lds_u8* a1 = (const at_lds_u8*)(lds + img_off(rbase_hi + q, c0, cpr) + 8 * (p & 1));
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  at_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack_frag(const f32x4& a, const f32x4& b) {
  u32x4 v = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ float group_sum(float v) {   // across the 4 lane groups holding one query
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// NKB counts the blocks of 16 keys (queries) that hold any real token; images and contraction steps come in pairs of
// blocks (the 16x16x32 products take 32 keys a step), so an odd count is padded by one block of zero rows.
constexpr int at_padded_blocks(int nkb) { return (nkb + 1) / 2 * 2; }

// Scores for 16 queries (block qb) against all keys, transposed: acc[kb][j] = S[query][key = 16kb + 4g + j].
template <int NKB, int PCHQ>
__device__ __forceinline__ void scores_T(f32x4 (&acc)[NKB], const uint8_t* ldsK, const uint8_t* ldsQ, int qb, int dsteps,
                                         int lane) {
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) acc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ks = 0; ks < dsteps; ++ks) {
    const bf16x8 qf = frag_rows(ldsQ, qb * 16, ks, lane, PCHQ);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const bf16x8 kf = frag_rows(ldsK, kb * 16, ks, lane, PCHQ);
      acc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, acc[kb], 0, 0, 0);
    }
  }
}

// Dropout multipliers (on or 0) of keys key0 .. key0 + 3 for one query: from the stored byte mask if there is one, else
// drawn from the seed words at the element's index in the [BH, S, S] mask.  Straight to floats: the vector ALUs set
// these kernels' time, and flags packed into a word and picked apart again were a good part of their instructions.
// Padding keys may get any value (their probability is zero).
__device__ __forceinline__ void keep_mult4(float (&mk)[4], const uint8_t* __restrict__ keep, uint32_t seed_lo,
                                           uint32_t seed_hi, uint32_t thresh, size_t at, int key0, int S, float on) {
  if (keep != nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) mk[j] = (key0 + j < S && keep[at + j]) ? on : 0.f;
  } else {
    const uint32_t i0 = (uint32_t)at;
#pragma unroll
    for (int j = 0; j < 4; ++j) mk[j] = (mt_rng_hash(seed_lo, seed_hi, i0 + (uint32_t)j) >= thresh) ? on : 0.f;
  }
}

// Scores of one block of 16 queries -> probabilities (dropout applied), in place; writes the row statistic.  KIND and
// MASKED are compile-time so the 4 * NKB elements a lane owns are straight-line code: with run-time flags every element
// carried its own branches, and the vector ALU, not the matrix unit, set this kernel's time.
template <int NKB, int KIND, bool MASKED>
__device__ __forceinline__ void fwd_probs(f32x4 (&acc)[NKB], float* __restrict__ stat, const uint8_t* __restrict__ keep,
                                          uint32_t seed_lo, uint32_t seed_hi, uint32_t thresh, int S, int bh, int qb,
                                          int lane, float scale, float keep_scale) {
  const int g = lane >> 4, fr = lane & 15;
  const int query = qb * 16 + fr;
  float norm;
  if constexpr (KIND == 0) {
    float m = -3.0e38f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = kb * 16 + g * 4 + j;
        acc[kb][j] = (key < S) ? acc[kb][j] * scale : -3.0e38f;
        m = fmaxf(m, acc[kb][j]);
      }
    m = group_max(m);
    float l = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p = (kb * 16 + g * 4 + j < S) ? __expf(acc[kb][j] - m) : 0.f;
        acc[kb][j] = p;
        l += p;
      }
    l = group_sum(l);
    norm = 1.f / l;
    if (g == 0 && query < S) stat[(size_t)bh * S + query] = m + __logf(l);
  } else {
    float z = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = fmaxf(acc[kb][j] * scale, 0.f);     // padded keys have zero K rows -> a = 0
        acc[kb][j] = a;
        z += a;
      }
    z = group_sum(z) + AT_RELU_EPS;
    norm = 1.f / z;
    if (g == 0 && query < S) stat[(size_t)bh * S + query] = z;
  }
  const size_t row_at = ((size_t)bh * S + (query < S ? query : 0)) * S;      // a padding query reads row 0: in bounds, unused
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    if constexpr (MASKED) {
      float mk[4];
      const int key0 = kb * 16 + g * 4;
      keep_mult4(mk, keep, seed_lo, seed_hi, thresh, row_at + key0, key0, S, norm * keep_scale);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[kb][j] *= mk[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[kb][j] *= norm;
    }
  }
}

// NKB = blocks of 16 tokens (2 .. 8); PCHQ = 16-byte chunks per image row; F32: Q, K, V, O are f32
// (converted while staging) instead of bf16.  Operands are [B, S, H * d] with row strides ldq (Q and K), ldv, ldo, so
// the halves of a fused QK projection and a [B, S, d_model] value / output tensor are used in place; H = 1 with
// ld = d is the plain contiguous [BH, S, d] form.
template <int NKB, int PCHQ, bool F32>
__device__ __forceinline__ void attn_fwd_body(uint8_t* lds, const void* __restrict__ Q_, const void* __restrict__ K_,
                                              const void* __restrict__ V_, const uint8_t* __restrict__ keep,
                                              void* __restrict__ O_, float* __restrict__ stat, int S, int dk, int dv,
                                              float scale, float keep_scale, int kind, int H, long ldq, long ldv, long ldo,
                                              bool seeded, uint32_t seed_lo, uint32_t seed_hi, uint32_t thresh, int bh) {
  // attention dropout: a stored [BH, S, S] byte mask if `keep` is given, else drawn here from the seed words
  const bool masked = keep != nullptr || seeded;
  constexpr int NKP = at_padded_blocks(NKB);
  constexpr int SP = NKP * 16;
  constexpr int IMG = SP * PCHQ * 16;
  uint8_t *ldsQ = lds, *ldsK = lds + IMG, *ldsV = lds + 2 * IMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, fr = lane & 15;
  constexpr int ES = F32 ? 4 : 2;       // bytes per operand element
  const size_t qoff = at_base(bh, H, S, ldq, dk), voff = at_base(bh, H, S, ldv, dv), ooff = at_base(bh, H, S, ldo, dv);
  {
    const uint8_t* gq = reinterpret_cast<const uint8_t*>(Q_) + qoff * ES;
    const uint8_t* gk = reinterpret_cast<const uint8_t*>(K_) + qoff * ES;
    const uint8_t* gv = reinterpret_cast<const uint8_t*>(V_) + voff * ES;
    StageRegs<PCHQ, F32, SP> rq, rk, rv;
    stage_issue(rq, gq, S, dk, ldq, tid);
    stage_issue(rk, gk, S, dk, ldq, tid);
    if constexpr (stage_all_at_once<PCHQ, F32, SP>()) stage_issue(rv, gv, S, dv, ldv, tid);
    stage_commit(rq, ldsQ, gq, S, dk, ldq, tid);
    stage_commit(rk, ldsK, gk, S, dk, ldq, tid);
    if constexpr (!stage_all_at_once<PCHQ, F32, SP>()) stage_issue(rv, gv, S, dv, ldv, tid);
    stage_commit(rv, ldsV, gv, S, dv, ldv, tid);
  }
  __syncthreads();
  const int dsteps = (dk + 31) / 32, nvb = (dv + 15) / 16;

  for (int qb = wave; qb < NKB && qb * 16 < S; qb += 4) {     // blocks of padding queries produce nothing
    f32x4 acc[NKB];
    scores_T<NKB, PCHQ>(acc, ldsK, ldsQ, qb, dsteps, lane);
    if (kind == 0) {
      if (masked) fwd_probs<NKB, 0, true>(acc, stat, keep, seed_lo, seed_hi, thresh, S, bh, qb, lane, scale, keep_scale);
      else fwd_probs<NKB, 0, false>(acc, stat, keep, seed_lo, seed_hi, thresh, S, bh, qb, lane, scale, keep_scale);
    } else {
      if (masked) fwd_probs<NKB, 1, true>(acc, stat, keep, seed_lo, seed_hi, thresh, S, bh, qb, lane, scale, keep_scale);
      else fwd_probs<NKB, 1, false>(acc, stat, keep, seed_lo, seed_hi, thresh, S, bh, qb, lane, scale, keep_scale);
    }
    // O = P . V with P straight from the accumulators
    for (int nb = 0; nb < nvb; ++nb) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < NKP / 2; ++s2) {
        const bf16x8 pf = pack_frag(acc[2 * s2], (2 * s2 + 1 < NKB) ? acc[(2 * s2 + 1 < NKB) ? 2 * s2 + 1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f});
        const bf16x8 vf = frag_cols(ldsV, nb * 16, 32 * s2 + 4 * g, 32 * s2 + 16 + 4 * g, lane, PCHQ);
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o, 0, 0, 0);
      }
      const int n = nb * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int qrow = qb * 16 + g * 4 + j;
        if (qrow < S && n < dv) {
          const size_t at = ooff + (size_t)qrow * ldo + n;
          if constexpr (F32) reinterpret_cast<float*>(O_)[at] = o[j];
          else reinterpret_cast<uint16_t*>(O_)[at] = (uint16_t)f32_to_bf16_bits(o[j]);
        }
      }
    }
  }
}

template <int NKB, int PCHQ, bool F32>
__global__ __launch_bounds__(AT_THREADS) void mt_attn_fwd_kernel(const void* __restrict__ Q_, const void* __restrict__ K_,
                                                                 const void* __restrict__ V_,
                                                                 const uint8_t* __restrict__ keep,
                                                                 void* __restrict__ O_, float* __restrict__ stat,
                                                                 int S, int dk, int dv, float scale, float keep_scale,
                                                                 int kind, int H, long ldq, long ldv, long ldo,
                                                                 const int64_t* __restrict__ seed, uint32_t thresh) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[3 * at_padded_blocks(NKB) * 16 * PCHQ * 16];
  uint32_t seed_lo = 0u, seed_hi = 0u;
  if (seed != nullptr) { const uint64_t sv = (uint64_t)seed[0]; seed_lo = (uint32_t)sv; seed_hi = (uint32_t)(sv >> 32); }
  attn_fwd_body<NKB, PCHQ, F32>(lds, Q_, K_, V_, keep, O_, stat, S, dk, dv, scale, keep_scale, kind, H, ldq, ldv, ldo,
                                seed != nullptr, seed_lo, seed_hi, thresh, blockIdx.x);
}

// Backward, one block of 16 queries: scores p and dP^T dp (both [key][query] accumulators) -> probabilities and dS^T in
// dp; P^T (dropout applied) and dS^T go to LDS as bf16 for the two products that contract over queries.  KIND and
// MASKED are compile-time for the same reason as in fwd_probs.
template <int NKB, int PCH, int KIND, bool MASKED>
__device__ __forceinline__ void bwd_block(f32x4 (&p)[NKB], f32x4 (&dp)[NKB], uint8_t* ldsPT, uint8_t* ldsST,
                                          const float* __restrict__ stat, const uint8_t* __restrict__ keep,
                                          uint32_t seed_lo, uint32_t seed_hi, uint32_t thresh, int S, int bh, int qb,
                                          int lane, float scale, float keep_scale) {
  const int g = lane >> 4, fr = lane & 15;
  const int query = qb * 16 + fr;
  const float st = (query < S) ? stat[(size_t)bh * S + query] : 1.f;
  const float inv_z = (KIND == 1) ? 1.f / st : 0.f;
  // where key row 4g + j keeps this query's element; key block kb is kb * 16 rows further (16 rows leave the swizzle as is)
  int at_row[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) at_row[j] = img_off(g * 4 + j, query >> 3, PCH) + (query & 7) * 2;
  constexpr int KB_STEP = 16 * PCH * 16;
  float delta = 0.f;
  uint32_t relu_pos = 0u;       // bit (kb*4 + j): pre-activation > 0 (kind 1)
  const size_t row_at = ((size_t)bh * S + (query < S ? query : 0)) * S;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    float mk[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (MASKED) {
      const int key0 = kb * 16 + g * 4;
      keep_mult4(mk, keep, seed_lo, seed_hi, thresh, row_at + key0, key0, S, keep_scale);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = kb * 16 + g * 4 + j;
      const float s = p[kb][j] * scale;
      float pv;
      if constexpr (KIND == 0) pv = (key < S && query < S) ? __expf(s - st) : 0.f;
      else { pv = fmaxf(s, 0.f) * inv_z; if (s > 0.f) relu_pos |= 1u << (kb * 4 + j); }
      const float dpe = MASKED ? dp[kb][j] * mk[j] : dp[kb][j];      // gradient w.r.t. the un-dropped probability
      delta += dpe * pv;
      p[kb][j] = pv;
      dp[kb][j] = dpe;
      // P^T with dropout applied feeds dV
      *reinterpret_cast<uint16_t*>(ldsPT + at_row[j] + kb * KB_STEP) =
          (uint16_t)f32_to_bf16_bits(MASKED ? pv * mk[j] : pv);
    }
  }
  delta = group_sum(delta);
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float ds;
      if constexpr (KIND == 0) ds = p[kb][j] * (dp[kb][j] - delta) * scale;
      else ds = ((relu_pos >> (kb * 4 + j)) & 1u) ? (dp[kb][j] - delta) * inv_z * scale : 0.f;
      if (query >= S) ds = 0.f;
      dp[kb][j] = ds;
      *reinterpret_cast<uint16_t*>(ldsST + at_row[j] + kb * KB_STEP) = (uint16_t)f32_to_bf16_bits(ds);
    }
}

// A gradient element: f32, or (gbf: bf16 operands whose gradients are wanted in bf16 too) rounded to bf16 on the way out,
// which saves the caller a cast pass over the whole gradient tensor.  idx counts elements of either type.
__device__ __forceinline__ void at_store_grad(float* __restrict__ base, size_t idx, float v, bool gbf) {
  if (gbf) reinterpret_cast<uint16_t*>(base)[idx] = (uint16_t)f32_to_bf16_bits(v);
  else base[idx] = v;
}

template <int NKB, int PCHQ, bool F32>     // dQ, dK, dV are laid out like Q, K, V (same row strides); f32 unless gbf
__device__ __forceinline__ void attn_bwd_body(uint8_t* lds, const void* __restrict__ Q_, const void* __restrict__ K_,
                                              const void* __restrict__ V_, const void* __restrict__ dO_,
                                              const uint8_t* __restrict__ keep, const float* __restrict__ stat,
                                              float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV, int S,
                                              int dk, int dv, float scale, float keep_scale, int kind, int H, long ldq,
                                              long ldv, long ldo, bool seeded, uint32_t seed_lo, uint32_t seed_hi,
                                              uint32_t thresh, int bh, bool gbf = false) {
  const bool masked = keep != nullptr || seeded;
  constexpr int NKP = at_padded_blocks(NKB);
  constexpr int SP = NKP * 16;
  constexpr int PCH = (SP <= 64) ? 8 : 16;        // chunks per row of the [key][query] images
  constexpr int PT_BYTES = SP * PCH * 16;
  constexpr int IMG = SP * PCHQ * 16;
  uint8_t *ldsQ = lds, *ldsK = lds + IMG, *ldsV = lds + 2 * IMG, *ldsdO = lds + 3 * IMG;
  uint8_t *ldsPT = lds + 4 * IMG, *ldsST = ldsPT + PT_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, fr = lane & 15;
  constexpr int ES = F32 ? 4 : 2;
  const size_t qoff = at_base(bh, H, S, ldq, dk), voff = at_base(bh, H, S, ldv, dv), ooff = at_base(bh, H, S, ldo, dv);
  {
    const uint8_t* gq = reinterpret_cast<const uint8_t*>(Q_) + qoff * ES;
    const uint8_t* gk = reinterpret_cast<const uint8_t*>(K_) + qoff * ES;
    const uint8_t* gv = reinterpret_cast<const uint8_t*>(V_) + voff * ES;
    const uint8_t* go = reinterpret_cast<const uint8_t*>(dO_) + ooff * ES;
    StageRegs<PCHQ, F32, SP> rq, rk, rv, ro;
    constexpr bool ALL = stage_all_at_once<PCHQ, F32, SP>();
    stage_issue(rq, gq, S, dk, ldq, tid);
    stage_issue(rk, gk, S, dk, ldq, tid);
    if constexpr (ALL) { stage_issue(rv, gv, S, dv, ldv, tid); stage_issue(ro, go, S, dv, ldo, tid); }
    stage_commit(rq, ldsQ, gq, S, dk, ldq, tid);
    stage_commit(rk, ldsK, gk, S, dk, ldq, tid);
    if constexpr (!ALL) { stage_issue(rv, gv, S, dv, ldv, tid); stage_issue(ro, go, S, dv, ldo, tid); }
    stage_commit(rv, ldsV, gv, S, dv, ldv, tid);
    stage_commit(ro, ldsdO, go, S, dv, ldo, tid);
  }
  __syncthreads();
  const int dsteps = (dk + 31) / 32, vsteps = (dv + 31) / 32, nkd = (dk + 15) / 16, nvb = (dv + 15) / 16;

  // ---------------------------------------------------------------- query-parallel phase
  for (int qb = wave; qb < NKP; qb += 4) {
    if (qb * 16 >= S) {      // a block of padding queries: its 32 bytes of every P^T and dS^T row are zero, nothing else
      for (int idx = lane; idx < SP * 2; idx += 64) {
        const int off = img_off(idx >> 1, qb * 2 + (idx & 1), PCH);
        *reinterpret_cast<u32x4*>(ldsPT + off) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(ldsST + off) = u32x4{0u, 0u, 0u, 0u};
      }
      continue;
    }
    f32x4 p[NKB], dp[NKB];
    scores_T<NKB, PCHQ>(p, ldsK, ldsQ, qb, dsteps, lane);
    scores_T<NKB, PCHQ>(dp, ldsV, ldsdO, qb, vsteps, lane);     // dP^T[key][query] = V . dO^T, same layout
    const int query = qb * 16 + fr;
    if (kind == 0) {
      if (masked) bwd_block<NKB, PCH, 0, true>(p, dp, ldsPT, ldsST, stat, keep, seed_lo, seed_hi, thresh, S, bh, qb, lane, scale, keep_scale);
      else bwd_block<NKB, PCH, 0, false>(p, dp, ldsPT, ldsST, stat, keep, seed_lo, seed_hi, thresh, S, bh, qb, lane, scale, keep_scale);
    } else {
      if (masked) bwd_block<NKB, PCH, 1, true>(p, dp, ldsPT, ldsST, stat, keep, seed_lo, seed_hi, thresh, S, bh, qb, lane, scale, keep_scale);
      else bwd_block<NKB, PCH, 1, false>(p, dp, ldsPT, ldsST, stat, keep, seed_lo, seed_hi, thresh, S, bh, qb, lane, scale, keep_scale);
    }
    // dQ^T[d][query] = sum_key K[key][d] * dS[query][key]   (dS registers are the B operand)
    for (int db = 0; db < nkd; ++db) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < NKP / 2; ++s2) {
        const bf16x8 kf = frag_cols(ldsK, db * 16, 32 * s2 + 4 * g, 32 * s2 + 16 + 4 * g, lane, PCHQ);
        const bf16x8 sf = pack_frag(dp[2 * s2], (2 * s2 + 1 < NKB) ? dp[(2 * s2 + 1 < NKB) ? 2 * s2 + 1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f});
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, sf, a, 0, 0, 0);
      }
      // a[j] = dQ[query = qb*16 + fr][d = db*16 + 4g + j]
      if (query < S) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int d = db * 16 + g * 4 + j;
          if (d < dk) at_store_grad(dQ, qoff + (size_t)query * ldq + d, a[j], gbf);
        }
      }
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- key-parallel phase
  for (int kb = wave; kb < NKB && kb * 16 < S; kb += 4) {
    for (int nb = 0; nb < nvb; ++nb) {       // dV[key][n] = sum_query P^T[key][query] dO[query][n]
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < NKP / 2; ++s2) {
        const bf16x8 pf = frag_rows(ldsPT, kb * 16, s2, lane, PCH);
        const bf16x8 of = frag_cols(ldsdO, nb * 16, 32 * s2 + 8 * g, 32 * s2 + 8 * g + 4, lane, PCHQ);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, of, a, 0, 0, 0);
      }
      const int n = nb * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = kb * 16 + g * 4 + j;
        if (key < S && n < dv) at_store_grad(dV, voff + (size_t)key * ldv + n, a[j], gbf);
      }
    }
    for (int db = 0; db < nkd; ++db) {       // dK[key][d] = sum_query dS^T[key][query] Q[query][d]
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < NKP / 2; ++s2) {
        const bf16x8 sf = frag_rows(ldsST, kb * 16, s2, lane, PCH);
        const bf16x8 qf = frag_cols(ldsQ, db * 16, 32 * s2 + 8 * g, 32 * s2 + 8 * g + 4, lane, PCHQ);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, qf, a, 0, 0, 0);
      }
      const int d = db * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = kb * 16 + g * 4 + j;
        if (key < S && d < dk) at_store_grad(dK, qoff + (size_t)key * ldq + d, a[j], gbf);
      }
    }
  }
}

template <int NKB, int PCHQ>
constexpr int at_bwd_lds_bytes() {
  constexpr int SP = at_padded_blocks(NKB) * 16;
  return 4 * SP * PCHQ * 16 + 2 * SP * ((SP <= 64) ? 8 : 16) * 16;
}

template <int NKB, int PCHQ, bool F32>
__global__ __launch_bounds__(AT_THREADS) void mt_attn_bwd_kernel(const void* __restrict__ Q_, const void* __restrict__ K_,
                                                                 const void* __restrict__ V_, const void* __restrict__ dO_,
                                                                 const uint8_t* __restrict__ keep,
                                                                 const float* __restrict__ stat, float* __restrict__ dQ,
                                                                 float* __restrict__ dK, float* __restrict__ dV, int S,
                                                                 int dk, int dv, float scale, float keep_scale,
                                                                 int kind, int H, long ldq, long ldv, long ldo,
                                                                 const int64_t* __restrict__ seed, uint32_t thresh,
                                                                 int gbf) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[at_bwd_lds_bytes<NKB, PCHQ>()];
  uint32_t seed_lo = 0u, seed_hi = 0u;
  if (seed != nullptr) { const uint64_t sv = (uint64_t)seed[0]; seed_lo = (uint32_t)sv; seed_hi = (uint32_t)(sv >> 32); }
  attn_bwd_body<NKB, PCHQ, F32>(lds, Q_, K_, V_, dO_, keep, stat, dQ, dK, dV, S, dk, dv, scale, keep_scale, kind, H, ldq,
                                ldv, ldo, seed != nullptr, seed_lo, seed_hi, thresh, blockIdx.x, !F32 && gbf != 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Exact-f32 form for short sequences (S <= 32, heads of at most 64 elements; f32 operands): every product and sum is an
// f32 fused multiply-add in a fixed order, exp and log are the accurate library functions -- the arithmetic of an eager
// f32 attention, nothing rounded to bf16.  These problems are tiny (the reference's sample model attends over 10
// tokens with heads of 8-32 elements: a 10 x 10 score tile), so the vector ALUs do them in less time than the matrix
// path spends on its 32-row padded images.  Same interface, same saved row statistic, same dropout stream (element
// index in the [BH, S, S] mask) as the matrix form; one workgroup per (batch, head).
#define ATX_MAX_S 32
#define ATX_MAX_D 64
#define ATX_PD 65          // row pitch (floats) of the Q / K / V / dO images: odd, so a column walk hits every bank once
#define ATX_PS 33          // row pitch of the S x S images
#define ATX_FWD_LDS ((3 * ATX_MAX_S * ATX_PD + ATX_MAX_S * ATX_PS) * 4)
#define ATX_BWD_LDS ((4 * ATX_MAX_S * ATX_PD + 3 * ATX_MAX_S * ATX_PS + ATX_MAX_S) * 4)

// Two-phase staging of an f32 [rows][d] block (row stride ld) into LDS rows of `pitch` floats: ax_issue requests up to
// NV 16-byte pieces per thread, ax_commit writes them.  Issuing all operands of a workgroup before committing any puts
// their global-memory latencies on top of each other; a loop that loads and stores piece by piece pays them one after
// the other, and these kernels do little else.  Blocks whose rows cannot be read as aligned vectors are copied element
// by element in ax_commit.
template <int NV>
struct AxRegs { f32x4 v[NV]; bool vec; };

template <int NV>
__device__ __forceinline__ void ax_issue(AxRegs<NV>& b, const float* __restrict__ src, int rows, int d, long ld, int tid) {
  b.vec = ((d & 3) == 0) && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) &&
          rows * (d >> 2) <= NV * AT_THREADS;
  if (!b.vec) return;
  const int dq = d >> 2, n = rows * dq;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int q = tid + u * AT_THREADS;
    b.v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < n) {
      const int r = q / dq, c = q - r * dq;
      b.v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)r * ld + c * 4);
    }
  }
}

template <int NV>
__device__ __forceinline__ void ax_commit(const AxRegs<NV>& b, float* dst, int pitch, const float* __restrict__ src, int rows,
                                          int d, long ld, int tid) {
  if (!b.vec) {
    for (int q = tid; q < rows * d; q += AT_THREADS) {
      const int r = q / d, c = q - r * d;
      dst[r * pitch + c] = src[(size_t)r * ld + c];
    }
    return;
  }
  const int dq = d >> 2, n = rows * dq;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int q = tid + u * AT_THREADS;
    if (q < n) {
      const int r = q / dq, c = q - r * dq;
      float* o = dst + r * pitch + c * 4;
      o[0] = b.v[u][0]; o[1] = b.v[u][1]; o[2] = b.v[u][2]; o[3] = b.v[u][3];
    }
  }
}

__device__ __forceinline__ float atx_keep(const uint8_t* __restrict__ keep, bool seeded, uint32_t s0, uint32_t s1,
                                          uint32_t thresh, size_t at, float keep_scale) {
  if (keep != nullptr) return keep[at] ? keep_scale : 0.f;
  if (seeded) return mt_rng_hash(s0, s1, (uint32_t)at) >= thresh ? keep_scale : 0.f;
  return 1.f;
}

__device__ __forceinline__ void attn_fwd_exact_body(float* lds, const float* __restrict__ Q, const float* __restrict__ K,
                                                    const float* __restrict__ V, const uint8_t* __restrict__ keep,
                                                    float* __restrict__ O, float* __restrict__ stat, int S, int dk, int dv,
                                                    float scale, float keep_scale, int kind, int H, long ldq, long ldv,
                                                    long ldo, bool seeded, uint32_t s0, uint32_t s1, uint32_t thresh,
                                                    int bh) {
  float *q = lds, *k = q + ATX_MAX_S * ATX_PD, *v = k + ATX_MAX_S * ATX_PD, *sc = v + ATX_MAX_S * ATX_PD;
  const int tid = threadIdx.x;
  const size_t qoff = at_base(bh, H, S, ldq, dk), voff = at_base(bh, H, S, ldv, dv), ooff = at_base(bh, H, S, ldo, dv);
  {
    AxRegs<2> rq, rk, rv;        // 32 rows x 64 floats = 512 pieces, two per thread
    ax_issue(rq, Q + qoff, S, dk, ldq, tid);
    ax_issue(rk, K + qoff, S, dk, ldq, tid);
    ax_issue(rv, V + voff, S, dv, ldv, tid);
    ax_commit(rq, q, ATX_PD, Q + qoff, S, dk, ldq, tid);
    ax_commit(rk, k, ATX_PD, K + qoff, S, dk, ldq, tid);
    ax_commit(rv, v, ATX_PD, V + voff, S, dv, ldv, tid);
  }
  __syncthreads();
  for (int p = tid; p < S * S; p += AT_THREADS) {
    const int i = p / S, j = p - i * S;
    float acc = 0.f;
    for (int d = 0; d < dk; ++d) acc = fmaf(q[i * ATX_PD + d], k[j * ATX_PD + d], acc);
    sc[i * ATX_PS + j] = acc * scale;
  }
  __syncthreads();
  if (tid < S) {
    float* row = sc + tid * ATX_PS;
    float norm;
    if (kind == 0) {
      float m = row[0];
      for (int j = 1; j < S; ++j) m = fmaxf(m, row[j]);
      float l = 0.f;
      for (int j = 0; j < S; ++j) { const float e = expf(row[j] - m); row[j] = e; l += e; }
      norm = 1.f / l;
      stat[(size_t)bh * S + tid] = m + logf(l);
    } else {
      float z = 0.f;
      for (int j = 0; j < S; ++j) { const float a = fmaxf(row[j], 0.f); row[j] = a; z += a; }
      z += AT_RELU_EPS;
      norm = 1.f / z;
      stat[(size_t)bh * S + tid] = z;
    }
    const size_t at = ((size_t)bh * S + tid) * S;
    for (int j = 0; j < S; ++j) row[j] = row[j] * norm * atx_keep(keep, seeded, s0, s1, thresh, at + j, keep_scale);
  }
  __syncthreads();
  for (int p = tid; p < S * dv; p += AT_THREADS) {
    const int i = p / dv, n = p - i * dv;
    float acc = 0.f;
    for (int j = 0; j < S; ++j) acc = fmaf(sc[i * ATX_PS + j], v[j * ATX_PD + n], acc);
    O[ooff + (size_t)i * ldo + n] = acc;
  }
}

__device__ __forceinline__ void attn_bwd_exact_body(float* lds, const float* __restrict__ Q, const float* __restrict__ K,
                                                    const float* __restrict__ V, const float* __restrict__ dO,
                                                    const uint8_t* __restrict__ keep, const float* __restrict__ stat,
                                                    float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV,
                                                    int S, int dk, int dv, float scale, float keep_scale, int kind, int H,
                                                    long ldq, long ldv, long ldo, bool seeded, uint32_t s0, uint32_t s1,
                                                    uint32_t thresh, int bh) {
  float *q = lds, *k = q + ATX_MAX_S * ATX_PD, *v = k + ATX_MAX_S * ATX_PD, *go = v + ATX_MAX_S * ATX_PD;
  float *pr = go + ATX_MAX_S * ATX_PD, *ds = pr + ATX_MAX_S * ATX_PS, *pm = ds + ATX_MAX_S * ATX_PS;
  float* delta = pm + ATX_MAX_S * ATX_PS;
  const int tid = threadIdx.x;
  const size_t qoff = at_base(bh, H, S, ldq, dk), voff = at_base(bh, H, S, ldv, dv), ooff = at_base(bh, H, S, ldo, dv);
  {
    AxRegs<2> rq, rk, rv, ro;
    ax_issue(rq, Q + qoff, S, dk, ldq, tid);
    ax_issue(rk, K + qoff, S, dk, ldq, tid);
    ax_issue(rv, V + voff, S, dv, ldv, tid);
    ax_issue(ro, dO + ooff, S, dv, ldo, tid);
    ax_commit(rq, q, ATX_PD, Q + qoff, S, dk, ldq, tid);
    ax_commit(rk, k, ATX_PD, K + qoff, S, dk, ldq, tid);
    ax_commit(rv, v, ATX_PD, V + voff, S, dv, ldv, tid);
    ax_commit(ro, go, ATX_PD, dO + ooff, S, dv, ldo, tid);
  }
  __syncthreads();
  for (int p = tid; p < S * S; p += AT_THREADS) {       // probabilities again, and the gradient reaching them
    const int i = p / S, j = p - i * S;
    float s = 0.f, dp = 0.f;
    for (int d = 0; d < dk; ++d) s = fmaf(q[i * ATX_PD + d], k[j * ATX_PD + d], s);
    for (int n = 0; n < dv; ++n) dp = fmaf(go[i * ATX_PD + n], v[j * ATX_PD + n], dp);
    s *= scale;
    const float st = stat[(size_t)bh * S + i];
    const float pv = (kind == 0) ? expf(s - st) : fmaxf(s, 0.f) / st;
    const float mk = atx_keep(keep, seeded, s0, s1, thresh, ((size_t)bh * S + i) * S + j, keep_scale);
    pr[i * ATX_PS + j] = pv;
    ds[i * ATX_PS + j] = dp * mk;            // gradient w.r.t. the un-dropped probability
    pm[i * ATX_PS + j] = pv * mk;            // the dropped probability feeds dV
  }
  __syncthreads();
  if (tid < S) {
    float a = 0.f;
    for (int j = 0; j < S; ++j) a = fmaf(ds[tid * ATX_PS + j], pr[tid * ATX_PS + j], a);
    delta[tid] = a;
  }
  __syncthreads();
  for (int p = tid; p < S * S; p += AT_THREADS) {
    const int i = p / S, j = p - i * S;
    const float pv = pr[i * ATX_PS + j], g = ds[i * ATX_PS + j] - delta[i];
    float r;
    if (kind == 0) r = pv * g * scale;
    else r = (pv > 0.f) ? g * (1.f / stat[(size_t)bh * S + i]) * scale : 0.f;
    ds[i * ATX_PS + j] = r;
  }
  __syncthreads();
  for (int p = tid; p < S * dk; p += AT_THREADS) {       // dQ = dS . K and dK = dS^T . Q
    const int i = p / dk, d = p - i * dk;
    float a = 0.f, b = 0.f;
    for (int j = 0; j < S; ++j) {
      a = fmaf(ds[i * ATX_PS + j], k[j * ATX_PD + d], a);
      b = fmaf(ds[j * ATX_PS + i], q[j * ATX_PD + d], b);
    }
    dQ[qoff + (size_t)i * ldq + d] = a;
    dK[qoff + (size_t)i * ldq + d] = b;
  }
  for (int p = tid; p < S * dv; p += AT_THREADS) {       // dV = (dropped P)^T . dO
    const int j = p / dv, n = p - j * dv;
    float a = 0.f;
    for (int i = 0; i < S; ++i) a = fmaf(pm[i * ATX_PS + j], go[i * ATX_PD + n], a);
    dV[voff + (size_t)j * ldv + n] = a;
  }
}

__global__ __launch_bounds__(AT_THREADS) void mt_attn_fwd_exact_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                       const float* __restrict__ V,
                                                                       const uint8_t* __restrict__ keep,
                                                                       float* __restrict__ O, float* __restrict__ stat, int S,
                                                                       int dk, int dv, float scale, float keep_scale,
                                                                       int kind, int H, long ldq, long ldv, long ldo,
                                                                       const int64_t* __restrict__ seed, uint32_t thresh) {
  __shared__ __attribute__((aligned(16))) float lds[ATX_FWD_LDS / 4];
  uint32_t s0 = 0u, s1 = 0u;
  if (seed != nullptr) { const uint64_t sv = (uint64_t)seed[0]; s0 = (uint32_t)sv; s1 = (uint32_t)(sv >> 32); }
  attn_fwd_exact_body(lds, Q, K, V, keep, O, stat, S, dk, dv, scale, keep_scale, kind, H, ldq, ldv, ldo, seed != nullptr, s0,
                      s1, thresh, blockIdx.x);
}

__global__ __launch_bounds__(AT_THREADS) void mt_attn_bwd_exact_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                       const float* __restrict__ V,
                                                                       const float* __restrict__ dO,
                                                                       const uint8_t* __restrict__ keep,
                                                                       const float* __restrict__ stat, float* __restrict__ dQ,
                                                                       float* __restrict__ dK, float* __restrict__ dV, int S,
                                                                       int dk, int dv, float scale, float keep_scale,
                                                                       int kind, int H, long ldq, long ldv, long ldo,
                                                                       const int64_t* __restrict__ seed, uint32_t thresh) {
  __shared__ __attribute__((aligned(16))) float lds[ATX_BWD_LDS / 4];
  uint32_t s0 = 0u, s1 = 0u;
  if (seed != nullptr) { const uint64_t sv = (uint64_t)seed[0]; s0 = (uint32_t)sv; s1 = (uint32_t)(sv >> 32); }
  attn_bwd_exact_body(lds, Q, K, V, dO, keep, stat, dQ, dK, dV, S, dk, dv, scale, keep_scale, kind, H, ldq, ldv, ldo,
                      seed != nullptr, s0, s1, thresh, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// The same exact-f32 arithmetic for sequences of 33..128 tokens: K and V stay in LDS for the whole (batch, head), the
// queries are walked in blocks of ATL_QB rows (their Q / dO rows and S-wide score rows are the only per-block images),
// and in the backward pass every thread keeps the dK and dV elements it owns in registers across the blocks.  Row pitches follow the head width (d + 1 floats) and the sequence (S + 1), so the
// images of a narrow head take a fraction of the LDS; the launcher passes the byte count as dynamic LDS.
#define ATL_QB 32
#define ATL_MAX_S 128

static size_t atl_fwd_lds(int S, int dk, int dv) {
  return (size_t)(S * (dk + 1) + S * (dv + 1) + ATL_QB * (dk + 1) + ATL_QB * (S + 1)) * 4;
}
static size_t atl_bwd_lds(int S, int dk, int dv) {
  return (size_t)(S * (dk + 1) + S * (dv + 1) + ATL_QB * (dk + 1) + ATL_QB * (dv + 1) + 3 * ATL_QB * (S + 1)) * 4;
}

// Register blocking: a thread forms a 4 x 4 block of dot products at a time (queries 4 * ib + r, keys jl + 32 * c with
// jl = tid & 31, ib = tid >> 5), so one step of the contraction is 8 LDS reads for 16 multiply-adds instead of 32 for 16
// -- LDS issue, not the ALUs, bounds these kernels.  Lanes walk consecutive keys (odd row pitch: every bank once) and
// share their query rows (a broadcast).  Each dot product is still one fused-multiply-add chain in ascending index
// order.  NC = key groups of 32 in use (ceil(S / 32)).
template <int NC>
__device__ __forceinline__ void atl_dots(float (&acc)[4][NC], const float* __restrict__ a, const int (&ai)[4],
                                         const float* __restrict__ b, const int (&bj)[NC], int depth) {
  for (int d = 0; d < depth; ++d) {
    float av[4], bv[NC];
#pragma unroll
    for (int r = 0; r < 4; ++r) av[r] = a[ai[r] + d];
#pragma unroll
    for (int c = 0; c < NC; ++c) bv[c] = b[bj[c] + d];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[r][c] = fmaf(av[r], bv[c], acc[r][c]);
  }
}

// out[i][n] = sum_j w[i][j] * m[j][n] for this thread's queries 4 * ib + r and columns jl + 32 * c (c < 2): per j four
// broadcast reads of w and one or two reads of m for four or eight multiply-adds.
template <bool TWO>
__device__ __forceinline__ void atl_rows_times_n(float (&acc)[4][2], const float* __restrict__ w, const int (&wi)[4],
                                                 const float* __restrict__ m, int pitch, const int (&col)[2], int S) {
#pragma unroll 4
  for (int j = 0; j < S; ++j) {
    float wv[4], mv[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) wv[r] = w[wi[r] + j];
    mv[0] = m[j * pitch + col[0]];
    if constexpr (TWO) mv[1] = m[j * pitch + col[1]];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[r][0] = fmaf(wv[r], mv[0], acc[r][0]);
      if constexpr (TWO) acc[r][1] = fmaf(wv[r], mv[1], acc[r][1]);
    }
  }
}

__device__ __forceinline__ void atl_rows_times(float (&acc)[4][2], const float* __restrict__ w, const int (&wi)[4],
                                               const float* __restrict__ m, int pitch, const int (&col)[2], bool two,
                                               int S) {
  if (two) atl_rows_times_n<true>(acc, w, wi, m, pitch, col, S);
  else atl_rows_times_n<false>(acc, w, wi, m, pitch, col, S);
}

template <int NC>
__device__ __forceinline__ void atl_fwd_body(
    float* atl_lds, const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const uint8_t* __restrict__ keep, float* __restrict__ O, float* __restrict__ stat, int S, int dk, int dv, float scale,
    float keep_scale, int kind, int H, long ldq, long ldv, long ldo, bool seeded, uint32_t s0, uint32_t s1, uint32_t thresh,
    int bh) {
  const int pk = dk + 1, pv = dv + 1, ps = S + 1;
  float *k = atl_lds, *v = k + S * pk, *q = v + S * pv, *sc = q + ATL_QB * pk;
  const int tid = threadIdx.x;
  const int jl = tid & 31, ib = tid >> 5;
  const size_t qoff = at_base(bh, H, S, ldq, dk), voff = at_base(bh, H, S, ldv, dv), ooff = at_base(bh, H, S, ldo, dv);
  AxRegs<2> rq;                  // a block of 32 queries x 64 floats: two pieces per thread
  {
    AxRegs<8> rk, rv;            // 128 keys x 64 floats: eight pieces per thread
    ax_issue(rk, K + qoff, S, dk, ldq, tid);
    ax_issue(rv, V + voff, S, dv, ldv, tid);
    ax_issue(rq, Q + qoff, (S < ATL_QB) ? S : ATL_QB, dk, ldq, tid);
    ax_commit(rk, k, pk, K + qoff, S, dk, ldq, tid);
    ax_commit(rv, v, pv, V + voff, S, dv, ldv, tid);
  }
  int kj[NC], vcol[2];
#pragma unroll
  for (int c = 0; c < NC; ++c) kj[c] = ((jl + 32 * c < S) ? jl + 32 * c : S - 1) * pk;
#pragma unroll
  for (int c = 0; c < 2; ++c) vcol[c] = (jl + 32 * c < dv) ? jl + 32 * c : dv - 1;
  for (int q0 = 0; q0 < S; q0 += ATL_QB) {
    const int nq = (S - q0 < ATL_QB) ? S - q0 : ATL_QB;
    __syncthreads();                      // the previous block's images are free
    if (q0 > 0) ax_issue(rq, Q + qoff + (size_t)q0 * ldq, nq, dk, ldq, tid);     // the first block is already on its way
    ax_commit(rq, q, pk, Q + qoff + (size_t)q0 * ldq, nq, dk, ldq, tid);
    __syncthreads();
    int qi[4], si[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = (ib * 4 + r < nq) ? ib * 4 + r : nq - 1;
      qi[r] = i * pk;
      si[r] = i * ps;
    }
    if (ib * 4 < nq) {
      float acc[4][NC] = {};
      atl_dots<NC>(acc, q, qi, k, kj, dk);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c)
          if (ib * 4 + r < nq && jl + 32 * c < S) sc[(ib * 4 + r) * ps + jl + 32 * c] = acc[r][c] * scale;
    }
    __syncthreads();
    {   // row pass: eight lanes per query row, each every eighth key; maxima and sums meet in a fixed xor tree
      const int row = tid >> 3, sub = tid & 7;
      const bool live = row < nq;
      float* rp = sc + (live ? row : 0) * ps;
      const int qi_g = q0 + row;
      float norm;
      if (kind == 0) {
        float m = -3.0e38f;
        for (int j = sub; j < S; j += 8) m = fmaxf(m, rp[j]);
        m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
        float l = 0.f;
        for (int j = sub; j < S; j += 8) { const float e = expf(rp[j] - m); if (live) rp[j] = e; l += e; }
        l += __shfl_xor(l, 1, 64); l += __shfl_xor(l, 2, 64); l += __shfl_xor(l, 4, 64);
        norm = 1.f / l;
        if (live && sub == 0) stat[(size_t)bh * S + qi_g] = m + logf(l);
      } else {
        float z = 0.f;
        for (int j = sub; j < S; j += 8) { const float a = fmaxf(rp[j], 0.f); if (live) rp[j] = a; z += a; }
        z += __shfl_xor(z, 1, 64); z += __shfl_xor(z, 2, 64); z += __shfl_xor(z, 4, 64);
        z += AT_RELU_EPS;
        norm = 1.f / z;
        if (live && sub == 0) stat[(size_t)bh * S + qi_g] = z;
      }
      if (live) {
        const size_t at = ((size_t)bh * S + qi_g) * S;
        for (int j = sub; j < S; j += 8) rp[j] = rp[j] * norm * atx_keep(keep, seeded, s0, s1, thresh, at + j, keep_scale);
      }
    }
    __syncthreads();
    if (ib * 4 < nq && jl < dv) {
      float acc[4][2] = {};
      atl_rows_times(acc, sc, si, v, pv, vcol, jl + 32 < dv, S);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (ib * 4 + r < nq && jl + 32 * c < dv) O[ooff + (size_t)(q0 + ib * 4 + r) * ldo + jl + 32 * c] = acc[r][c];
    }
  }
}

template <int NC>
__global__ __launch_bounds__(AT_THREADS) void mt_attn_fwd_exact_tiled_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const uint8_t* __restrict__ keep,
    float* __restrict__ O, float* __restrict__ stat, int S, int dk, int dv, float scale, float keep_scale, int kind, int H,
    long ldq, long ldv, long ldo, const int64_t* __restrict__ seed, uint32_t thresh) {
  extern __shared__ __attribute__((aligned(16))) float atl_lds[];
  uint32_t s0 = 0u, s1 = 0u;
  if (seed != nullptr) { const uint64_t sv = (uint64_t)seed[0]; s0 = (uint32_t)sv; s1 = (uint32_t)(sv >> 32); }
  atl_fwd_body<NC>(atl_lds, Q, K, V, keep, O, stat, S, dk, dv, scale, keep_scale, kind, H, ldq, ldv, ldo, seed != nullptr, s0,
                   s1, thresh, blockIdx.x);
}

// One query block's share of dK = dS^T . Q and dV = (dropped P)^T . dO for the (key, column) elements a thread owns:
// per query NC + NC reads along the keys and NE + NE broadcast reads along the columns, all independent, then
// 2 * NC * NE multiply-adds.  NE (column groups of eight in use) is compile-time so the loop body has no branches;
// columns past the head are clamped reads whose sums are never stored.
template <int NC, int NE>
__device__ __forceinline__ void atl_dkdv(float (&acck)[NC][NE], float (&accv)[NC][NE], const float* __restrict__ ds,
                                         const float* __restrict__ pm, const float* __restrict__ q,
                                         const float* __restrict__ go, const int (&sj)[NC], const int (&qcol)[NE],
                                         const int (&gcol)[NE], int nq, int ps, int pk, int pv) {
  for (int i = 0; i < nq; ++i) {
    float dsv[NC], pmv[NC], qv[NE], gv[NE];
#pragma unroll
    for (int c = 0; c < NC; ++c) { dsv[c] = ds[i * ps + sj[c]]; pmv[c] = pm[i * ps + sj[c]]; }
#pragma unroll
    for (int e = 0; e < NE; ++e) { qv[e] = q[i * pk + qcol[e]]; gv[e] = go[i * pv + gcol[e]]; }
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        acck[c][e] = fmaf(dsv[c], qv[e], acck[c][e]);
        accv[c][e] = fmaf(pmv[c], gv[e], accv[c][e]);
      }
  }
}

template <int NC, int NE>     // NE: column groups of eight the widest head spans (2, 4 or 8)
__device__ __forceinline__ void atl_bwd_body(
    float* atl_lds, const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const uint8_t* __restrict__ keep, const float* __restrict__ stat, float* __restrict__ dQ,
    float* __restrict__ dK, float* __restrict__ dV, int S, int dk, int dv, float scale, float keep_scale, int kind, int H,
    long ldq, long ldv, long ldo, bool seeded, uint32_t s0, uint32_t s1, uint32_t thresh, int bh) {
  const int pk = dk + 1, pv = dv + 1, ps = S + 1;
  float *k = atl_lds, *v = k + S * pk, *q = v + S * pv, *go = q + ATL_QB * pk;
  float *pr = go + ATL_QB * pv, *ds = pr + ATL_QB * ps, *pm = ds + ATL_QB * ps;
  const int tid = threadIdx.x;
  const int jl = tid & 31, ib = tid >> 5;
  const size_t qoff = at_base(bh, H, S, ldq, dk), voff = at_base(bh, H, S, ldv, dv), ooff = at_base(bh, H, S, ldo, dv);
  AxRegs<2> rq, ro;
  {
    AxRegs<8> rk, rv;
    ax_issue(rk, K + qoff, S, dk, ldq, tid);
    ax_issue(rv, V + voff, S, dv, ldv, tid);
    ax_issue(rq, Q + qoff, (S < ATL_QB) ? S : ATL_QB, dk, ldq, tid);
    ax_issue(ro, dO + ooff, (S < ATL_QB) ? S : ATL_QB, dv, ldo, tid);
    ax_commit(rk, k, pk, K + qoff, S, dk, ldq, tid);
    ax_commit(rv, v, pv, V + voff, S, dv, ldv, tid);
  }
  // dK / dV: a thread owns keys jl + 32 * c (c < NC) x columns ib + 8 * e (e < NE) -- up to 128 keys x 64 columns over 256
  // threads -- and sums them over the query blocks in order, in registers
  float acck[NC][NE], accv[NC][NE];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < NE; ++e) { acck[c][e] = 0.f; accv[c][e] = 0.f; }
  int kj[NC], vj[NC], sj[NC], kcol[2];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int j = (jl + 32 * c < S) ? jl + 32 * c : S - 1;
    kj[c] = j * pk;
    vj[c] = j * pv;
    sj[c] = j;
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) kcol[c] = (jl + 32 * c < dk) ? jl + 32 * c : dk - 1;
  int qcol[NE], gcol[NE];       // the columns ib + 8 * e this thread owns of dK / dV, clamped into the head
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    qcol[e] = (ib + 8 * e < dk) ? ib + 8 * e : dk - 1;
    gcol[e] = (ib + 8 * e < dv) ? ib + 8 * e : dv - 1;
  }
  for (int q0 = 0; q0 < S; q0 += ATL_QB) {
    const int nq = (S - q0 < ATL_QB) ? S - q0 : ATL_QB;
    __syncthreads();
    // this block's rows were requested before the loop or during the previous block's dK / dV pass
    ax_commit(rq, q, pk, Q + qoff + (size_t)q0 * ldq, nq, dk, ldq, tid);
    ax_commit(ro, go, pv, dO + ooff + (size_t)q0 * ldo, nq, dv, ldo, tid);
    __syncthreads();
    int qi[4], gi[4], si[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = (ib * 4 + r < nq) ? ib * 4 + r : nq - 1;
      qi[r] = i * pk;
      gi[r] = i * pv;
      si[r] = i * ps;
    }
    if (ib * 4 < nq) {
      float st4[4];                // the rows' saved statistic, on its way while the dot products run
#pragma unroll
      for (int r = 0; r < 4; ++r) st4[r] = stat[(size_t)bh * S + q0 + ((ib * 4 + r < nq) ? ib * 4 + r : nq - 1)];
      float as[4][NC] = {}, ad[4][NC] = {};
      atl_dots<NC>(as, q, qi, k, kj, dk);
      atl_dots<NC>(ad, go, gi, v, vj, dv);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ib * 4 + r, qg = q0 + i;
        if (i < nq) {
          const float st = st4[r];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int j = jl + 32 * c;
            if (j < S) {
              const float sv = as[r][c] * scale;
              const float pval = (kind == 0) ? expf(sv - st) : fmaxf(sv, 0.f) / st;
              const float mk = atx_keep(keep, seeded, s0, s1, thresh, ((size_t)bh * S + qg) * S + j, keep_scale);
              pr[i * ps + j] = pval;
              ds[i * ps + j] = ad[r][c] * mk;
              pm[i * ps + j] = pval * mk;
            }
          }
        }
      }
    }
    __syncthreads();
    {   // row pass, eight lanes per query row: delta = sum_j dP * P (partial sums meet in a fixed xor tree), then dS in
        // place -- the lanes that read a row are the ones that overwrite it, so no barrier in between
      const int row = tid >> 3, sub = tid & 7;
      const bool live = row < nq;
      const int ro = (live ? row : 0) * ps;
      float a = 0.f;
      for (int j = sub; j < S; j += 8) a = fmaf(ds[ro + j], pr[ro + j], a);
      a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
      if (live) {
        const float inv = (kind == 0) ? 0.f : 1.f / stat[(size_t)bh * S + q0 + row];
        for (int j = sub; j < S; j += 8) {
          const float pval = pr[ro + j], g = ds[ro + j] - a;
          ds[ro + j] = (kind == 0) ? pval * g * scale : ((pval > 0.f) ? g * inv * scale : 0.f);
        }
      }
    }
    __syncthreads();
    if (ib * 4 < nq && jl < dk) {                            // dQ rows of this block
      float acc[4][2] = {};
      atl_rows_times(acc, ds, si, k, pk, kcol, jl + 32 < dk, S);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (ib * 4 + r < nq && jl + 32 * c < dk) dQ[qoff + (size_t)(q0 + ib * 4 + r) * ldq + jl + 32 * c] = acc[r][c];
    }
    // this block's share of dK = dS^T . Q and dV = (dropped P)^T . dO
    if (q0 + ATL_QB < S) {      // the next block's rows travel while this block's dK / dV are summed
      const int q1 = q0 + ATL_QB, n1 = (S - q1 < ATL_QB) ? S - q1 : ATL_QB;
      ax_issue(rq, Q + qoff + (size_t)q1 * ldq, n1, dk, ldq, tid);
      ax_issue(ro, dO + ooff + (size_t)q1 * ldo, n1, dv, ldo, tid);
    }
    atl_dkdv<NC, NE>(acck, accv, ds, pm, q, go, sj, qcol, gcol, nq, ps, pk, pv);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int j = jl + 32 * c;
    if (j < S) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int col = ib + 8 * e;
        if (col < dk) dK[qoff + (size_t)j * ldq + col] = acck[c][e];
        if (col < dv) dV[voff + (size_t)j * ldv + col] = accv[c][e];
      }
    }
  }
}

template <int NC, int NE>
__global__ __launch_bounds__(AT_THREADS) void mt_attn_bwd_exact_tiled_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ dO,
    const uint8_t* __restrict__ keep, const float* __restrict__ stat, float* __restrict__ dQ, float* __restrict__ dK,
    float* __restrict__ dV, int S, int dk, int dv, float scale, float keep_scale, int kind, int H, long ldq, long ldv,
    long ldo, const int64_t* __restrict__ seed, uint32_t thresh) {
  extern __shared__ __attribute__((aligned(16))) float atl_lds[];
  uint32_t s0 = 0u, s1 = 0u;
  if (seed != nullptr) { const uint64_t sv = (uint64_t)seed[0]; s0 = (uint32_t)sv; s1 = (uint32_t)(sv >> 32); }
  atl_bwd_body<NC, NE>(atl_lds, Q, K, V, dO, keep, stat, dQ, dK, dV, S, dk, dv, scale, keep_scale, kind, H, ldq, ldv, ldo,
                   seed != nullptr, s0, s1, thresh, blockIdx.x);
}

// Exactness level of f32 attention: 0 = f32 operands always go through the matrix form (rounded to bf16); 1 = sequences
// of up to 32 tokens run the exact-f32 form; 2 (default) = so do sequences of up to 128 tokens, tiled (heads of up to 64
// elements in both; wider heads keep the matrix form).  f32 models therefore multiply in f32 unless told otherwise
// (mt_attn_exact_level, driven by the package's ``attn_exact`` option).
static int at_level = 2;
static int at_exact_level() { return at_level; }
extern "C" int mt_attn_exact_level(int set) {      // set >= 0 replaces the level; returns the one in force before
  const int was = at_exact_level();
  if (set >= 0) at_level = set;
  return was;
}
// key groups of 32 a sequence of 33..128 tokens spans: the tiled kernels are compiled for 2, 3 and 4
#define ATL_NC(LAUNCH, S) do { if ((S) <= 64) LAUNCH(2); else if ((S) <= 96) LAUNCH(3); else LAUNCH(4); } while (0)
// ... and the backward kernels per number of 8-column groups of the widest head as well: 2, 4 or 8
#define ATL_NCNE(LAUNCH, S, D)                                                                          \
  do {                                                                                                  \
    if ((D) <= 16) { if ((S) <= 64) LAUNCH(2, 2); else if ((S) <= 96) LAUNCH(3, 2); else LAUNCH(4, 2); } \
    else if ((D) <= 32) { if ((S) <= 64) LAUNCH(2, 4); else if ((S) <= 96) LAUNCH(3, 4); else LAUNCH(4, 4); } \
    else { if ((S) <= 64) LAUNCH(2, 8); else if ((S) <= 96) LAUNCH(3, 8); else LAUNCH(4, 8); }          \
  } while (0)
static bool atl_fits(int S, int dk, int dv) {
  return at_exact_level() >= 2 && S > ATX_MAX_S && S <= ATL_MAX_S && dk <= ATX_MAX_D && dv <= ATX_MAX_D;
}
static int atl_prepare(const void* fn, size_t bytes) {      // kernels above 64 KiB of dynamic LDS have to be told so once
  if (bytes > 160u * 1024u) return -2;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return -9;
  return 0;
}

// Level 0 sends f32 operands of short sequences through the matrix form too (operands rounded to bf16).
static bool at_exact_on() { return at_exact_level() >= 1; }
static bool at_exact_fits(int S, int dk, int dv) { return S <= ATX_MAX_S && dk <= ATX_MAX_D && dv <= ATX_MAX_D; }

// ---------------------------------------------------------------------------------------------------------------
// Grouped form for packs of sequence-model trials: one launch covers the attention of one encoder layer of EVERY
// resident trial.  Workgroup (x, y) is (batch, head) problem x of trial slot ids[y]; each slot brings its own head
// count, query / key width, value width, attention kind (0 softmax, 1 relu-normalised), dropout rate and seed.
// Operands are read in place from the slot's projection output: q, k [B*S][..] with row stride ldq (head h at column
// h * dk), v with row stride ldv (head h at column h * dv) -- the three blocks of one fused projection, or v from a
// buffer of its own (values mixed by a local convolution) -- and the output o [B*S][H * dv] with row stride ldo.
struct MtAttnSlot {
  const float* q;
  const float* k;
  const float* v;
  float* o;
  float* stat;          // [B*H][S] row statistic saved for the backward pass
  const float* d_o;     // backward: gradient of o (row stride ldo)
  float* dq;            // backward: gradients, laid out like q, k, v
  float* dkk;
  float* dvv;
  const uint32_t* seed; // {lo, hi} dropout seed words of the slot's current epoch
  int32_t B, H, dk, dv;
  int32_t ldq, ldv, ldo, kind;
  int32_t active_steps, site;
  float scale, keep_scale;
  uint32_t thresh;      // p * 2^32, 0 = no attention dropout
  int32_t pad;
};

#define AT_SLOT_FWD_ARGS(a) (a).q, (a).k, (a).v, nullptr, (a).o, (a).stat, S, (a).dk, (a).dv, (a).scale
#define AT_SLOT_BWD_ARGS(a) (a).q, (a).k, (a).v, (a).d_o, nullptr, (a).stat, (a).dq, (a).dkk, (a).dvv, S, (a).dk, (a).dv, (a).scale
#define AT_SLOT_TAIL(a) (a).kind, (a).H, (long)(a).ldq, (long)(a).ldv, (long)(a).ldo, seeded, s0, s1, (a).thresh, blockIdx.x

template <int NKB, int PCHQ>
__global__ __launch_bounds__(AT_THREADS) void mt_attn_fwd_grouped_kernel(const MtAttnSlot* __restrict__ slots,
                                                                         const int32_t* __restrict__ ids, int S,
                                                                         int step_imm, int b_override, int train,
                                                                         const int* __restrict__ step_base) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[3 * at_padded_blocks(NKB) * 16 * PCHQ * 16];
  const int step = mt_step(step_imm, step_base);
  const MtAttnSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const int B = b_override > 0 ? b_override : a.B;
  if ((int)blockIdx.x >= B * a.H) return;
  const bool seeded = train && a.thresh != 0u && a.seed != nullptr;
  uint32_t s0 = 0u, s1 = 0u;
  if (seeded) mt_site_seed(a.seed, a.site, step, s0, s1);
  attn_fwd_body<NKB, PCHQ, true>(lds, AT_SLOT_FWD_ARGS(a), seeded ? a.keep_scale : 1.f, AT_SLOT_TAIL(a));
}

template <int NKB, int PCHQ>
__global__ __launch_bounds__(AT_THREADS) void mt_attn_bwd_grouped_kernel(const MtAttnSlot* __restrict__ slots,
                                                                         const int32_t* __restrict__ ids, int S,
                                                                         int step_imm, const int* __restrict__ step_base) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[at_bwd_lds_bytes<NKB, PCHQ>()];
  const int step = mt_step(step_imm, step_base);
  const MtAttnSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  if ((int)blockIdx.x >= a.B * a.H) return;
  const bool seeded = a.thresh != 0u && a.seed != nullptr;
  uint32_t s0 = 0u, s1 = 0u;
  if (seeded) mt_site_seed(a.seed, a.site, step, s0, s1);
  attn_bwd_body<NKB, PCHQ, true>(lds, AT_SLOT_BWD_ARGS(a), seeded ? a.keep_scale : 1.f, AT_SLOT_TAIL(a));
}

__global__ __launch_bounds__(AT_THREADS) void mt_attn_fwd_grouped_exact_kernel(const MtAttnSlot* __restrict__ slots,
                                                                               const int32_t* __restrict__ ids, int S,
                                                                               int step_imm, int b_override, int train,
                                                                               const int* __restrict__ step_base) {
  __shared__ __attribute__((aligned(16))) float lds[ATX_FWD_LDS / 4];
  const int step = mt_step(step_imm, step_base);
  const MtAttnSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const int B = b_override > 0 ? b_override : a.B;
  if ((int)blockIdx.x >= B * a.H) return;
  const bool seeded = train && a.thresh != 0u && a.seed != nullptr;
  uint32_t s0 = 0u, s1 = 0u;
  if (seeded) mt_site_seed(a.seed, a.site, step, s0, s1);
  attn_fwd_exact_body(lds, AT_SLOT_FWD_ARGS(a), seeded ? a.keep_scale : 1.f, AT_SLOT_TAIL(a));
}

__global__ __launch_bounds__(AT_THREADS) void mt_attn_bwd_grouped_exact_kernel(const MtAttnSlot* __restrict__ slots,
                                                                               const int32_t* __restrict__ ids, int S,
                                                                               int step_imm,
                                                                               const int* __restrict__ step_base) {
  __shared__ __attribute__((aligned(16))) float lds[ATX_BWD_LDS / 4];
  const int step = mt_step(step_imm, step_base);
  const MtAttnSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  if ((int)blockIdx.x >= a.B * a.H) return;
  const bool seeded = a.thresh != 0u && a.seed != nullptr;
  uint32_t s0 = 0u, s1 = 0u;
  if (seeded) mt_site_seed(a.seed, a.site, step, s0, s1);
  attn_bwd_exact_body(lds, AT_SLOT_BWD_ARGS(a), seeded ? a.keep_scale : 1.f, AT_SLOT_TAIL(a));
}

// Sequences of 33..128 tokens, exact f32: the tiled bodies, dynamic LDS sized by the launcher for the widest head.
template <int NC>
__global__ __launch_bounds__(AT_THREADS) void mt_attn_fwd_grouped_tiled_kernel(const MtAttnSlot* __restrict__ slots,
                                                                               const int32_t* __restrict__ ids, int S,
                                                                               int step_imm, int b_override, int train,
                                                                               const int* __restrict__ step_base) {
  extern __shared__ __attribute__((aligned(16))) float atl_lds[];
  const int step = mt_step(step_imm, step_base);
  const MtAttnSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const int B = b_override > 0 ? b_override : a.B;
  if ((int)blockIdx.x >= B * a.H) return;
  const bool seeded = train && a.thresh != 0u && a.seed != nullptr;
  uint32_t s0 = 0u, s1 = 0u;
  if (seeded) mt_site_seed(a.seed, a.site, step, s0, s1);
  atl_fwd_body<NC>(atl_lds, AT_SLOT_FWD_ARGS(a), seeded ? a.keep_scale : 1.f, AT_SLOT_TAIL(a));
}

template <int NC, int NE>
__global__ __launch_bounds__(AT_THREADS) void mt_attn_bwd_grouped_tiled_kernel(const MtAttnSlot* __restrict__ slots,
                                                                               const int32_t* __restrict__ ids, int S,
                                                                               int step_imm,
                                                                               const int* __restrict__ step_base) {
  extern __shared__ __attribute__((aligned(16))) float atl_lds[];
  const int step = mt_step(step_imm, step_base);
  const MtAttnSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  if ((int)blockIdx.x >= a.B * a.H) return;
  const bool seeded = a.thresh != 0u && a.seed != nullptr;
  uint32_t s0 = 0u, s1 = 0u;
  if (seeded) mt_site_seed(a.seed, a.site, step, s0, s1);
  // NE covers the widest head of the launch; a slot with narrower heads takes the body compiled for its own width
  const int wmax = a.dk > a.dv ? a.dk : a.dv;
#define ATL_SLOT_BWD(E) atl_bwd_body<NC, E>(atl_lds, AT_SLOT_BWD_ARGS(a), seeded ? a.keep_scale : 1.f, AT_SLOT_TAIL(a))
  if constexpr (NE >= 8) { if (wmax > 32) { ATL_SLOT_BWD(8); return; } }
  if constexpr (NE >= 4) { if (wmax > 16) { ATL_SLOT_BWD(4); return; } }
  ATL_SLOT_BWD(2);
#undef ATL_SLOT_BWD
}

extern "C" int mt_attn_slot_sizeof() { return (int)sizeof(MtAttnSlot); }

static int nkb_for(int S) { return ((S + 31) / 32) * 2; }   // grouped launches: the sequence padded to a multiple of 32
// one problem shape per launch: blocks of 16 tokens that hold any (3, 5 and 7 have kernels of their own: a sequence of
// 65 tokens -- 64 features and a summary token -- then does 5 x 5 blocks of score work instead of 6 x 6)
static int nkb_live_for(int S) { const int n = (S + 15) / 16; return n < 2 ? 2 : n; }

// Strided form.  Q, K: [B, S, H * dk] with row stride ldq elements (the two halves of a fused QK projection pass the
// same ldq and pointers dk * H apart); V: row stride ldv; O (and dO): row stride ldo.  f32 != 0: operands and output
// are f32, otherwise bf16.  All strides and head dims must keep 16-byte alignment of every 8-element chunk.
static int at_check(int BH, int S, int dk, int dv, int H, long ldq, long ldv, long ldo, int f32) {
  if (S < 1 || S > 128 || dk < 1 || dv < 1 || dk > 128 || dv > 128 || H < 1 || BH % H != 0) return -2;
  if ((dk > 64 || dv > 64) && S > 96) return -2;      // 128-wide images of a 128-token sequence exceed the LDS (backward)
  (void)f32;
  if (ldq < (long)H * dk || ldv < (long)H * dv || ldo < (long)H * dv) return -2;
  return 0;
}

extern "C" int mt_attn_fwd_strided_launch(const void* Q, const void* K, const void* V, const void* keep, void* O,
                                          float* stat, int BH, int S, int dk, int dv, float scale, float keep_scale,
                                          int kind, int H, long ldq, long ldv, long ldo, int f32, const int64_t* seed,
                                          float p, hipStream_t stream) {
  if (BH <= 0) return 0;
  const int rc = at_check(BH, S, dk, dv, H, ldq, ldv, ldo, f32);
  if (rc) return rc;
  const uint32_t thresh = (seed != nullptr && p > 0.f) ? (uint32_t)((double)p * 4294967296.0) : 0u;
  if (thresh == 0u) seed = nullptr;
  dim3 grid(BH), block(AT_THREADS);
  if (f32 && at_exact_on() && at_exact_fits(S, dk, dv)) {
    hipLaunchKernelGGL(mt_attn_fwd_exact_kernel, grid, block, 0, stream, (const float*)Q, (const float*)K, (const float*)V,
                       (const uint8_t*)keep, (float*)O, stat, S, dk, dv, scale, keep_scale, kind, H, ldq, ldv, ldo, seed,
                       thresh);
    return (int)hipGetLastError();
  }
  if (f32 && atl_fits(S, dk, dv)) {
    const size_t bytes = atl_fwd_lds(S, dk, dv);
#define ATL_F(NC)                                                                                                     \
  do {                                                                                                                \
    const int rc2 = atl_prepare((const void*)mt_attn_fwd_exact_tiled_kernel<NC>, bytes);                              \
    if (rc2) return rc2;                                                                                              \
    hipLaunchKernelGGL(mt_attn_fwd_exact_tiled_kernel<NC>, grid, block, bytes, stream, (const float*)Q, (const float*)K, \
                       (const float*)V, (const uint8_t*)keep, (float*)O, stat, S, dk, dv, scale, keep_scale, kind, H, ldq, \
                       ldv, ldo, seed, thresh);                                                                       \
  } while (0)
    ATL_NC(ATL_F, S);
#undef ATL_F
    return (int)hipGetLastError();
  }
  const bool narrow = dk <= 32 && dv <= 32;
#define AT_FWD2(N, P, F)                                                                                        \
  hipLaunchKernelGGL((mt_attn_fwd_kernel<N, P, F>), grid, block, 0, stream, Q, K, V, (const uint8_t*)keep, O, stat, S, dk, \
                     dv, scale, keep_scale, kind, H, ldq, ldv, ldo, seed, thresh)
#define AT_FWD1(N, P) do { if (f32) AT_FWD2(N, P, true); else AT_FWD2(N, P, false); } while (0)
  const bool wide = dk > 64 || dv > 64;       // heads of 65..128 elements: 256-byte image rows (S <= 96, checked above)
#define AT_FWD(N) do { if (narrow) AT_FWD1(N, 4); else if (!wide) AT_FWD1(N, 8); else AT_FWD1(N, 16); } while (0)
  switch (nkb_live_for(S)) {
    case 2: AT_FWD(2); break;
    case 3: AT_FWD(3); break;
    case 4: AT_FWD(4); break;
    case 5: AT_FWD(5); break;
    case 6: AT_FWD(6); break;
    case 7: if (narrow) AT_FWD1(7, 4); else AT_FWD1(7, 8); break;
    case 8: if (narrow) AT_FWD1(8, 4); else AT_FWD1(8, 8); break;
    default: return -3;
  }
#undef AT_FWD
#undef AT_FWD1
#undef AT_FWD2
  return (int)hipGetLastError();
}

extern "C" int mt_attn_bwd_strided_launch(const void* Q, const void* K, const void* V, const void* dO, const void* keep,
                                          const float* stat, float* dQ, float* dK, float* dV, int BH, int S, int dk,
                                          int dv, float scale, float keep_scale, int kind, int H, long ldq, long ldv,
                                          long ldo, int f32, const int64_t* seed, float p, hipStream_t stream) {
  if (BH <= 0) return 0;
  // f32: 1 = f32 operands and gradients; 0 = bf16 operands, f32 gradients; 2 = bf16 operands, bf16 gradients
  const int gbf = (f32 == 2) ? 1 : 0;
  if (gbf) f32 = 0;
  const int rc = at_check(BH, S, dk, dv, H, ldq, ldv, ldo, f32);
  if (rc) return rc;
  const uint32_t thresh = (seed != nullptr && p > 0.f) ? (uint32_t)((double)p * 4294967296.0) : 0u;
  if (thresh == 0u) seed = nullptr;
  dim3 grid(BH), block(AT_THREADS);
  if (f32 && at_exact_on() && at_exact_fits(S, dk, dv)) {
    hipLaunchKernelGGL(mt_attn_bwd_exact_kernel, grid, block, 0, stream, (const float*)Q, (const float*)K, (const float*)V,
                       (const float*)dO, (const uint8_t*)keep, stat, dQ, dK, dV, S, dk, dv, scale, keep_scale, kind, H, ldq,
                       ldv, ldo, seed, thresh);
    return (int)hipGetLastError();
  }
  if (f32 && atl_fits(S, dk, dv)) {
    const size_t bytes = atl_bwd_lds(S, dk, dv);
#define ATL_B(NC, NE)                                                                                                 \
  do {                                                                                                                \
    const int rc2 = atl_prepare((const void*)mt_attn_bwd_exact_tiled_kernel<NC, NE>, bytes);                          \
    if (rc2) return rc2;                                                                                              \
    hipLaunchKernelGGL((mt_attn_bwd_exact_tiled_kernel<NC, NE>), grid, block, bytes, stream, (const float*)Q, (const float*)K, \
                       (const float*)V, (const float*)dO, (const uint8_t*)keep, stat, dQ, dK, dV, S, dk, dv, scale,    \
                       keep_scale, kind, H, ldq, ldv, ldo, seed, thresh);                                             \
  } while (0)
    ATL_NCNE(ATL_B, S, (dk > dv) ? dk : dv);
#undef ATL_B
    return (int)hipGetLastError();
  }
  const bool narrow = dk <= 32 && dv <= 32;
#define AT_BWD2(N, P, F)                                                                                        \
  hipLaunchKernelGGL((mt_attn_bwd_kernel<N, P, F>), grid, block, 0, stream, Q, K, V, dO, (const uint8_t*)keep, stat, dQ, \
                     dK, dV, S, dk, dv, scale, keep_scale, kind, H, ldq, ldv, ldo, seed, thresh, gbf)
#define AT_BWD1(N, P) do { if (f32) AT_BWD2(N, P, true); else AT_BWD2(N, P, false); } while (0)
  const bool wide = dk > 64 || dv > 64;
#define AT_BWD(N) do { if (narrow) AT_BWD1(N, 4); else if (!wide) AT_BWD1(N, 8); else AT_BWD1(N, 16); } while (0)
  switch (nkb_live_for(S)) {
    case 2: AT_BWD(2); break;
    case 3: AT_BWD(3); break;
    case 4: AT_BWD(4); break;
    case 5: AT_BWD(5); break;
    case 6: AT_BWD(6); break;
    case 7: if (narrow) AT_BWD1(7, 4); else AT_BWD1(7, 8); break;
    case 8: if (narrow) AT_BWD1(8, 4); else AT_BWD1(8, 8); break;
    default: return -3;
  }
#undef AT_BWD
#undef AT_BWD1
#undef AT_BWD2
  return (int)hipGetLastError();
}

// Contiguous [BH, S, d] bf16 operands (one head per "batch" entry).
extern "C" int mt_attn_fwd_launch(const void* Q, const void* K, const void* V, const void* keep, void* O, float* stat,
                                  int BH, int S, int dk, int dv, float scale, float keep_scale, int kind,
                                  hipStream_t stream) {
  return mt_attn_fwd_strided_launch(Q, K, V, keep, O, stat, BH, S, dk, dv, scale, keep_scale, kind, 1, dk, dv, dv, 0,
                                    nullptr, 0.f, stream);
}

extern "C" int mt_attn_bwd_launch(const void* Q, const void* K, const void* V, const void* dO, const void* keep,
                                  const float* stat, float* dQ, float* dK, float* dV, int BH, int S, int dk, int dv,
                                  float scale, float keep_scale, int kind, hipStream_t stream) {
  return mt_attn_bwd_strided_launch(Q, K, V, dO, keep, stat, dQ, dK, dV, BH, S, dk, dv, scale, keep_scale, kind, 1, dk, dv,
                                    dv, 0, nullptr, 0.f, stream);
}

// max_bh: the largest B * H among the active slots; wide: the widest head among them in elements (1 is read as 64, 0 as
// 32: the flag this argument used to be).  Heads wider than 32 take the 64-element images of the matrix form.
static int at_group_dmax(int wide) { return wide > 1 ? (wide > 64 ? 64 : wide) : (wide ? 64 : 32); }
extern "C" int mt_attn_fwd_grouped_launch(const MtAttnSlot* slots, const int32_t* ids, int n_active, int max_bh, int S,
                                          int wide, int step, int b_override, int train, hipStream_t stream) {
  if (n_active <= 0 || max_bh <= 0) return 0;
  if (S < 1 || S > 128) return -2;
  dim3 grid(max_bh, n_active), block(AT_THREADS);
  if (at_exact_on() && S <= ATX_MAX_S) {      // the slots' operands are f32 and their heads at most 64 wide
    hipLaunchKernelGGL(mt_attn_fwd_grouped_exact_kernel, grid, block, 0, stream, slots, ids, S, step, b_override, train,
                       mt_step_base_tls);
    return (int)hipGetLastError();
  }
  if (at_exact_level() >= 2) {
    const int dmax = at_group_dmax(wide);
    const size_t bytes = atl_fwd_lds(S, dmax, dmax);
#define ATL_GF(NC)                                                                                                    \
  do {                                                                                                                \
    const int rc2 = atl_prepare((const void*)mt_attn_fwd_grouped_tiled_kernel<NC>, bytes);                            \
    if (rc2) return rc2;                                                                                              \
    hipLaunchKernelGGL(mt_attn_fwd_grouped_tiled_kernel<NC>, grid, block, bytes, stream, slots, ids, S, step, b_override, \
                       train, mt_step_base_tls);                                                                      \
  } while (0)
    ATL_NC(ATL_GF, S);
#undef ATL_GF
    return (int)hipGetLastError();
  }
#define AT_GF(N, P) hipLaunchKernelGGL((mt_attn_fwd_grouped_kernel<N, P>), grid, block, 0, stream, slots, ids, S, step, \
                                       b_override, train, mt_step_base_tls)
#define AT_GFW(N) do { if (at_group_dmax(wide) > 32) AT_GF(N, 8); else AT_GF(N, 4); } while (0)
  switch (nkb_for(S)) {
    case 2: AT_GFW(2); break;
    case 4: AT_GFW(4); break;
    case 6: AT_GFW(6); break;
    case 8: AT_GFW(8); break;
    default: return -3;
  }
#undef AT_GFW
#undef AT_GF
  return (int)hipGetLastError();
}

extern "C" int mt_attn_bwd_grouped_launch(const MtAttnSlot* slots, const int32_t* ids, int n_active, int max_bh, int S,
                                          int wide, int step, hipStream_t stream) {
  if (n_active <= 0 || max_bh <= 0) return 0;
  if (S < 1 || S > 128) return -2;
  dim3 grid(max_bh, n_active), block(AT_THREADS);
  if (at_exact_on() && S <= ATX_MAX_S) {
    hipLaunchKernelGGL(mt_attn_bwd_grouped_exact_kernel, grid, block, 0, stream, slots, ids, S, step, mt_step_base_tls);
    return (int)hipGetLastError();
  }
  if (at_exact_level() >= 2) {
    const int dmax = at_group_dmax(wide);
    const size_t bytes = atl_bwd_lds(S, dmax, dmax);
#define ATL_GB(NC, NE)                                                                                                \
  do {                                                                                                                \
    const int rc2 = atl_prepare((const void*)mt_attn_bwd_grouped_tiled_kernel<NC, NE>, bytes);                        \
    if (rc2) return rc2;                                                                                              \
    hipLaunchKernelGGL((mt_attn_bwd_grouped_tiled_kernel<NC, NE>), grid, block, bytes, stream, slots, ids, S, step,   \
                       mt_step_base_tls);                                                                             \
  } while (0)
    ATL_NCNE(ATL_GB, S, dmax);
#undef ATL_GB
    return (int)hipGetLastError();
  }
#define AT_GB(N, P) hipLaunchKernelGGL((mt_attn_bwd_grouped_kernel<N, P>), grid, block, 0, stream, slots, ids, S, step, \
                                       mt_step_base_tls)
#define AT_GBW(N) do { if (at_group_dmax(wide) > 32) AT_GB(N, 8); else AT_GB(N, 4); } while (0)
  switch (nkb_for(S)) {
    case 2: AT_GBW(2); break;
    case 4: AT_GBW(4); break;
    case 6: AT_GBW(6); break;
    case 8: AT_GBW(8); break;
    default: return -3;
  }
#undef AT_GBW
#undef AT_GB
  return (int)hipGetLastError();
}
