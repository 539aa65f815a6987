// Shared declarations for the mitune gfx950 kernels (device helpers + host-visible descriptors).
//
// Everything here is written for CDNA4 only: 64-lane wavefronts, MFMA 16x16x32 bf16, LDS tiles
// read with ds_read_b128 / ds_read_b64_tr_b16.  The descriptor structs are mirrored field by field
// as ctypes.Structure classes in ops/lib.py; keep the two in sync.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MT_WAVE 64
#define MT_MAX_LAYERS 8      // hidden layers of an MLP trial (plan arrays of the step drivers)

// Optional device-resident step base.  Launchers forward this thread's pointer to every step-aware kernel, which
// adds *step_base to its immediate `step` argument.  The step driver sets it while it captures a hipGraph, so that
// a captured block of steps can be replayed at any position of an epoch; everywhere else it is null.
extern thread_local const int* mt_step_base_tls;
__device__ __forceinline__ int mt_step(int step, const int* __restrict__ step_base) {
  return step_base != nullptr ? step + *step_base : step;
}

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// Round-to-nearest-even f32 -> bf16 through the compiler's cast (v_cvt_pk_bf16_f32 on gfx950),
// which keeps NaN a NaN; the integer-add trick can turn some NaNs into zero or infinity.
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;
  return (uint32_t)__builtin_bit_cast(uint16_t, h);
}
// Two floats -> one dword of two bf16 (round to nearest even): the vector conversion is ONE v_cvt_pk_bf16_f32; converting
// the halves separately and or-ing them costs four instructions.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef float mt_f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 mt_bf16x2 __attribute__((ext_vector_type(2)));
  const mt_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mt_bf16x2));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------
// Grouped GEMM descriptors.  One group = one trial slot's GEMM for one layer and direction:
//   C[I x J] = sum_kk A(i, kk) * B(kk, j)
// mode 0 "NT" (forward):   A is [I][KK] (kk contiguous), B is [J][KK] (kk contiguous)
// mode 1 "NN" (dgrad):     A is [I][KK] (kk contiguous), B is [KK][J] (j contiguous)
// mode 2 "TN" (wgrad):     A is [KK][I] (i contiguous),  B is [KK][J] (j contiguous)
// A, B are bf16; C is bf16 for modes 0/1 and f32 for mode 2.
struct MtGemmGroup {
  const void* A;
  const void* B;
  void* C;
  const void* aux;     // mode 0: bias f32[J]; mode 1: ReLU source [I][ldaux] (output masked where <= 0), or with
                       // MT_GROUP_ADD the addend [I][ldaux]
  void* aux_out;       // mode 2: bias gradient f32[I] (column sums of A), written by the tj == 0 tiles
  const uint32_t* seed;  // f32 kernels, mode 0, thresh != 0: {lo, hi} words of the slot's dropout seed for this epoch
  int32_t I, J, KK;
  int32_t lda, ldb, ldc, ldaux;
  int32_t active_steps;  // the group takes part while step < active_steps
  // Per-group extras of the f32 kernels (zero = off; the bf16 kernels ignore them): sequence-model packs put dense
  // layers with different epilogues and dropout rates into one launch.
  uint32_t thresh;       // mode 0: drop an output element when its hash is below this (p * 2^32), after the ReLU
  float scale;           // mode 0: factor of the kept elements, 1 / (1 - p); mode 1: factor applied with the mask (0 = 1)
  int32_t gflags;        // MT_GROUP_* bits, OR-ed to the launch's flags
  int32_t site;          // which dropout site of the model this layer is (decorrelates the sites of one slot)
};

#define MT_GROUP_ACT 1    // mode 0: ReLU after bias; mode 1: multiply by (aux > 0)
#define MT_GROUP_ADD 2    // mode 1: output += aux (the gradient arriving over a residual connection); excludes the mask

struct MtTile {
  int32_t group;
  int16_t ti, tj;
};

#define MT_GEMM_FLAG_RELU 1   // mode 0: ReLU after bias
#define MT_GEMM_FLAG_MASK 1   // mode 1: multiply by (aux > 0)
#define MT_GEMM_FLAG_DBIAS 1  // mode 2: also produce aux_out = column sums of A
#define MT_GEMM_FLAG_BF16_PRODUCTS 0x100   // grouped f32 launches: round the operands to bf16 for the matrix instruction

// ------------------------------------------------------------------------------------------------
struct MtGatherSlot {
  void* X0;            // out: bf16 [B][Fpad]
  float* y;            // out: f32 [B]
  uint32_t key[4];     // Feistel round keys of (seed, trial, epoch)
  int32_t B;
  int32_t active_steps;
};

struct MtHeadSlot {
  const void* H;       // bf16 [B][ldh] last hidden activations (post-ReLU)
  void* dH;            // bf16 [B][ldh] out: dLoss/dZ of the last hidden layer (ReLU mask applied)
  const float* w;      // f32 [W] head weight
  const float* b;      // f32 [1] head bias
  float* gw;           // f32 [W] out
  float* gb;           // f32 [1] out
  const float* y;      // f32 [B] targets
  float* loss_acc;     // f32 [1] += mean batch loss
  double* metrics;     // eval: f64 [3] += {sum loss, sum squared error, sum abs pct error}
  int32_t B, W, ldh;
  int32_t loss_type;   // 0 mse, 1 mae, 2 huber(beta=1), 3 mape
  int32_t active_steps;
  int32_t pad;
};

struct MtOptSlot {
  float lr, beta1, beta2, eps, wd, momentum, clip;
  int32_t opt_type;      // 0 adam, 1 adamw, 2 sgd, 3 rmsprop
  int32_t warmup, total; // LR schedule: linear warm-up then linear decay (0 = off)
  int32_t n_params;      // live elements of this slot's slab
  int32_t active_steps;
  int32_t step_base;     // optimiser steps taken before this epoch
  int32_t first_end;     // elements [0, first_end) are layer 0's weights and bias (a multiple of 8) ...
  int32_t first_splits;  // ... whose gradient is spread over this many slabs (0: same as the rest of the slot)
  int32_t pad;
};

// Feistel permutation shared with packed/permutation.py (bit-identical).
__device__ __forceinline__ uint32_t mt_mix32(uint32_t x) {
  x = (x ^ (x >> 16)) * 0x85EBCA6Bu;
  x = (x ^ (x >> 13)) * 0xC2B2AE35u;
  return x ^ (x >> 16);
}

// Counter-based dropout: element i of a tensor is kept iff mt_rng_hash(seed, i) >= p * 2^32.  The 64-bit seed lives in
// device memory (ops/rng.py), so captured graphs draw fresh masks per replay and a backward pass recomputes the
// forward's mask instead of reading a stored one.
// One mixing round per element: 32-bit integer multiplies issue at a quarter of the vector rate, and in the attention
// kernels (one draw per score, forward and again in backward) a second round was most of the kernel's vector work.
__device__ __forceinline__ uint32_t mt_rng_hash(uint32_t s0, uint32_t s1, uint32_t i) {
  return mt_mix32((i * 0x9E3779B9u + s0) ^ s1);
}
// Keep flags (one byte each, 0 or 1) of elements i .. i+3, from a stored mask if there is one, else from the seed.
// Seed words of one dropout site at one step: derived from the slot's epoch seed so that forward and backward of the
// same step agree and no two sites or steps share a stream.
__device__ __forceinline__ void mt_site_seed(const uint32_t* __restrict__ seed, int site, int step, uint32_t& s0, uint32_t& s1) {
  s0 = seed[0] + (uint32_t)site * 0x9E3779B9u;
  s1 = seed[1] ^ mt_mix32((uint32_t)step * 0x85EBCA6Bu + (uint32_t)site + 0x51ED270Bu);
}
__device__ __forceinline__ uint32_t mt_keep_word4s(uint32_t s0, uint32_t s1, uint32_t thresh, uint32_t i) {
  uint32_t w = 0u;
#pragma unroll
  for (int e = 0; e < 4; ++e) w |= (mt_rng_hash(s0, s1, i + e) >= thresh ? 1u : 0u) << (8 * e);
  return w;
}
__device__ __forceinline__ uint32_t mt_keep_word4(const uint8_t* __restrict__ keep, const int64_t* __restrict__ seed,
                                                  uint32_t thresh, size_t i) {
  if (keep != nullptr) return *reinterpret_cast<const uint32_t*>(keep + i);
  if (seed == nullptr) return 0x01010101u;
  const uint64_t s = (uint64_t)seed[0];
  const uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
  uint32_t w = 0u;
#pragma unroll
  for (int e = 0; e < 4; ++e) w |= (mt_rng_hash(s0, s1, (uint32_t)(i + e)) >= thresh ? 1u : 0u) << (8 * e);
  return w;
}
__device__ __forceinline__ uint32_t mt_feistel(uint32_t x, const uint32_t* key, int hb) {
  const uint32_t mask = (1u << hb) - 1u;
  uint32_t left = (x >> hb) & mask, right = x & mask;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    uint32_t f = mt_mix32(right ^ key[r]) & mask;
    uint32_t nl = right;
    right = left ^ f;
    left = nl;
  }
  return (left << hb) | right;
}
__device__ __forceinline__ uint32_t mt_permute(uint32_t pos, uint32_t n, const uint32_t* key, int hb) {
  uint32_t x = mt_feistel(pos, key, hb);
  while (x >= n) x = mt_feistel(x, key, hb);   // cycle walking: ends because mt_feistel is a bijection
  return x;
}
