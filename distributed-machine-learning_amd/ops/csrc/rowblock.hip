// Row-block training kernel for packs of exact-f32 MLP trials (gfx950).
//
// Batch rows are independent through the forward pass and through the chain of input gradients; only the weight
// gradient sums over rows.  So one workgroup takes 16 (or 32) batch rows of ONE trial through
//     input rows -> every forward layer (+bias, ReLU) -> head + loss + dLoss/dH -> every input-gradient layer (ReLU mask)
// with the activations of its rows staying in LDS from the first layer to the last: no workgroup ever waits for
// another, and a training step of a pack is three launches (this one, the weight gradients of all layers, the
// optimiser) instead of 2 L + 2 chip-wide dependent ones.  What the weight-gradient launch needs is written to memory on
// the way: H_l for l < L - 1, dZ_l for every l, and the head's partial records {gw, gb, loss} per 16 rows.
// This is the step of /root/reference/ray-tune-hpo-regression-sample.py:110-120 for one mini-batch slice.
//
// Numbers: bit for bit those of the tile kernels (mt_gemm_f32_tile.h, mt_head_body.h).  Every product runs on
// v_mfma_f32_16x16x4_f32 with the same contraction order per output element -- MFMA number (ks, e) of a 32-deep K-step
// contracts kk = 16 ks + 4 g + e in lane group g -- bias and ReLU are applied to the finished sum, the mask of the input
// gradient is (forward output > 0), and the head walks its rows in the order of mt_head_loss_body.  A pack may therefore
// switch between the two forms from one epoch to the next (the engine picks by pack size) without a trial noticing.
//
// Work split inside the workgroup (512 threads = 8 waves): a layer's output columns are cut into tiles of 16; wave w owns
// tiles w, w + 8, w + 16, w + 24 (layers of up to 512 columns).  The activations [rows][K] are one LDS image that every
// wave reads (ds_read_b128: four consecutive kk of a row = the B operands of four MFMAs).  A tile's weights are read by
// nobody else in the workgroup; each wave streams its own through a private LDS ring of 1 KiB fragments (one 16-column
// tile x 16 kk), filled by LDS-DMA (global_load_lds, 16 bytes a lane) as many K-steps ahead as the ring holds for the
// wave's tile count -- 3 K-steps for 4 tiles, 12 for one -- and across layer boundaries too: the next layer's first
// fragments are requested before the current layer's epilogue.  The ring is private, so no barrier guards it: a wave
// waits for its own DMA with a counted s_waitcnt vmcnt and reads.  (A register ring would do the same job, but the
// compiler copies its registers wherever two code paths meet and waits for the loads in order to do so.)  Outputs are
// written back into the same LDS image once every wave has finished reading it (two barriers per layer, the only
// synchronisation there is).
#include "mt_head_body.h"
#include "mt_rowblock.h"

// Geometry per row-tile count MT.  16 rows: sixteen waves of at most two tiles each -- four waves per SIMD cover each
// other's DMA issue and LDS round trips better than two (critical workgroup of the 16-survivor pack 184 k -> 163 k
// clocks); 32 rows: eight waves of at most four tiles (the MFMA work per fragment doubles, and so do the registers) --
// measured no faster on any pack; the launcher builds the 16-row form only.
template <int MT> struct RbCfg {
  static constexpr int NW = MT == 1 ? 16 : 8;                  // waves per workgroup
  static constexpr int THREADS = NW * 64;
  static constexpr int TPW = 512 / 16 / NW;                    // column tiles per wave (RB_MAXW / 16 / NW)
  static constexpr int CAP = MT == 1 ? 6 : 8;                  // fragments per wave ring
  static constexpr int MASK_LAYERS = MT == 1 ? MT_MAX_LAYERS : 4;
};
#define RB_MAXW 512                              // widest layer / input row the LDS image holds
#define RB_PITCH (RB_MAXW * 4 + 128)             // bytes per image row; = 128 mod 256, so rows alternate bank halves
#define RB_MASK_BYTES (MT_MAX_LAYERS * (RB_MAXW / 16) * 64)
#define RB_FRAG 1024                             // bytes of one weight fragment: 16 columns x 16 kk of f32

// Pointers read from descriptor tables are generic to the compiler, and loads through them would be flat_load: counted
// on both wait counters and never left in flight past the next LDS access.  Everything below goes through these
// global-address-space types instead.
typedef const __attribute__((address_space(1))) float* rb_gcf;
typedef __attribute__((address_space(1))) float* rb_gf;
typedef const __attribute__((address_space(1))) f32x4* rb_gc4;
typedef __attribute__((address_space(1))) f32x4* rb_g4;

// Byte offset of the 16-byte chunk holding elements k .. k+3 (k % 4 == 0) of image row `row`.
__device__ __forceinline__ int rb_act_off(int row, int k) {
  return row * RB_PITCH + ((k >> 5) << 7) + ((((k >> 2) & 7) ^ (row & 7)) << 4);
}

typedef const __attribute__((address_space(1))) char* rb_gcc;

// A wave-uniform 64-bit value, told to the compiler as such: the weight bases below then live in scalar registers and
// the fragment loads take the form global_load v, v_offset, s[base] -- one 32-bit vector offset per load instead of a
// 64-bit vector address computed by the vector ALU for each.
__device__ __forceinline__ uint64_t rb_uniform64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

struct RbPhase {
  rb_gcc W;              // forward: [N][ldw] (kk contiguous); input gradient: [K][ldw] (output column contiguous)
  rb_gcf bias;           // forward only
  rb_gf out;             // memory copy of the output rows, [B][ldo] (null: not needed)
  int ldw, ldo, N, K, ns, nn, tv, layer;
  int tbase;             // first 16-column tile of this workgroup's share of the layer (0 unless the block is split)
  int depth;             // K-steps (of 16) the wave's ring holds for this layer's tile count
  uint32_t lane_off;     // byte offset of this lane's 16-byte piece inside a fragment's source block
};

// Every layer width and the input row length are multiples of 32 (the caller checks): a K-step is always whole and a
// layer has an even number of 16-column tiles.  CAP: fragments of the wave's ring.
template <int NW, int TPW, int CAP>
__device__ __forceinline__ void rb_make_phase(RbPhase& ph, const MtGemmGroup& g, int nn, int layer, int wave, int lane,
                                              int part, int nparts) {
  ph.W = (rb_gcc)rb_uniform64((uint64_t)g.B);
  ph.bias = nn ? nullptr : (rb_gcf)rb_uniform64((uint64_t)g.aux);
  ph.out = (rb_gf)rb_uniform64((uint64_t)g.C);
  ph.ldw = __builtin_amdgcn_readfirstlane(g.ldb);
  ph.ldo = __builtin_amdgcn_readfirstlane(g.ldc);
  ph.N = __builtin_amdgcn_readfirstlane(g.J);
  ph.K = __builtin_amdgcn_readfirstlane(g.KK);
  ph.ns = ph.K >> 4;
  ph.nn = nn;
  ph.layer = layer;
  const int ntp = (ph.N >> 4) / nparts;              // tiles of this workgroup: [tbase, tbase + ntp)
  ph.tbase = part * ntp;
  int tv = (ntp - wave + NW - 1) / NW;
  ph.tv = tv < 0 ? 0 : (tv > TPW ? TPW : tv);
  ph.depth = ph.tv > 0 ? CAP / ph.tv : 0;
  // Which 16 bytes of a fragment's source block a lane fetches (they land at fragment byte 16 * lane).  In both forms
  // four consecutive lanes fetch the four 16-byte pieces of one 64-byte row segment: the memory pipeline handles a
  // wave's request quad by quad, and a quad that touches four different rows costs four times the address cycles (this
  // alone took the forward K-loops from 2030 to about 1300 cycles per K-step of 512 columns).
  // Forward: rows are W[n0 + r][k0 .. k0 + 15], r = lane >> 2; the lane takes piece (lane & 3) ^ f(r >> 2) of the row, f =
  // {0, 3, 2, 1}, so that MFMA lane (i, g), which needs piece g of row i, finds it at position 4 i + (g ^ f(i >> 2)) and
  // the sixteen lanes that share an LDS cycle of ds_read_b128 hit sixteen different bank groups.
  // Input gradient: rows are W[k0 + r][j0 .. j0 + 15] with r = 4 (rr & 3) + (rr >> 2), rr = lane >> 2: position rr * 4 + c
  // then holds row 4 g + e at rr = 4 e + g, which spreads the four ds_read_b32 of a lane group over all banks.
  {
    const int rr = lane >> 2, c = lane & 3;
    if (nn) {
      const int r = 4 * (rr & 3) + (rr >> 2);
      ph.lane_off = 4u * ((uint32_t)r * (uint32_t)ph.ldw + 4u * (uint32_t)c);
    } else {
      const int piece = c ^ ((4 - (rr >> 2)) & 3);
      ph.lane_off = 4u * ((uint32_t)rr * (uint32_t)ph.ldw + 4u * (uint32_t)piece);
    }
  }
}

template <int N>
__device__ __forceinline__ void rb_wait_vm_imm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Wait until at most n of this wave's vector-memory operations are outstanding (n wave-uniform, 0 .. 15).
__device__ __forceinline__ void rb_wait_vm(int n) {
  switch (n) {
    case 0: rb_wait_vm_imm<0>(); break;
    case 1: rb_wait_vm_imm<1>(); break;
    case 2: rb_wait_vm_imm<2>(); break;
    case 3: rb_wait_vm_imm<3>(); break;
    case 4: rb_wait_vm_imm<4>(); break;
    case 5: rb_wait_vm_imm<5>(); break;
    case 6: rb_wait_vm_imm<6>(); break;
    case 7: rb_wait_vm_imm<7>(); break;
    case 8: rb_wait_vm_imm<8>(); break;
    case 9: rb_wait_vm_imm<9>(); break;
    case 10: rb_wait_vm_imm<10>(); break;
    case 11: rb_wait_vm_imm<11>(); break;
    default: rb_wait_vm_imm<12>(); break;
  }
}

// Request the fragments of K-step s (16 kk) of the wave's tiles into ring stage `stage`.
template <int NW>
__device__ __forceinline__ void rb_issue_stage(const RbPhase& ph, uint8_t* ring, int stage, int s, int wave) {
  uint8_t* dst = ring + stage * ph.tv * RB_FRAG;
  for (int t = 0; t < ph.tv; ++t) {
    const int c0 = (ph.tbase + wave + NW * t) * 16;
    const size_t elem = ph.nn ? (size_t)(s * 16) * ph.ldw + c0 : (size_t)c0 * ph.ldw + s * 16;      // wave-uniform
    rb_gcc src = ph.W + elem * 4 + (uint64_t)ph.lane_off;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(dst + t * RB_FRAG), 16, 0, 0);
  }
}

// The first `depth` K-steps of a layer (all of them when the layer has fewer): the whole ring.
template <int NW>
__device__ __forceinline__ void rb_prologue(const RbPhase& ph, uint8_t* ring, int wave) {
  if (ph.tv == 0) return;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of the ring's previous contents are back
  const int n = (ph.depth < ph.ns) ? ph.depth : ph.ns;
  for (int s = 0; s < n; ++s) rb_issue_stage<NW>(ph, ring, s, s, wave);
}

// K-loop of one layer for a wave with TV tiles; its first `depth` K-steps are on their way already (rb_prologue).
// Per K-step s: wait for the fragments of K-step s, read them (and the activation chunk) into registers, run the MFMAs
// of K-step s - 1 from the other register set, and only then -- the reads are back by now -- refill the ring stage just
// read with K-step s + depth.  So the matrix pipe never waits for the LDS round trip, and the whole ring is in flight
// except for the one stage in registers.  Measured on the critical workgroup of the 16-survivor pack (512 x 512 layer, 16
// rows): this loop alone, weights not fetched, 39.9 k clocks (its MFMAs need 32.8 k); the DMA stream alone, MFMAs replaced
// by vector multiply-adds, 38.5 k clocks = 27 bytes per clock per CU, which is what LDS-DMA delivers from L2 here; both
// together 53.8 k.  A 16-row block needs a byte of weights for every 8 multiply-adds, so the two streams are the same
// length and whatever fails to overlap shows.
template <int MT, int TV, bool NN, int CAP>
__device__ __forceinline__ void rb_kloop(f32x4 (&acc)[RbCfg<MT>::TPW][MT], const RbPhase& ph, const uint8_t* act, uint8_t* ring,
                                         int wave, int lane) {
  const int i = lane & 15, g = lane >> 4;
  constexpr int D = CAP / TV;
  const int ns = ph.ns;
  int st = 0;                                    // ring stage of K-step s (s mod D)
  // image offset of this lane's activation chunk in even / odd K-steps (a 32-kk block is 128 bytes of an image row)
  int aoff[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row = m * 16 + i;
    aoff[m][0] = row * RB_PITCH + ((g ^ (row & 7)) << 4);
    aoff[m][1] = row * RB_PITCH + (((4 + g) ^ (row & 7)) << 4);
  }
  const int woff = NN ? ((g << 6) + ((i >> 2) << 4) + ((i & 3) << 2)) : ((4 * i + (g ^ ((4 - (i >> 2)) & 3))) << 4);

  auto fetch = [&](int s, f32x4 (&a)[MT], f32x4 (&w)[TV]) __attribute__((always_inline)) {
    // K-steps requested after K-step s and still possibly in flight: s + 1 .. s + D - 1, as far as the layer goes
    if (ns - 1 - s >= D - 1) rb_wait_vm_imm<(D - 1) * TV>();
    else rb_wait_vm((ns - 1 - s) * TV);
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(act + ((s >> 1) << 7) + aoff[m][s & 1]);
    const uint8_t* frag = ring + st * (TV * RB_FRAG) + woff;
#pragma unroll
    for (int t = 0; t < TV; ++t) {
      if constexpr (!NN) {
        w[t] = *reinterpret_cast<const f32x4*>(frag + t * RB_FRAG);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) w[t][e] = *reinterpret_cast<const float*>(frag + t * RB_FRAG + (e << 8));
      }
    }
  };
  auto refill = [&](int s) __attribute__((always_inline)) {       // K-step s has been read: its stage takes K-step s + D
    if (s + D < ns) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // a DMA may not overtake the reads of the stage it overwrites
      rb_issue_stage<RbCfg<MT>::NW>(ph, ring, st, s + D, wave);
    }
    st = (st + 1 == D) ? 0 : st + 1;
  };
  auto mma = [&](const f32x4 (&a)[MT], const f32x4 (&w)[TV]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < TV; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][e], a[m][e], acc[t][m], 0, 0, 0);
  };

  f32x4 a0[MT], w0[TV], a1[MT], w1[TV];
  fetch(0, a0, w0);
  refill(0);
  int s = 1;
  for (; s + 1 < ns; s += 2) {
    fetch(s, a1, w1);
    mma(a0, w0);
    refill(s);
    fetch(s + 1, a0, w0);
    mma(a1, w1);
    refill(s + 1);
  }
  if (s < ns) {
    fetch(s, a1, w1);
    mma(a0, w0);
    mma(a1, w1);
  } else {
    mma(a0, w0);
  }
}

template <int MT, bool NN, int CAP>
__device__ __forceinline__ void rb_run_kloop(f32x4 (&acc)[RbCfg<MT>::TPW][MT], const RbPhase& ph, const uint8_t* act, uint8_t* ring,
                                             int wave, int lane) {
#pragma unroll
  for (int t = 0; t < RbCfg<MT>::TPW; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  switch (ph.tv) {
    case 1: rb_kloop<MT, 1, NN, CAP>(acc, ph, act, ring, wave, lane); break;
    case 2: rb_kloop<MT, 2, NN, CAP>(acc, ph, act, ring, wave, lane); break;
    case 3: if constexpr (RbCfg<MT>::TPW >= 3) rb_kloop<MT, 3, NN, CAP>(acc, ph, act, ring, wave, lane); break;
    case 4: if constexpr (RbCfg<MT>::TPW >= 4) rb_kloop<MT, 4, NN, CAP>(acc, ph, act, ring, wave, lane); break;
    default: break;
  }
}

__device__ __forceinline__ void rb_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// The head over the rows of one 16-row record, by the four waves hw = 0..3 of row tile m: the arithmetic of
// mt_head_loss_body<1, true> (same row -> wave deal, same chains), reading H from the LDS image and leaving dZ of the
// last layer in its place.  rb_head_fetch asks for what it needs from memory early.
struct RbHeadRegs {
  float w[8], y[4], bias;
};

template <int MT>
__device__ __forceinline__ void rb_head_fetch(RbHeadRegs& hr, const MtHeadSlot& hs, int row0, int wave, int lane) {
#pragma unroll
  for (int e = 0; e < 8; ++e) hr.w[e] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) hr.y[q] = 0.f;
  hr.bias = 0.f;
  if (wave >= 4 * MT) return;
  const int m = wave >> 2, hw = wave & 3;
  const int crow0 = row0 + 16 * m;
  if (crow0 >= hs.B) return;
  const int nrows = (hs.B - crow0 < 16) ? hs.B - crow0 : 16;
  const int k0 = lane * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (k0 + e < hs.W) hr.w[e] = ((rb_gcf)hs.w)[k0 + e];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (hw + 4 * q < nrows) hr.y[q] = ((rb_gcf)hs.y)[crow0 + hw + 4 * q];
  hr.bias = ((rb_gcf)hs.b)[0];
}

template <int MT>
__device__ __forceinline__ void rb_head(const RbHeadRegs& hr, const MtHeadSlot& hs, uint8_t* act, float (*red)[4][RB_MAXW],
                                        float (*red_s)[4][2], int row0, int wave, int lane, bool to_mem) {
  if (wave >= 4 * MT) return;
  const int m = wave >> 2, hw = wave & 3;
  const int crow0 = row0 + 16 * m;
  if (crow0 >= hs.B) return;
  const int nrows = (hs.B - crow0 < 16) ? hs.B - crow0 : 16;
  const int k0 = lane * 8;
  const bool live = k0 < hs.W;
  float gw[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) gw[e] = 0.f;
  const float inv_b = 1.f / (float)hs.B;
  float loss_sum = 0.f, gb = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = hw + 4 * q;
    if (r >= nrows) break;
    const int lr = 16 * m + r;
    uint8_t* p0 = act + rb_act_off(lr, k0);
    uint8_t* p1 = act + rb_act_off(lr, k0 + 4);
    f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0;
    if (live) {
      h0 = *reinterpret_cast<const f32x4*>(p0);
      h1 = *reinterpret_cast<const f32x4*>(p1);
    }
    const float h[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dot = __builtin_fmaf(h[e], hr.w[e], dot);
    const float out = wave_sum(dot) + hr.bias;
    float l, gl;
    loss_and_grad(hs.loss_type, out, hr.y[q], l, gl);
    const float dout = gl * inv_b;
    loss_sum += l;
    gb += dout;
    if (live) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gw[e] = __builtin_fmaf(dout, h[e], gw[e]);
        o[e] = h[e] > 0.f ? dout * hr.w[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(p0) = f32x4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>(p1) = f32x4{o[4], o[5], o[6], o[7]};
      if (to_mem) {
        rb_gf dst = (rb_gf)hs.dH + (size_t)(crow0 + r) * hs.ldh + k0;
        *(rb_g4)dst = f32x4{o[0], o[1], o[2], o[3]};
        *(rb_g4)(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[m][hw][k0 + e] = gw[e];
  if (lane == 0) {
    red_s[m][hw][0] = loss_sum;
    red_s[m][hw][1] = gb;
  }
}

template <int MT>
__device__ __forceinline__ void rb_head_record(const MtHeadSlot& hs, const MtRbArgs& a, int slot, float (*red)[4][RB_MAXW],
                                               float (*red_s)[4][2], int row0, int wave, int lane) {
  if (wave >= 4 * MT) return;
  const int m = wave >> 2;
  const int crow0 = row0 + 16 * m;
  if (crow0 >= hs.B) return;
  const int t4 = (wave & 3) * 64 + lane;
  const float inv_b = 1.f / (float)hs.B;
  rb_gf out = (rb_gf)a.scratch + ((size_t)slot * a.max_chunks + (crow0 >> 4)) * a.rec;
  for (int k = t4; k < hs.W; k += 256) out[k] = (red[m][0][k] + red[m][1][k]) + (red[m][2][k] + red[m][3][k]);
  if (t4 == 0) {
    out[a.rec - 2] = (red_s[m][0][1] + red_s[m][1][1]) + (red_s[m][2][1] + red_s[m][3][1]);
    out[a.rec - 1] = ((red_s[m][0][0] + red_s[m][1][0]) + (red_s[m][2][0] + red_s[m][3][0])) * inv_b;
  }
}

// A lane's value that the optimiser must not treat as loop-invariant: what is derived from it (column indices, image
// offsets) is recomputed where it is used instead of being kept in registers, or spilled, across the layer loops.
__device__ __forceinline__ int rb_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// Outputs of the wave's tiles -> the LDS image (in place of the layer's input), between two barriers.
template <int MT>
__device__ __forceinline__ void rb_store_image(const f32x4 (&acc)[RbCfg<MT>::TPW][MT], uint8_t* act, int tv, int tbase, int wave,
                                               int lane, bool last_barrier = true) {
  rb_lds_barrier();          // every wave has read the layer's input for the last time
  const int i = rb_opaque(lane & 15), g = lane >> 4;
#pragma unroll
  for (int t = 0; t < RbCfg<MT>::TPW; ++t) {
    if (t < tv) {
      const int col = (tbase + wave + RbCfg<MT>::NW * t) * 16 + 4 * g;
#pragma unroll
      for (int m = 0; m < MT; ++m) *reinterpret_cast<f32x4*>(act + rb_act_off(m * 16 + i, col)) = acc[t][m];
    }
  }
  if (last_barrier) rb_lds_barrier();          // the output image is complete
}

// ---- split blocks: nparts workgroups (2 or 4) share one row block, each computing 1 / nparts of every layer's output
// columns (a quarter or half of the weights to stream and of the matrix instructions per workgroup).  After a layer each
// part needs the others' columns: they are in memory already (the weight-gradient launch reads H and dZ from there), so
// the exchange is a flag per workgroup.  A part publishes boundary number `seq` (counting up through the epoch: the
// engine zeroes the flags before an epoch's first step) once its stores are visible to the device, waits for its
// partners' flags to reach `seq`, and copies their columns into its LDS image.  The partners of a block sit 8 entries
// apart in the work list (same XCD under round-robin dispatch, consecutive in that XCD's dispatch order), and the engine
// never lists more workgroups than fit the chip at once, so a waiting part's partner is running or next in line.  The
// wait is bounded all the same: after RB_SPIN_LIMIT polls a part raises the error word behind the flags and goes on
// with whatever is in memory -- the grid always drains, and the engine refuses the epoch's results.
#define RB_SPIN_LIMIT (1 << 22)
// Device-scope accesses (sc1): the store is written through the XCD's L2, the load is not served from a cache that another
// XCD's stores do not reach.  Only the exchanged columns and the flags travel this way; a cache write-back / invalidate
// fence per boundary would also throw out the weights every workgroup of the XCD is streaming (measured: 45 us per
// boundary).  The asm loads are not tracked by the compiler's wait counters: rb_exchange waits for them itself.
__device__ __forceinline__ void rb_store4_dev(rb_g4 p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void rb_load4_dev(f32x4& v, rb_gc4 p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}

template <int MT>
__device__ __forceinline__ void rb_exchange(const MtRbArgs& a, int part, int nparts, int seq, rb_gcf out, int ldo, int N, int row0,
                                            int B, uint8_t* act, int tid, int wave, int lane) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's output stores are acknowledged, its image rows written
  __builtin_amdgcn_s_barrier();
  if (wave == 0) {
    if (lane == 0) __hip_atomic_store(a.flags + blockIdx.x, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane < nparts && lane != part) {
      const int32_t* f = a.flags + ((int)blockIdx.x + (lane - part) * 8);
      int spins = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > RB_SPIN_LIMIT) {
          __hip_atomic_store(a.flags + a.n_flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
  }
  __builtin_amdgcn_s_barrier();
  const int wq = N / nparts;                 // columns per part
  const int cpr = (N - wq) >> 2;             // 16-byte chunks of the other parts' columns per row
  const int total = 16 * MT * cpr;           // at most 16 MT x 96 chunks: RB_XCH per thread
  constexpr int RB_XCH = (16 * MT * (RB_MAXW * 3 / 16) + RbCfg<MT>::THREADS - 1) / RbCfg<MT>::THREADS;
  f32x4 v[RB_XCH];
  int off[RB_XCH];
#pragma unroll
  for (int u = 0; u < RB_XCH; ++u) {
    const int idx = tid + u * RbCfg<MT>::THREADS;
    off[u] = -1;
    v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (idx < total) {
      const int r = idx / cpr;
      int k = (idx - r * cpr) * 4;
      if (k >= part * wq) k += wq;           // step over this part's own columns
      if (row0 + r < B) {
        off[u] = rb_act_off(r, k);
        rb_load4_dev(v[u], (rb_gc4)(out + (size_t)(row0 + r) * ldo + k));
      }
    }
  }
#pragma unroll
  for (int u = 0; u < RB_XCH; ++u) asm volatile("" : "+v"(v[u]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int u = 0; u < RB_XCH; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
  for (int u = 0; u < RB_XCH; ++u)
    if (off[u] >= 0) *reinterpret_cast<f32x4*>(act + off[u]) = v[u];
  rb_lds_barrier();
}

// LDS of one workgroup: the activation image, the ReLU masks of every layer (one byte per tile and lane), the head's
// reduction buffers and the private weight rings.  16 rows: 34 + 16 + 8 + 96 KiB; 32 rows (at most 4 layers): 68 + 8 + 16 +
// 64 KiB.
template <int MT>
__global__ __launch_bounds__(RbCfg<MT>::THREADS) void mt_mlp_rowblock_kernel(MtRbArgs a, int step_imm,
                                                                     const int* __restrict__ step_base) {
  constexpr int RB = 16 * MT, CAP = RbCfg<MT>::CAP, NW = RbCfg<MT>::NW, TPW = RbCfg<MT>::TPW;
  __shared__ __attribute__((aligned(1024))) uint8_t rings[NW * CAP * RB_FRAG];
  __shared__ __attribute__((aligned(16))) uint8_t act[RB * RB_PITCH];
  __shared__ uint8_t mask[RbCfg<MT>::MASK_LAYERS * (RB_MAXW / 16) * 64];
  __shared__ float red[MT][4][RB_MAXW];
  __shared__ float red_s[MT][4][2];
  static_assert(HL_ROWS == 16, "one head record per 16-row MFMA row tile");

  const int step = mt_step(step_imm, step_base);
  int seq = step * 32;                          // exchange boundaries of a split block count up from here
  const MtRbBlock blk = a.blocks[blockIdx.x];
  if (blk.slot < 0) return;
  // row0 carries the split in its upper bits: part index (bits 20-23) and number of parts (bits 24-27; 0 reads as 1)
  const int slot = blk.slot, row0 = blk.row0 & 0xFFFFF;
  const int part = __builtin_amdgcn_readfirstlane((blk.row0 >> 20) & 15);
  const int nparts = __builtin_amdgcn_readfirstlane(((blk.row0 >> 24) & 15) > 1 ? ((blk.row0 >> 24) & 15) : 1);
  const MtHeadSlot hs = a.head[slot];
  if (step >= hs.active_steps || row0 >= hs.B) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* ring = rings + wave * (CAP * RB_FRAG);
  int L = 0;
  while (L < a.n_hidden && L < RbCfg<MT>::MASK_LAYERS && a.fwd[L][slot].B != nullptr) ++L;
  if (L == 0) return;

  // Diagnostic stamps (a.prof != null only in profiling runs): shader clock of thread 0 at the phase boundaries.
  int n_stamp = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if (a.prof != nullptr && tid == 0 && n_stamp < 16) a.prof[(size_t)blockIdx.x * 16 + n_stamp] = __builtin_amdgcn_s_memtime();
    ++n_stamp;
  };
  stamp();
  f32x4 acc[TPW][MT];
  RbPhase cur;
  {
    const MtGemmGroup g0 = a.fwd[0][slot];
    rb_make_phase<NW, TPW, CAP>(cur, g0, 0, 0, wave, lane, part, nparts);
    rb_prologue<NW>(cur, ring, wave);
    // the block's input rows -> LDS image (zero beyond the batch, as the tile kernel's staging does)
    rb_gcf X = (rb_gcf)rb_uniform64((uint64_t)g0.A);
    const int cpr = cur.K >> 2;                         // 16-byte chunks per image row
    for (int idx = tid; idx < RB * cpr; idx += RbCfg<MT>::THREADS) {
      const int r = idx / cpr, k = (idx - r * cpr) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row0 + r < hs.B) v = *(rb_gc4)(X + (size_t)(row0 + r) * g0.lda + k);
      *reinterpret_cast<f32x4*>(act + rb_act_off(r, k)) = v;
    }
  }
  // what the head needs from memory, asked for now: a load waited for later would also wait for every weight fragment
  // requested in between
  RbHeadRegs hr;
  rb_head_fetch<MT>(hr, hs, row0, wave, lane);
  rb_lds_barrier();
  stamp();

  // ---- forward layers: H_l = relu(H_{l-1} W_l^T + b_l)
  for (int l = 0; l < L; ++l) {
    RbPhase nxt = {};
    if (l + 1 < L) rb_make_phase<NW, TPW, CAP>(nxt, a.fwd[l + 1][slot], 0, l + 1, wave, lane, part, nparts);
    else if (L > 1) rb_make_phase<NW, TPW, CAP>(nxt, a.dgrad[L - 1][slot], 1, L - 1, wave, lane, part, nparts);
    const int i = rb_opaque(lane & 15), g = lane >> 4;
    // bias of the wave's tiles (the array is padded with zeros to a multiple of 8 columns), waited for before the next
    // layer's fragments are requested
    f32x4 bb[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      bb[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int col = (cur.tbase + wave + NW * t) * 16 + 4 * g;
      if (cur.bias != nullptr && t < cur.tv) bb[t] = *(rb_gc4)(cur.bias + col);
    }
    rb_run_kloop<MT, false, CAP>(acc, cur, act, ring, wave, lane);
    stamp();
#pragma unroll
    for (int t = 0; t < TPW; ++t) asm volatile("" : "+v"(bb[t]));
    rb_prologue<NW>(nxt, ring, wave);

    // H of the last layer is consumed by the head below, from LDS -- unless the block is split: then it travels
    const bool to_mem = cur.out != nullptr && (l + 1 < L || nparts > 1);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if (t < cur.tv) {
        const int tile = cur.tbase + wave + NW * t;
        const int col = tile * 16 + 4 * g;
        uint32_t bits = 0u;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 v = acc[t][m] + bb[t];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = fmaxf(v[j], 0.f);
            bits |= (v[j] > 0.f ? 1u : 0u) << (m * 4 + j);
          }
          acc[t][m] = v;
          const int row = row0 + m * 16 + i;
          if (to_mem && row < hs.B) {
            if (nparts > 1) rb_store4_dev((rb_g4)(cur.out + (size_t)row * cur.ldo + col), v);
            else *(rb_g4)(cur.out + (size_t)row * cur.ldo + col) = v;
          }
        }
        mask[(l * (RB_MAXW / 16) + tile) * 64 + lane] = (uint8_t)bits;
      }
    }
    rb_store_image<MT>(acc, act, cur.tv, cur.tbase, wave, lane, nparts == 1);
    if (nparts > 1) rb_exchange<MT>(a, part, nparts, ++seq, cur.out, cur.ldo, cur.N, row0, hs.B, act, tid, wave, lane);
    stamp();
    cur = nxt;
  }

  // ---- head, loss, dLoss/dH of the last layer (left in the image)
  // (every part of a split block runs the head on the whole rows: each needs dLoss/dH of all columns; part 0 writes)
  rb_head<MT>(hr, hs, act, red, red_s, row0, wave, lane, part == 0);
  rb_lds_barrier();
  if (part == 0) rb_head_record<MT>(hs, a, slot, red, red_s, row0, wave, lane);
  stamp();

  // ---- input gradients: dZ_{l-1} = (dZ_l W_l) * (H_{l-1} > 0), l = L-1 .. 1
  for (int l = L - 1; l >= 1; --l) {
    RbPhase nxt = {};
    if (l > 1) rb_make_phase<NW, TPW, CAP>(nxt, a.dgrad[l - 1][slot], 1, l - 1, wave, lane, part, nparts);
    rb_run_kloop<MT, true, CAP>(acc, cur, act, ring, wave, lane);
    stamp();
    rb_prologue<NW>(nxt, ring, wave);
    const int i = rb_opaque(lane & 15), g = lane >> 4;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if (t < cur.tv) {
        const int tile = cur.tbase + wave + NW * t;
        const int col = tile * 16 + 4 * g;
        const uint32_t bits = mask[((l - 1) * (RB_MAXW / 16) + tile) * 64 + lane];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 v = acc[t][m];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ((bits >> (m * 4 + j)) & 1u) ? v[j] : 0.f;
          acc[t][m] = v;
          const int row = row0 + m * 16 + i;
          if (cur.out != nullptr && row < hs.B) {
            if (nparts > 1) rb_store4_dev((rb_g4)(cur.out + (size_t)row * cur.ldo + col), v);
            else *(rb_g4)(cur.out + (size_t)row * cur.ldo + col) = v;
          }
        }
      }
    }
    if (l > 1) {                                                       // the last input gradient is read by nobody in here
      rb_store_image<MT>(acc, act, cur.tv, cur.tbase, wave, lane, nparts == 1);
      if (nparts > 1) rb_exchange<MT>(a, part, nparts, ++seq, cur.out, cur.ldo, cur.N, row0, hs.B, act, tid, wave, lane);
    }
    stamp();
    cur = nxt;
  }
}

// Diagnostics: a buffer of 16 stamps per workgroup that every later row-block launch of this process fills (null: off).
static unsigned long long* rb_prof_buffer = nullptr;
extern "C" void mt_rowblock_set_prof(unsigned long long* buf) { rb_prof_buffer = buf; }

extern "C" int mt_rb_args_sizeof() { return (int)sizeof(MtRbArgs); }
extern "C" int mt_rowblock_max_width() { return RB_MAXW; }

// blocks: n_blocks entries (slot, first row), one workgroup each; rows per workgroup = 16 * mt (mt = 1: the 32-row form,
// eight waves of four tiles, measured no faster on any pack and is no longer built), and a slot's entries must cover its
// batch in steps of that many rows.  Every layer of every listed slot must be at most
// mt_rowblock_max_width() wide (input rows included); the caller checks (the tables live in device memory).
extern "C" int mt_mlp_rowblock_launch(const MtRbArgs* args, int n_blocks, int mt, int step, hipStream_t stream) {
  if (n_blocks <= 0) return 0;
  if (args == nullptr || args->n_hidden < 1 || args->n_hidden > MT_MAX_LAYERS || args->blocks == nullptr ||
      args->head == nullptr || args->scratch == nullptr || args->rec < 10 || args->max_chunks < 1 ||
      (args->flags != nullptr && args->n_flags < n_blocks))
    return -2;
  MtRbArgs a = *args;
  a.prof = rb_prof_buffer;
  if (mt == 1)
    hipLaunchKernelGGL(mt_mlp_rowblock_kernel<1>, dim3(n_blocks), dim3(RbCfg<1>::THREADS), 0, stream, a, step, mt_step_base_tls);
  else
    return -2;
  return (int)hipGetLastError();
}
