// Mini-batch gather for packed trials (gfx950): the device-side replacement of a shuffling DataLoader.
//
// For slot s and batch row i the kernel computes which dataset row the trial visits at position
// step*B + i of its current epoch (keyed Feistel permutation, no index array), reads that f32 row
// and writes it as bf16 into the slot's layer-0 activation buffer, plus the f32 target.  Rows are
// padded with zeros up to Fpad (a multiple of 8) so the first GEMM can load 16-byte chunks.
#include "mt_common.h"

__global__ __launch_bounds__(256) void mt_gather_kernel(const MtGatherSlot* __restrict__ slots,
                                                        const int32_t* __restrict__ ids,
                                                        const float* __restrict__ X, const float* __restrict__ Y,
                                                        uint32_t n_rows, int F, int Fpad, int hb, int step_imm,
                                                        const int* __restrict__ step_base) {
  const int step = mt_step(step_imm, step_base);
  const MtGatherSlot s = slots[ids[blockIdx.y]];
  if (step >= s.active_steps) return;
  const int cpr = Fpad >> 3;                 // 16-byte output chunks per row
  const int rows_per_block = 256 / cpr;
  const int tid = threadIdx.x;
  const int r = tid / cpr, c = tid - r * cpr;
  if (r >= rows_per_block) return;
  const int i = blockIdx.x * rows_per_block + r;
  if (i >= s.B) return;
  const uint32_t pos = (uint32_t)step * (uint32_t)s.B + (uint32_t)i;
  const uint32_t src = mt_permute(pos, n_rows, s.key, hb);
  const float* xr = X + (size_t)src * F;
  float v[8];
  const int f0 = c * 8;
  if ((F & 3) == 0 && f0 + 8 <= F) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(xr + f0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(xr + f0 + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (f0 + e < F) ? xr[f0 + e] : 0.f;
  }
  u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(s.X0) + (size_t)i * Fpad + f0) = o;
  if (c == 0) s.y[i] = Y[src];
}

extern "C" int mt_gather_launch(const MtGatherSlot* slots, const int32_t* ids, int n_slots, int max_batch, const float* X,
                                const float* Y, uint32_t n_rows, int F, int Fpad, int hb, int step,
                                hipStream_t stream) {
  if (n_slots <= 0) return 0;
  if (Fpad % 8 != 0 || Fpad / 8 > 256 || Fpad < F) return -2;
  const int rows_per_block = 256 / (Fpad / 8);
  dim3 grid((max_batch + rows_per_block - 1) / rows_per_block, n_slots), block(256);
  hipLaunchKernelGGL(mt_gather_kernel, grid, block, 0, stream, slots, ids, X, Y, n_rows, F, Fpad, hb, step,
                     mt_step_base_tls);
  return (int)hipGetLastError();
}
