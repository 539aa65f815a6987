// Mini-batch gather for packed trials (gfx950): the device-side replacement of a shuffling DataLoader.
//
// For slot s and batch row i the kernel computes which dataset row the trial visits at position
// step*B + i of its current epoch (keyed Feistel permutation, no index array), reads that f32 row
// and writes it as bf16 into the slot's layer-0 activation buffer, plus the f32 target.  Rows are
// padded with zeros up to Fpad (a multiple of 8) so the first GEMM can load 16-byte chunks.
#include "mt_bodies.h"

__global__ __launch_bounds__(256) void mt_gather_kernel(const MtGatherSlot* __restrict__ slots,
                                                        const int32_t* __restrict__ ids,
                                                        const float* __restrict__ X, const float* __restrict__ Y,
                                                        uint32_t n_rows, int F, int Fpad, int hb, int step_imm,
                                                        const int* __restrict__ step_base) {
  mt_gather_body(slots, ids, X, Y, n_rows, F, Fpad, hb, mt_step(step_imm, step_base), blockIdx.x, blockIdx.y);
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
