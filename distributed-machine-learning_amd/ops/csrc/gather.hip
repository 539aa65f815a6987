// Mini-batch gather for packed trials (gfx950): the device-side replacement of a shuffling DataLoader.
//
// For slot s and batch row i the kernel computes which dataset row the trial visits at position
// step*B + i of its current epoch (keyed Feistel permutation, no index array), reads that f32 row
// and writes it as bf16 (or, for the exact-f32 engine, as f32) into the slot's layer-0 activation buffer, plus the
// f32 target.  Rows are padded with zeros up to Fpad (a multiple of 8) so the first GEMM can load 16-byte chunks.
#include "mt_bodies.h"

template <bool F32>
__global__ __launch_bounds__(256) void mt_gather_kernel(const MtGatherSlot* __restrict__ slots,
                                                        const int32_t* __restrict__ ids,
                                                        const float* __restrict__ X, const float* __restrict__ Y,
                                                        uint32_t n_rows, int F, int Fpad, int hb, int step_imm,
                                                        const int* __restrict__ step_base) {
  mt_gather_body<F32>(slots, ids, X, Y, n_rows, F, Fpad, hb, mt_step(step_imm, step_base), blockIdx.x, blockIdx.y);
}

static int gather_launch(const MtGatherSlot* slots, const int32_t* ids, int n_slots, int max_batch, const float* X,
                         const float* Y, uint32_t n_rows, int F, int Fpad, int hb, int step, bool f32,
                         hipStream_t stream) {
  if (n_slots <= 0) return 0;
  const int cpr = f32 ? Fpad / 4 : Fpad / 8;
  if (Fpad % 8 != 0 || cpr > 256 || Fpad < F) return -2;
  const int rows_per_block = 256 / cpr;
  dim3 grid((max_batch + rows_per_block - 1) / rows_per_block, n_slots), block(256);
  if (f32)
    hipLaunchKernelGGL(mt_gather_kernel<true>, grid, block, 0, stream, slots, ids, X, Y, n_rows, F, Fpad, hb, step,
                       mt_step_base_tls);
  else
    hipLaunchKernelGGL(mt_gather_kernel<false>, grid, block, 0, stream, slots, ids, X, Y, n_rows, F, Fpad, hb, step,
                       mt_step_base_tls);
  return (int)hipGetLastError();
}

extern "C" int mt_gather_launch(const MtGatherSlot* slots, const int32_t* ids, int n_slots, int max_batch, const float* X,
                                const float* Y, uint32_t n_rows, int F, int Fpad, int hb, int step,
                                hipStream_t stream) {
  return gather_launch(slots, ids, n_slots, max_batch, X, Y, n_rows, F, Fpad, hb, step, false, stream);
}

// Same gather with f32 output rows (the exact-f32 engine).
extern "C" int mt_gather_f32_launch(const MtGatherSlot* slots, const int32_t* ids, int n_slots, int max_batch,
                                    const float* X, const float* Y, uint32_t n_rows, int F, int Fpad, int hb, int step,
                                    hipStream_t stream) {
  return gather_launch(slots, ids, n_slots, max_batch, X, Y, n_rows, F, Fpad, hb, step, true, stream);
}
