// Batch assembly and loss of the single-trial captured step (gfx950).
//
// A captured training step of a small model is a chain of launches a few microseconds long each, so every launch
// that only shuffles indices or reduces 32 numbers is as expensive as a matrix product.  Two of those chains become
// one launch each:
//
//   mt_cursor_batch   x[i] = X[perm[pos + i]], y[i] = Y[perm[pos + i]] for i < B, with the epoch cursor `pos` and
//                     the permutation read from device memory (was: index add, three index_select launches)
//   mt_mse_fwd        loss = mean((out - y)^2), grad = 2 (out - y) / n kept for the backward pass, running += loss
//                     (was: elementwise square, mean reduction, accumulate, and an elementwise backward); the same
//                     single-workgroup launch can advance the epoch cursor, which every workgroup of mt_cursor_batch
//                     has finished reading by then (stream order)
//   mt_seed_refill    table[i] = counter + (i + 1) * stride for i < n, counter += n * stride: the per-step refill of
//                     the dropout seed table (ops/rng.py) in one launch instead of two
#include <hip/hip_runtime.h>
#include <stdint.h>

// One workgroup per batch row.  xe: floats per sample of X, ye: floats per sample of Y.  Out-of-range cursor positions
// and permutation entries are clamped (the trainer never issues them; a clamp is cheaper than a fault).
__global__ __launch_bounds__(256) void mt_cursor_batch_kernel(const int64_t* __restrict__ perm, const int64_t* __restrict__ pos,
                                                              long n_perm, const float* __restrict__ X,
                                                              const float* __restrict__ Y, long n_rows, long xe, long ye,
                                                              float* __restrict__ xb, float* __restrict__ yb) {
  const long i = blockIdx.x;
  long at = pos[0] + i;
  at = at < 0 ? 0 : (at >= n_perm ? n_perm - 1 : at);
  long src = perm[at];
  src = src < 0 ? 0 : (src >= n_rows ? n_rows - 1 : src);
  const float* xs = X + src * xe;
  float* xd = xb + i * xe;
  if ((xe & 3) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(xs);
    float4* d4 = reinterpret_cast<float4*>(xd);
    for (long k = threadIdx.x; k < (xe >> 2); k += 256) d4[k] = s4[k];
  } else {
    for (long k = threadIdx.x; k < xe; k += 256) xd[k] = xs[k];
  }
  for (long k = threadIdx.x; k < ye; k += 256) yb[i * ye + k] = Y[src * ye + k];
}

extern "C" int mt_cursor_batch_launch(const int64_t* perm, const int64_t* pos, long n_perm, const float* X, const float* Y,
                                      long n_rows, long xe, long ye, float* xb, float* yb, int B, hipStream_t stream) {
  if (B <= 0) return 0;
  if (!perm || !pos || !X || !Y || !xb || !yb || n_perm <= 0 || n_rows <= 0 || xe <= 0 || ye <= 0) return -2;
  if ((xe & 3) == 0 && ((((uintptr_t)X | (uintptr_t)xb) & 15) != 0)) return -3;
  hipLaunchKernelGGL(mt_cursor_batch_kernel, dim3(B), dim3(256), 0, stream, perm, pos, n_perm, X, Y, n_rows, xe, ye, xb, yb);
  return (int)hipGetLastError();
}

// One workgroup (n is a batch: tens to thousands of numbers).  loss and running are f32 scalars in device memory.
__global__ __launch_bounds__(256) void mt_mse_fwd_kernel(const float* __restrict__ out, const float* __restrict__ y, long n,
                                                         float* __restrict__ grad, float* __restrict__ loss,
                                                         float* __restrict__ running, int64_t* __restrict__ cursor,
                                                         long cursor_inc) {
  __shared__ float red[256];
  const float inv = 1.f / (float)n;
  float s = 0.f;
  for (long k = threadIdx.x; k < n; k += 256) {
    const float d = out[k] - y[k];
    grad[k] = 2.f * d * inv;
    s += d * d;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {            // fixed tree: bitwise repeatable
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float l = red[0] * inv;
    loss[0] = l;
    if (running != nullptr) running[0] += l;
    if (cursor != nullptr) cursor[0] += cursor_inc;
  }
}

extern "C" int mt_mse_fwd_launch(const float* out, const float* y, long n, float* grad, float* loss, float* running,
                                 int64_t* cursor, long cursor_inc, hipStream_t stream) {
  if (n <= 0 || !out || !y || !grad || !loss) return -2;
  hipLaunchKernelGGL(mt_mse_fwd_kernel, dim3(1), dim3(256), 0, stream, out, y, n, grad, loss, running, cursor, cursor_inc);
  return (int)hipGetLastError();
}

// One workgroup; n <= 256.  Every thread reads the counter before thread 0 advances it (barrier in between).
__global__ __launch_bounds__(256) void mt_seed_refill_kernel(int64_t* __restrict__ counter, int64_t* __restrict__ table, int n,
                                                             int64_t stride) {
  const uint64_t c = (uint64_t)counter[0];
  __syncthreads();
  const int i = threadIdx.x;
  if (i < n) table[i] = (int64_t)(c + (uint64_t)(i + 1) * (uint64_t)stride);      // wraps modulo 2^64
  if (i == 0) counter[0] = (int64_t)(c + (uint64_t)n * (uint64_t)stride);
}

extern "C" int mt_seed_refill_launch(int64_t* counter, int64_t* table, int n, int64_t stride, hipStream_t stream) {
  if (!counter || !table || n <= 0 || n > 256) return -2;
  hipLaunchKernelGGL(mt_seed_refill_kernel, dim3(1), dim3(256), 0, stream, counter, table, n, stride);
  return (int)hipGetLastError();
}
