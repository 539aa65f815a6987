// How fast can one 16-wave workgroup per CU stream a weight matrix that sits in its XCD's L2: by LDS-DMA into per-wave LDS
// rings (what rowblock.hip does) or by plain 16-byte global loads into registers (MFMA fragment shape)?  Both forms read
// W[512][512] f32 once per workgroup, wave w taking 16-row tiles w and w + 16, 16 kk per step, and fold what they read
// into a checksum so that nothing is optimised away.  256 workgroups; 16 of them share one matrix (one trial's blocks).
//   hipcc --offload-arch=gfx950 -O3 -o stream_paths stream_paths.hip && ./stream_paths
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define N 512
#define K 512
#define NW 16
#define FRAG 1024

__global__ __launch_bounds__(1024) void dma_kernel(const float* __restrict__ Wall, float* out, long long* clk, int reps) {
  __shared__ __attribute__((aligned(1024))) unsigned char rings[NW * 6 * FRAG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* W = Wall + (size_t)((blockIdx.x / 8) % 16) * N * K;
  unsigned char* ring = rings + wave * 6 * FRAG;
  const int rr = lane >> 2, c = lane & 3;
  const unsigned lane_off = 4u * (rr * K + 4 * c);
  f32x4 acc = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    const int ns = K / 16;
    auto issue = [&](int stage, int s) {
      for (int t = 0; t < 2; ++t) {
        const int c0 = (wave + NW * t) * 16;
        const char* src = (const char*)W + ((size_t)c0 * K + s * 16) * 4 + lane_off;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ring + (stage * 2 + t) * FRAG), 16, 0, 0);
      }
    };
    for (int s = 0; s < 3; ++s) issue(s, s);
    int st = 0;
    for (int s = 0; s < ns; ++s) {
      if (ns - 1 - s >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (ns - 1 - s == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(ring + (st * 2) * FRAG + lane * 16);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(ring + (st * 2 + 1) * FRAG + lane * 16);
      acc += w0 + w1;
      if (s + 3 < ns) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue(st, s + 3);
      }
      st = st == 2 ? 0 : st + 1;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[(size_t)blockIdx.x * 1024 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int D>
__global__ __launch_bounds__(1024) void reg_kernel(const float* __restrict__ Wall, float* out, long long* clk, int reps) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __attribute__((address_space(1))) char* W =
      (const __attribute__((address_space(1))) char*)(Wall + (size_t)((blockIdx.x / 8) % 16) * N * K);
  const int i = lane & 15, g = lane >> 4;
  const unsigned lane_off = 4u * (i * K + 4 * g);
  f32x4 acc = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    const int ns = K / 16, last = ns - 1;
    f32x4 w[D][2];
    auto load = [&](f32x4& dst, int t, int s) {
      const size_t u = ((size_t)((wave + NW * t) * 16) * K + (size_t)s * 16) * 4;
      dst = *(const __attribute__((address_space(1))) f32x4*)(W + u + (unsigned long long)lane_off);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) { load(w[d][0], 0, d); load(w[d][1], 1, d); }
    for (int s0 = 0; s0 < ns; s0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        acc += w[d][0] + w[d][1];
        const int sn = s0 + d + D < last ? s0 + d + D : last;
        load(w[d][0], 0, sn);
        load(w[d][1], 1, sn);
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc += w[d][0] + w[d][1];
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[(size_t)blockIdx.x * 1024 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
  float* W; float* out; long long* clk;
  const size_t wbytes = (size_t)16 * N * K * 4;
  hipMalloc(&W, wbytes); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 256 * 8);
  std::vector<float> h(16 * N * K);
  for (size_t j = 0; j < h.size(); ++j) h[j] = (float)(j % 97) * 0.01f;
  hipMemcpy(W, h.data(), wbytes, hipMemcpyHostToDevice);
  std::vector<long long> c(256);
  const int reps = 4;
  for (int form = 0; form < 4; ++form) {
    for (int it = 0; it < 3; ++it) {
      if (form == 0) hipLaunchKernelGGL(dma_kernel, dim3(256), dim3(1024), 0, 0, W, out, clk, reps);
      if (form == 1) hipLaunchKernelGGL(reg_kernel<2>, dim3(256), dim3(1024), 0, 0, W, out, clk, reps);
      if (form == 2) hipLaunchKernelGGL(reg_kernel<4>, dim3(256), dim3(1024), 0, 0, W, out, clk, reps);
      if (form == 3) hipLaunchKernelGGL(reg_kernel<8>, dim3(256), dim3(1024), 0, 0, W, out, clk, reps);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    }
    hipMemcpy(c.data(), clk, 256 * 8, hipMemcpyDeviceToHost);
    long long mx = 0, sum = 0;
    for (auto v : c) { mx = v > mx ? v : mx; sum += v; }
    const char* names[] = {"LDS-DMA ring, 3 K-steps ahead", "registers, 2 K-steps ahead", "registers, 4 K-steps ahead", "registers, 8 K-steps ahead"};
    // s_memtime ticks at 100 MHz
    printf("%-34s mean %.1f ticks  max %lld ticks per %d passes over 1 MiB -> %.1f GB/s per CU (mean)\n", names[form],
           (double)sum / 256, mx, reps, reps * 1048576.0 / ((double)sum / 256 / 100e6) / 1e9);
  }
  return 0;
}
