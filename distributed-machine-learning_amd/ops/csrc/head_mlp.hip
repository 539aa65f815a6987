// Fused regression-head MLP (up to 5 Linear layers with ReLU between them), forward and backward, f32
// (gfx950).  The shape it exists for is the Transformer regressor's head d -> 128 -> 64 -> 32 -> 16 -> 1
// applied to the LAST token of each sequence.
//
// As separate ops that head is 5 GEMMs + 4 ReLUs forward and ~15 launches backward on a [batch, d] matrix
// with batch = 32: pure launch latency.  Here forward is ONE launch and backward TWO:
//
//   forward      one workgroup per 8 rows; the input rows and every layer's activations stay in LDS; each
//                weight matrix is staged through LDS in [out][64]-column slabs (coalesced global reads,
//                conflict-free padded LDS reads), so weights are read once per workgroup.  The input is
//                addressed as x + row * ldx, which folds the "take the last token" slice into the kernel:
//                no gather, no copy.
//   backward A   per 8 rows: dz_l for every layer (ReLU masks from the saved activations) and the input
//                gradient; weights are read coalesced along the input dimension.
//   backward B   one thread per weight element over all layers: dW_l[o][k] = sum_r dz_l[r][o] * a_{l-1}[r][k]
//                in row order (deterministic, no atomics); the k == 0 threads also produce the bias gradient.
#include "mt_common.h"

#define HM_MAXL 5
#define HM_ROWS 8
#define HM_MAXH 128      // widest hidden layer
#define HM_KSLAB 64

struct MtHeadMlp {
  const float* W[HM_MAXL];   // [dims[l+1]][dims[l]] row-major (torch Linear layout)
  const float* b[HM_MAXL];
  float* dW[HM_MAXL];
  float* db[HM_MAXL];
  int32_t dims[HM_MAXL + 1]; // dims[0] = input width, dims[n_layers] = output width
  int32_t n_layers;
  int32_t act_stride;        // sum of dims[1..n_layers-1]: saved hidden activations per row
  int32_t dz_stride;         // sum of dims[1..n_layers]
};

extern __shared__ float hm_smem[];

// acts: [B][act_stride] hidden activations (post-ReLU), out: [B][dims[n_layers]]
__global__ __launch_bounds__(256) void mt_head_mlp_fwd_kernel(MtHeadMlp hd, const float* __restrict__ x, long ldx,
                                                              float* __restrict__ acts, float* __restrict__ out, int B) {
  const int D0 = hd.dims[0];
  float* xs = hm_smem;                               // [HM_ROWS][D0]
  float* a0 = xs + HM_ROWS * D0;                     // [HM_ROWS][HM_MAXH]
  float* a1 = a0 + HM_ROWS * HM_MAXH;                // [HM_ROWS][HM_MAXH]
  float* ws = a1 + HM_ROWS * HM_MAXH;                // [HM_MAXH][HM_KSLAB + 1]
  const int t = threadIdx.x;
  const int row0 = blockIdx.x * HM_ROWS;
  for (int i = t; i < HM_ROWS * D0; i += 256) {
    const int r = i / D0, k = i % D0;
    xs[i] = (row0 + r < B) ? x[(long)(row0 + r) * ldx + k] : 0.f;
  }
  const float* in = xs;
  int K = D0, act_off = 0;
  float* cur = a0;
  for (int l = 0; l < hd.n_layers; ++l) {
    const int D = hd.dims[l + 1];
    const int groups = 256 / D;                                      // D <= 128: at least two row groups
    const int rpt = (HM_ROWS + groups - 1) / groups;                 // rows per thread
    const int o = t % D, g = t / D;
    float acc[HM_ROWS];
#pragma unroll
    for (int r = 0; r < HM_ROWS; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += HM_KSLAB) {
      const int kw = (K - k0) < HM_KSLAB ? (K - k0) : HM_KSLAB;
      __syncthreads();                                               // previous slab (and `in`) fully consumed/produced
      for (int i = t; i < D * HM_KSLAB; i += 256) {
        const int oo = i / HM_KSLAB, kk = i % HM_KSLAB;
        ws[oo * (HM_KSLAB + 1) + kk] = kk < kw ? hd.W[l][(long)oo * K + k0 + kk] : 0.f;
      }
      __syncthreads();
      if (g * rpt < HM_ROWS) {
        for (int kk = 0; kk < kw; ++kk) {
          const float wv = ws[o * (HM_KSLAB + 1) + kk];
#pragma unroll
          for (int r = 0; r < HM_ROWS; ++r)
            if (r < rpt) acc[r] += wv * in[(g * rpt + r) * K + k0 + kk];
        }
      }
    }
    const bool last = (l == hd.n_layers - 1);
    const float bv = hd.b[l][o];
    __syncthreads();                                                 // everyone is done reading `in`
#pragma unroll
    for (int r = 0; r < HM_ROWS; ++r) {
      const int rr = g * rpt + r;
      if (r < rpt && rr < HM_ROWS) {
        float v = acc[r] + bv;
        if (!last) v = v > 0.f ? v : 0.f;
        cur[rr * D + o] = v;
        if (row0 + rr < B) {
          if (last) out[(long)(row0 + rr) * D + o] = v;
          else acts[(long)(row0 + rr) * hd.act_stride + act_off + o] = v;
        }
      }
    }
    if (!last) act_off += D;
    in = cur;
    cur = (cur == a0) ? a1 : a0;
    K = D;
  }
}

// dz: [B][dz_stride] pre-activation gradients of every layer (layer 1 first); dx: [B][lddx] input gradient.
__global__ __launch_bounds__(256) void mt_head_mlp_bwd_rows_kernel(MtHeadMlp hd, const float* __restrict__ acts,
                                                                   const float* __restrict__ dout, float* __restrict__ dz,
                                                                   float* __restrict__ dx, long lddx, int B) {
  __shared__ float za[HM_ROWS][HM_MAXH], zb[HM_ROWS][HM_MAXH];
  const int t = threadIdx.x;
  const int row0 = blockIdx.x * HM_ROWS;
  const int L = hd.n_layers;
  float (*cur)[HM_MAXH] = za;
  float (*nxt)[HM_MAXH] = zb;
  int dz_off = hd.dz_stride - hd.dims[L];
  for (int i = t; i < HM_ROWS * hd.dims[L]; i += 256) {
    const int r = i / hd.dims[L], o = i % hd.dims[L];
    const float v = (row0 + r < B) ? dout[(long)(row0 + r) * hd.dims[L] + o] : 0.f;
    cur[r][o] = v;
    if (row0 + r < B) dz[(long)(row0 + r) * hd.dz_stride + dz_off + o] = v;
  }
  __syncthreads();
  int act_off = hd.act_stride;
  for (int l = L - 1; l >= 0; --l) {
    const int D = hd.dims[l + 1], K = hd.dims[l];
    if (l > 0) { act_off -= K; dz_off -= K; }
    for (int i = t; i < HM_ROWS * K; i += 256) {
      const int r = i / K, k = i % K;                   // lanes run along k: W[o][k] reads are coalesced
      float s = 0.f;
      for (int o = 0; o < D; ++o) s += cur[r][o] * hd.W[l][(long)o * K + k];
      if (l > 0) {
        const bool on = (row0 + r < B) && acts[(long)(row0 + r) * hd.act_stride + act_off + k] > 0.f;
        s = on ? s : 0.f;
        nxt[r][k] = s;
        if (row0 + r < B) dz[(long)(row0 + r) * hd.dz_stride + dz_off + k] = s;
      } else if (dx != nullptr && row0 + r < B) {
        dx[(long)(row0 + r) * lddx + k] = s;
      }
    }
    __syncthreads();
    float (*tmp)[HM_MAXH] = cur; cur = nxt; nxt = tmp;
  }
}

// One thread per weight element of every layer.
__global__ __launch_bounds__(256) void mt_head_mlp_bwd_weights_kernel(MtHeadMlp hd, const float* __restrict__ x, long ldx,
                                                                      const float* __restrict__ acts,
                                                                      const float* __restrict__ dz, int B, long total) {
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  int l = 0, act_off = 0, dz_off = 0;
  for (; l < hd.n_layers; ++l) {
    const long sz = (long)hd.dims[l + 1] * hd.dims[l];
    if (e < sz) break;
    e -= sz;
    if (l > 0) act_off += hd.dims[l];
    dz_off += hd.dims[l + 1];
  }
  const int K = hd.dims[l];
  const int o = (int)(e / K), k = (int)(e % K);
  const float* a = (l == 0) ? x : acts + act_off;
  const long lda = (l == 0) ? ldx : hd.act_stride;
  float s = 0.f, sb = 0.f;
  for (int r = 0; r < B; ++r) {
    const float z = dz[(long)r * hd.dz_stride + dz_off + o];
    s += z * a[(long)r * lda + k];
    sb += z;
  }
  hd.dW[l][e] = s;
  if (k == 0) hd.db[l][o] = sb;
}

extern "C" int mt_head_mlp_sizeof() { return (int)sizeof(MtHeadMlp); }

static int hm_check(const MtHeadMlp* hd) {
  if (hd->n_layers < 1 || hd->n_layers > HM_MAXL) return -2;
  if (hd->dims[0] < 1 || hd->dims[0] > 2048) return -2;
  int act = 0, dzs = 0;
  for (int l = 1; l <= hd->n_layers; ++l) {
    const int D = hd->dims[l];
    if (D < 1 || D > HM_MAXH) return -2;
    if (l < hd->n_layers) act += D;
    dzs += D;
  }
  if (act != hd->act_stride || dzs != hd->dz_stride) return -5;
  return 0;
}

extern "C" int mt_head_mlp_fwd_launch(const MtHeadMlp* hd, const float* x, long ldx, float* acts, float* out, int B,
                                      hipStream_t stream) {
  const int rc = hm_check(hd);
  if (rc) return rc;
  if (B <= 0) return 0;
  const size_t smem = sizeof(float) * ((size_t)HM_ROWS * hd->dims[0] + 2 * HM_ROWS * HM_MAXH + HM_MAXH * (HM_KSLAB + 1));
  static bool big_lds = false;
  if (!big_lds) {      // input rows + activations + one weight slab can pass the 64 KiB default (the CU has 160 KiB)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mt_head_mlp_fwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) return -7;
    big_lds = true;
  }
  hipLaunchKernelGGL(mt_head_mlp_fwd_kernel, dim3((B + HM_ROWS - 1) / HM_ROWS), dim3(256), smem, stream, *hd, x, ldx, acts,
                     out, B);
  return (int)hipGetLastError();
}

extern "C" int mt_head_mlp_bwd_launch(const MtHeadMlp* hd, const float* x, long ldx, const float* acts, const float* dout,
                                      float* dz, float* dx, long lddx, int B, hipStream_t stream) {
  const int rc = hm_check(hd);
  if (rc) return rc;
  if (B <= 0) return 0;
  hipLaunchKernelGGL(mt_head_mlp_bwd_rows_kernel, dim3((B + HM_ROWS - 1) / HM_ROWS), dim3(256), 0, stream, *hd, acts, dout,
                     dz, dx, lddx, B);
  long total = 0;
  for (int l = 0; l < hd->n_layers; ++l) total += (long)hd->dims[l + 1] * hd->dims[l];
  hipLaunchKernelGGL(mt_head_mlp_bwd_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, *hd, x,
                     ldx, acts, dz, B, total);
  return (int)hipGetLastError();
}
