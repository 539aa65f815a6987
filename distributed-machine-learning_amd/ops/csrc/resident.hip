// Resident epoch kernel for exact-f32 MLP packs (gfx950): a trial's whole training epoch inside ONE launch.
//
// The grouped launches of runtime.hip advance every trial of a pack in lockstep: 2L + 3 dependent launches per
// optimiser step, each of which has to drain before the next starts.  Late in a search a GPU holds a dozen survivors of
// different widths; their layers are a few dozen to a few hundred tiles per launch, so most of a step is launch
// boundaries and half-empty tails, and the matrix pipes idle (measured: 16 survivors, 147 us per step, 23 % of the f32
// matrix peak).  Trials do not depend on each other, so nothing forces the lockstep.  Here the grid is cut into GROUPS
// of workgroups; a group owns a list of trial slots and trains them one after the other, each for its whole epoch:
//
//   for slot in group's list:
//     gather(step 0) | for step: { L x forward | head + loss | L x (weight gradient, input gradient) | optimiser,
//                                  gather(step + 1) }
//
// where "|" is a barrier among the workgroups OF THAT GROUP ONLY (a counter in device memory).  Groups never wait for
// each other, so while one trial sits in a narrow layer another group's wide layer keeps the matrix pipes busy, and no
// launch boundary exists inside an epoch.  Every phase calls the same device bodies as the grouped launches (GEMM tile,
// head/loss, head reduction, optimiser block, mini-batch gather) on the same descriptor tables, with the weight gradient
// unsplit, so a trial's numbers are bit for bit those of the grouped path.
//
// Residency: a group barrier needs every workgroup of the group on the chip.  The launcher sizes the grid from the
// occupancy the runtime reports (never more workgroups than fit at once) and the host gives a group consecutive
// positions of one XCD (workgroup i runs on XCD i % 8), so a group also shares one L2.  Should the chip be shared with
// another process' resident kernel and a group never complete, every waiting workgroup gives up after `timeout_ticks`
// of the 100 MHz clock, raises the error word and leaves; all others see the word at their next barrier and leave too:
// the grid always drains, and the host reports the epoch as failed.
#include "mt_gemm_f32_tile.h"
#include "mt_head_body.h"
#include "mt_optim_body.h"

#define RES_MAX_LAYERS 8

struct MtResGroup {
  int32_t first, count;     // this group's slots: slot_list[first .. first + count)
  int32_t nwg, pad;         // workgroups in the group
};

struct MtResWg {
  int32_t group, member;    // per workgroup: which group, which of its nwg members
};

struct MtResidentArgs {
  const MtGemmGroup* fwd[RES_MAX_LAYERS];     // per-layer group tables, indexed by physical slot
  const MtGemmGroup* dgrad[RES_MAX_LAYERS];
  const MtGemmGroup* wgrad[RES_MAX_LAYERS];
  const MtGatherSlot* gather;
  const MtHeadSlot* head;
  const MtOptSlot* opt;
  const int32_t* slot_list;     // physical slot ids, group after group
  const MtResGroup* groups;
  const MtResWg* wgs;           // one entry per workgroup of the grid
  float* P;
  float* G;
  float* M;
  float* V;
  const float* X;
  const float* Y;
  float* head_scratch;
  unsigned int* sync;           // one arrival counter per group, zero at launch
  int* error;                   // set to 1 when a barrier timed out
  unsigned long long timeout_ticks;
  size_t slot_stride;
  uint32_t n_rows;
  int32_t F, Fpad, hb;
  int32_t n_hidden;             // layers the tables cover (a slot may use fewer: its group entries are empty)
  int32_t max_chunks, rec;      // layout of the head kernel's partial records
};

// Barrier among the nwg workgroups of one group.  Every wave publishes its stores device-wide before its workgroup
// arrives, and refreshes its view after the last arrival: the members of a group may sit on different XCDs.
__device__ __forceinline__ bool res_group_barrier(unsigned int* counter, unsigned int target, int* error,
                                                  unsigned long long timeout_ticks, int* ok_s) {
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    const unsigned long long t0 = wall_clock64();
    unsigned int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63u) == 0u) {
        if (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
        if (wall_clock64() - t0 > timeout_ticks) {
          __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    *ok_s = ok;
  }
  __syncthreads();
  __threadfence();
  return *ok_s != 0;
}

template <int T>
__device__ __forceinline__ int res_tiles(int n) { return (n + T - 1) / T; }

template <int T>
__global__ __launch_bounds__(F32_THREADS) void mt_mlp_resident_kernel(MtResidentArgs a) {
  // the GEMM tile's two stages; the head body's reduction arrays alias the front of it
  __shared__ __attribute__((aligned(16))) uint8_t lds[(2 * 2 * T * 128) > 8704 ? (2 * 2 * T * 128) : 8704];
  __shared__ int ok_s;
  float (*red)[512] = reinterpret_cast<float (*)[512]>(lds);
  float (*red_s)[2] = reinterpret_cast<float (*)[2]>(lds + HL_WAVES * 512 * 4);

  const MtResWg me = a.wgs[blockIdx.x];
  if (me.group < 0) return;                       // padding workgroup
  const MtResGroup grp = a.groups[me.group];
  const int w = me.member, nwg = grp.nwg;
  unsigned int* counter = a.sync + me.group;
  unsigned int target = 0u;
#define RES_BARRIER()                                                                                   \
  do {                                                                                                  \
    target += (unsigned int)nwg;                                                                        \
    if (!res_group_barrier(counter, target, a.error, a.timeout_ticks, &ok_s)) return;                   \
  } while (0)

  const int cpr = a.Fpad / 4;                     // gather: 16-byte chunks per row, rows per 256-thread block
  const int g_rows = 256 / cpr;

  for (int si = 0; si < grp.count; ++si) {
    const int by = grp.first + si;                // index into slot_list (the bodies look the slot up through it)
    const int slot = a.slot_list[by];
    const MtGatherSlot gs = a.gather[slot];
    const int nsteps = gs.active_steps, B = gs.B;
    int L = 0;
    while (L < a.n_hidden && a.fwd[L][slot].I > 0) ++L;
    if (nsteps <= 0 || L == 0) continue;
    const int gather_items = (B + g_rows - 1) / g_rows;
    const int head_items = (B + HL_ROWS - 1) / HL_ROWS;
    const int opt_items = (a.opt[slot].n_params + OPT_PER_BLOCK - 1) / OPT_PER_BLOCK;

    for (int k = w; k < gather_items; k += nwg)
      mt_gather_body<true>(a.gather, a.slot_list, a.X, a.Y, a.n_rows, a.F, a.Fpad, a.hb, 0, k, by);
    RES_BARRIER();

    for (int step = 0; step < nsteps; ++step) {
      // ---- forward
      for (int l = 0; l < L; ++l) {
        const int tj_n = res_tiles<T>(a.fwd[l][slot].J);
        const int n = res_tiles<T>(B) * tj_n;
        for (int k = w; k < n; k += nwg) {
          MtTile t;
          t.group = slot; t.ti = (int16_t)(k / tj_n); t.tj = (int16_t)(k % tj_n);
          f32_gemm_tile<0, 2, T>(a.fwd[l], t, step, MT_GEMM_FLAG_RELU, 0, 0, 1, 0, lds);
          __syncthreads();
        }
        RES_BARRIER();
      }
      // ---- head + loss + gradient entering the last hidden layer (partial records per 32 rows)
      for (int k = w; k < head_items; k += nwg) {
        mt_head_loss_body<1, true>(a.head, a.slot_list, step, a.head_scratch, a.max_chunks, a.rec, k, by, red, red_s);
        __syncthreads();
      }
      RES_BARRIER();
      // ---- backward, last layer first: weight-gradient tiles, input-gradient tiles, and (once) the head reduction
      for (int l = L - 1; l >= 0; --l) {
        const MtGemmGroup* wg = a.wgrad[l];
        const int wi_n = res_tiles<T>(wg[slot].I), wj_n = res_tiles<T>(wg[slot].J);
        const int n_w = wi_n * wj_n;
        int dj_n = 1, n_d = 0;
        if (l > 0) {
          dj_n = res_tiles<T>(a.dgrad[l][slot].J);
          n_d = res_tiles<T>(B) * dj_n;
        }
        const int n = n_w + n_d + (l == L - 1 ? 1 : 0);
        for (int k = w; k < n; k += nwg) {
          MtTile t;
          t.group = slot;
          if (k < n_w) {
            t.ti = (int16_t)(k / wj_n); t.tj = (int16_t)(k % wj_n);
            f32_gemm_tile<2, 2, T>(wg, t, step, MT_GEMM_FLAG_DBIAS, 0, 0, 1, 0, lds);
          } else if (k < n_w + n_d) {
            const int q = k - n_w;
            t.ti = (int16_t)(q / dj_n); t.tj = (int16_t)(q % dj_n);
            f32_gemm_tile<1, 2, T>(a.dgrad[l], t, step, MT_GEMM_FLAG_MASK, 0, 0, 1, 0, lds);
          } else {
            mt_head_reduce_body(a.head, a.slot_list, step, a.head_scratch, a.max_chunks, a.rec, by);
          }
          __syncthreads();
        }
        RES_BARRIER();
      }
      // ---- optimiser blocks, and the next step's mini-batch
      const int n = opt_items + (step + 1 < nsteps ? gather_items : 0);
      for (int k = w; k < n; k += nwg) {
        if (k < opt_items)
          optim_body<false>(a.P, a.G, a.M, a.V, (uint16_t*)nullptr, a.slot_stride, a.opt, a.slot_list, nullptr, 0, step, 1,
                            (size_t)0, k, by);
        else
          mt_gather_body<true>(a.gather, a.slot_list, a.X, a.Y, a.n_rows, a.F, a.Fpad, a.hb, step + 1, k - opt_items, by);
      }
      RES_BARRIER();
    }
  }
#undef RES_BARRIER
}

extern "C" int mt_resident_args_sizeof() { return (int)sizeof(MtResidentArgs); }

// Workgroups of the resident kernel the whole chip holds at once (tile edge 64 or 32); 0 on error.
extern "C" int mt_mlp_resident_capacity(int tile_edge) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  hipError_t e;
  if (tile_edge == 32)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mt_mlp_resident_kernel<32>, F32_THREADS, 0);
  else
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mt_mlp_resident_kernel<64>, F32_THREADS, 0);
  if (e != hipSuccess || per_cu <= 0) return 0;
  return per_cu * prop.multiProcessorCount;
}

// n_wgs workgroups (entries of args->wgs), n_groups arrival counters.  The caller guarantees n_wgs <= the capacity above,
// that every listed slot has Fpad % 8 == 0, at most 512 columns in its last hidden layer, no gradient clipping, and
// that its optimiser table was built for unsplit weight gradients.
extern "C" int mt_mlp_resident_launch(const MtResidentArgs* args, int n_wgs, int n_groups, int tile_edge,
                                      hipStream_t stream) {
  if (args == nullptr || n_wgs <= 0 || n_groups <= 0) return 0;
  if (args->n_hidden < 1 || args->n_hidden > RES_MAX_LAYERS) return -10;
  if ((args->Fpad & 7) || args->Fpad / 4 > 256 || args->Fpad < args->F) return -2;
  if (args->sync == nullptr || args->error == nullptr || args->timeout_ticks == 0ull) return -3;
  if (tile_edge != 64 && tile_edge != 32) return -7;
  if (n_wgs > mt_mlp_resident_capacity(tile_edge)) return -8;      // a group barrier needs the whole grid on the chip
  if (hipMemsetAsync(args->sync, 0, sizeof(unsigned int) * (size_t)n_groups, stream) != hipSuccess) return -9;
  if (tile_edge == 32)
    hipLaunchKernelGGL(mt_mlp_resident_kernel<32>, dim3(n_wgs), dim3(F32_THREADS), 0, stream, *args);
  else
    hipLaunchKernelGGL(mt_mlp_resident_kernel<64>, dim3(n_wgs), dim3(F32_THREADS), 0, stream, *args);
  return (int)hipGetLastError();
}
