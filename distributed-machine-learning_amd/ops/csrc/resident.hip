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
  int32_t sync_mode;            // 0: the barrier described at res_group_barrier; 1, 2: its measurement variants
  unsigned long long* prof;     // null, or 8 words: member 0 of group 0 adds its ticks in {forward, head, backward, optimiser,
                                // barriers} and its step count
};

// Barrier among the nwg workgroups of one group; the members may sit on different XCDs, so what they wrote has to become
// visible device-wide.  Every wave first waits for its own stores to reach the L2 (the vector cache is write-through);
// after the workgroup barrier ONE wave publishes (device-scope release: L2 write-back), arrives, waits for the last
// arrival and refreshes (device-scope acquire: the vector cache of this CU, which all waves of the workgroup share, and
// stale L2 lines are dropped).  sync_mode 1 runs the fences in every wave instead (for comparison), 2 drops them
// (measurement of the bare barrier only: results are undefined when a group spans XCDs).
__device__ __forceinline__ bool res_group_barrier(unsigned int* counter, unsigned int target, int* error,
                                                  unsigned long long timeout_ticks, int* ok_s, int sync_mode) {
  if (sync_mode == 1) __threadfence();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sync_mode != 2) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the write-back is complete before the arrival shows
    }
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    const unsigned long long t0 = wall_clock64();
    unsigned int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
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
    if (sync_mode != 2) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the invalidate is complete before the other waves go on
    }
    *ok_s = ok;
  }
  __syncthreads();
  if (sync_mode == 1) __threadfence();
  return *ok_s != 0;
}

template <int T>
__device__ __forceinline__ int res_tiles(int n) { return (n + T - 1) / T; }

template <int T, int NS>
__global__ __launch_bounds__(F32_THREADS) void mt_mlp_resident_kernel(MtResidentArgs a) {
  // the GEMM tile's NS stages; the head body's reduction arrays alias the front of it
  __shared__ __attribute__((aligned(16))) uint8_t lds[(NS * 2 * T * 128) > 8704 ? (NS * 2 * T * 128) : 8704];
  __shared__ int ok_s;
  float (*red)[512] = reinterpret_cast<float (*)[512]>(lds);
  float (*red_s)[2] = reinterpret_cast<float (*)[2]>(lds + HL_WAVES * 512 * 4);

  const MtResWg me = a.wgs[blockIdx.x];
  if (me.group < 0) return;                       // padding workgroup
  const MtResGroup grp = a.groups[me.group];
  const int w = me.member, nwg = grp.nwg;
  unsigned int* counter = a.sync + me.group;
  unsigned int target = 0u;
  const bool prof = a.prof != nullptr && me.group == 0 && w == 0;
  unsigned long long tick = prof ? wall_clock64() : 0ull, acc[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
#define RES_MARK(k)                                          \
  do {                                                       \
    if (prof) {                                              \
      const unsigned long long now_ = wall_clock64();        \
      acc[k] += now_ - tick;                                 \
      tick = now_;                                           \
    }                                                        \
  } while (0)
#define RES_BARRIER(k)                                                                                  \
  do {                                                                                                  \
    RES_MARK(k);                                                                                        \
    target += (unsigned int)nwg;                                                                        \
    if (!res_group_barrier(counter, target, a.error, a.timeout_ticks, &ok_s, a.sync_mode)) return;      \
    RES_MARK(4);                                                                                        \
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
    RES_BARRIER(3);

    for (int step = 0; step < nsteps; ++step) {
      // ---- forward
      for (int l = 0; l < L; ++l) {
        const int tj_n = res_tiles<T>(a.fwd[l][slot].J);
        const int n = res_tiles<T>(B) * tj_n;
        for (int k = w; k < n; k += nwg) {
          MtTile t;
          t.group = slot; t.ti = (int16_t)(k / tj_n); t.tj = (int16_t)(k % tj_n);
          f32_gemm_tile<0, NS, T>(a.fwd[l], t, step, MT_GEMM_FLAG_RELU, 0, 0, 1, 0, lds);
          __syncthreads();
        }
        RES_BARRIER(0);
      }
      // ---- head + loss + gradient entering the last hidden layer (partial records per 32 rows)
      for (int k = w; k < head_items; k += nwg) {
        mt_head_loss_body<1, true>(a.head, a.slot_list, step, a.head_scratch, a.max_chunks, a.rec, k, by, red, red_s);
        __syncthreads();
      }
      RES_BARRIER(1);
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
            f32_gemm_tile<2, NS, T>(wg, t, step, MT_GEMM_FLAG_DBIAS, 0, 0, 1, 0, lds);
          } else if (k < n_w + n_d) {
            const int q = k - n_w;
            t.ti = (int16_t)(q / dj_n); t.tj = (int16_t)(q % dj_n);
            f32_gemm_tile<1, NS, T>(a.dgrad[l], t, step, MT_GEMM_FLAG_MASK, 0, 0, 1, 0, lds);
          } else {
            mt_head_reduce_body(a.head, a.slot_list, step, a.head_scratch, a.max_chunks, a.rec, by);
          }
          __syncthreads();
        }
        RES_BARRIER(2);
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
      RES_BARRIER(3);
    }
    if (prof && threadIdx.x == 0) {
      for (int k = 0; k < 5; ++k) a.prof[k] += acc[k], acc[k] = 0ull;
      a.prof[5] += (unsigned long long)nsteps;
    }
  }
#undef RES_BARRIER
#undef RES_MARK
}

extern "C" int mt_resident_args_sizeof() { return (int)sizeof(MtResidentArgs); }

// tile_edge: 64 or 32 (two-stage tile pipeline), 65 or 33 (the same edges on the three-stage, one-barrier pipeline).
#define RES_DISPATCH(EDGE, CALL)                       \
  switch (EDGE) {                                      \
    case 64: { CALL(64, 2); } break;                   \
    case 32: { CALL(32, 2); } break;                   \
    case 65: { CALL(64, 3); } break;                   \
    default: { CALL(32, 3); } break;                   \
  }

static bool res_edge_ok(int e) { return e == 64 || e == 32 || e == 65 || e == 33; }

// Workgroups of the resident kernel the whole chip holds at once; 0 on error.
extern "C" int mt_mlp_resident_capacity(int tile_edge) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (!res_edge_ok(tile_edge)) return 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  hipError_t e = hipSuccess;
#define RES_OCC(T_, NS_) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mt_mlp_resident_kernel<T_, NS_>, F32_THREADS, 0)
  RES_DISPATCH(tile_edge, RES_OCC)
#undef RES_OCC
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
  if (!res_edge_ok(tile_edge)) return -7;
  if (n_wgs > mt_mlp_resident_capacity(tile_edge)) return -8;      // a group barrier needs the whole grid on the chip
  if (hipMemsetAsync(args->sync, 0, sizeof(unsigned int) * (size_t)n_groups, stream) != hipSuccess) return -9;
#define RES_GO(T_, NS_) hipLaunchKernelGGL((mt_mlp_resident_kernel<T_, NS_>), dim3(n_wgs), dim3(F32_THREADS), 0, stream, *args)
  RES_DISPATCH(tile_edge, RES_GO)
#undef RES_GO
  return (int)hipGetLastError();
}
