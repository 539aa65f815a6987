// Host-side sanitizer screen of the kernel library's launch and plan-walking code (no GPU needed).
//
// Linked against libmitune_hip_asan.so (host half built with AddressSanitizer + UBSan) and itself built with the same
// sanitizers.  It feeds the C entry points the kinds of tables the Python layer builds -- heap-allocated exactly to
// size, so a read one element past a table is an ASan report -- and checks the argument validation paths.  Kernel
// launches themselves fail cleanly on a machine without a GPU (the runtime reports "no device"); everything before
// them -- argument checks, grid arithmetic, the walk over the plan's per-layer arrays -- runs for real.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
int mt_abi_sizes(int* out);
const char* mt_source_digest();
long mt_head_scratch_floats(int n_slots, int max_rows, int max_width);
int mt_gemm_launch(int mode, const void* groups, const void* tiles, int ntiles, int step, int flags, int arow0,
                   int i_override, int ksplits, size_t split_stride, int nstage, void* stream);
int mt_gemm_f32_launch(int mode, const void* groups, const void* tiles, int ntiles, int step, int flags, int arow0,
                       int i_override, int ksplits, size_t split_stride, int tile_edge, void* stream);
int mt_gather_launch(const void* slots, const int32_t* ids, int n_slots, int max_batch, const float* X, const float* Y,
                     uint32_t n_rows, int F, int Fpad, int hb, int step, void* stream);
int mt_optim_launch(float* P, const float* G, float* M, float* V, void* Pb, size_t slot_stride, const void* hp,
                    const int32_t* ids, int n_active, int max_params, float* partials, int max_blocks, int use_clip,
                    int step, int gsplits, size_t gsplit_stride, void* stream);
int mt_mlp_train_steps(const void* plan, int step0, int nsteps, void* stream);
int mt_mlp_eval(const void* plan, int n_val_rows, int chunk_rows, void* stream);
int mt_wgrad_single_launch(const void* dY, const void* X, float* C, float* db, int M, int out, int in, int ksplits,
                           size_t split_stride, void* stream);
}

static int fails = 0;
#define EXPECT(cond)                                                  \
  do {                                                                \
    if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); ++fails; } \
  } while (0)

int main(int argc, char** argv) {
  // byte offsets of MtMlpPlan::ksplits and ::ksplits0, from the ctypes mirror of the struct (ops/lib.py)
  const size_t off_ks = argc > 1 ? (size_t)std::atol(argv[1]) : 0, off_ks0 = argc > 2 ? (size_t)std::atol(argv[2]) : 0;
  if (off_ks == 0 || off_ks0 == 0) { std::printf("usage: asan_driver <offsetof ksplits> <offsetof ksplits0>\n"); return 2; }
  int sizes[8] = {0};
  const int n = mt_abi_sizes(sizes);
  EXPECT(n == 6);
  for (int i = 0; i < n; ++i) EXPECT(sizes[i] > 0 && sizes[i] % 4 == 0);
  EXPECT(std::strlen(mt_source_digest()) == 64);

  // scratch sizing: monotone in every argument, never negative, no overflow at the largest pack the engine allows
  long prev = 0;
  for (int slots = 1; slots <= 16384; slots *= 4) {
    const long v = mt_head_scratch_floats(slots, 4096, 2048);
    EXPECT(v > prev);
    prev = v;
  }
  EXPECT(mt_head_scratch_floats(16384, 65536, 2048) > 0);

  // argument validation happens before anything is dereferenced: null tables with an empty launch are fine, bad split
  // counts / tile edges / strides are refused with their codes
  EXPECT(mt_gemm_launch(0, nullptr, nullptr, 0, 0, 0, 0, 0, 1, 0, 0, nullptr) == 0);
  EXPECT(mt_gemm_launch(0, nullptr, nullptr, 8, 0, 0, 0, 0, 2, 0, 0, nullptr) == -5);      // splits only for wgrad
  EXPECT(mt_gemm_launch(2, nullptr, nullptr, 8, 0, 0, 0, 0, 128, 0, 0, nullptr) == -5);    // split index field is 7 bits
  EXPECT(mt_gemm_f32_launch(0, nullptr, nullptr, 0, 0, 0, 0, 0, 1, 0, 128, nullptr) == 0);
  EXPECT(mt_gemm_f32_launch(0, nullptr, nullptr, 8, 0, 0, 0, 0, 1, 0, 96, nullptr) == -7);  // tile edge 128 or 64
  EXPECT(mt_gemm_f32_launch(1, nullptr, nullptr, 8, 0, 0, 0, 0, 3, 0, 64, nullptr) == -5);
  EXPECT(mt_gather_launch(nullptr, nullptr, 0, 256, nullptr, nullptr, 100, 10, 16, 4, 0, nullptr) == 0);
  EXPECT(mt_gather_launch(nullptr, nullptr, 4, 256, nullptr, nullptr, 100, 10, 12, 4, 0, nullptr) == -2);   // Fpad % 8
  EXPECT(mt_gather_launch(nullptr, nullptr, 4, 256, nullptr, nullptr, 100, 20, 16, 4, 0, nullptr) == -2);   // Fpad < F
  EXPECT(mt_optim_launch(nullptr, nullptr, nullptr, nullptr, nullptr, 64, nullptr, nullptr, 4, 1000, nullptr, 1, 0, 0, 0,
                         0, nullptr) == -6);
  EXPECT(mt_optim_launch(nullptr, nullptr, nullptr, nullptr, nullptr, 60, nullptr, nullptr, 4, 1000, nullptr, 1, 0, 0, 1,
                         0, nullptr) == -2);
  EXPECT(mt_optim_launch(nullptr, nullptr, nullptr, nullptr, nullptr, 64, nullptr, nullptr, 4, 5000, nullptr, 1, 0, 0, 1,
                         0, nullptr) == -4);     // more blocks than the partials table holds
  EXPECT(mt_wgrad_single_launch(nullptr, nullptr, nullptr, nullptr, 128, 60, 64, 1, 0, nullptr) == -2);
  EXPECT(mt_wgrad_single_launch(nullptr, nullptr, nullptr, nullptr, 0, 64, 64, 1, 0, nullptr) == 0);

  // a plan allocated exactly to size (sizes[5] bytes): depth outside 1..8 is refused before any per-layer array is read;
  // with a legal depth and no active slot every launcher returns early, so the walk over fwd/dgrad/wgrad arrays of all
  // layers runs to the end without a launch
  std::vector<uint8_t>* plan = new std::vector<uint8_t>((size_t)sizes[5], 0);
  int32_t* hdr = reinterpret_cast<int32_t*>(plan->data());
  hdr[0] = 0;
  EXPECT(mt_mlp_train_steps(plan->data(), 0, 3, nullptr) == -10);
  hdr[0] = 9;
  EXPECT(mt_mlp_train_steps(plan->data(), 0, 3, nullptr) == -10);
  EXPECT(mt_mlp_eval(plan->data(), 100, 0, nullptr) == -10);
  for (int depth = 1; depth <= 8; ++depth) {
    hdr[0] = depth;     // n_hidden
    hdr[1] = 0;         // n_active
    *reinterpret_cast<int32_t*>(plan->data() + off_ks) = 1;
    *reinterpret_cast<int32_t*>(plan->data() + off_ks0) = 1;
    EXPECT(mt_mlp_train_steps(plan->data(), 0, 5, nullptr) == 0);
    EXPECT(mt_mlp_eval(plan->data(), 1000, 256, nullptr) == 0);
  }
  delete plan;
  std::printf(fails == 0 ? "ASAN_DRIVER_OK\n" : "ASAN_DRIVER_FAILED %d\n", fails);
  return fails == 0 ? 0 : 1;
}
