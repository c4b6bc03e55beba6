// Shared helpers for the gfx950 kernels of libriggs_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/riggs_hip.h"

#define RIGGS_TILE 16
#define RIGGS_TILE_PIX 256
#define RIGGS_WAVE 64

namespace riggs {

void set_error(const char* fmt, ...);

#define RIGGS_HIP_CHECK(expr)                                                                  \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      riggs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

#define RIGGS_REQUIRE(cond, msg)                                   \
  do {                                                             \
    if (!(cond)) {                                                 \
      riggs::set_error("%s (%s:%d)", msg, __FILE__, __LINE__);     \
      return 2;                                                    \
    }                                                              \
  } while (0)

inline int debug_sync(int debug, hipStream_t s, const char* what) {
  if (!debug) return 0;
  hipError_t e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("debug: %s failed: %s", what, hipGetErrorString(e));
    return 1;
  }
  return 0;
}

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Function attributes (hipFuncAttributeMaxDynamicSharedMemorySize ...) are PER DEVICE: a process that drives a second GPU must
// set them there too.  `done` is a static bitset of the caller, one bit per device ordinal; true the first time on this device.
static inline bool once_per_device(unsigned long long& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return true; }
  const unsigned long long bit = 1ull << (dev & 63);
  if (done & bit) return false;
  done |= bit;
  return true;
}

// ---- library options (riggs_set_option / riggs_get_option in the ABI; process-wide, nothing is read from the environment) ----
enum OptId { OPT_FWD_WIDE_TILES = 0, OPT_FWD_WIDE_MIN, OPT_BIN_GROUPED, OPT_CNODE_BWD_ATOMICS, OPT_COLOR_SIDE_JOBS, OPT_PREPROCESS_BWD_LEAN, OPT_POSE_MLP_LAYERED, OPT_FWD_HIST_VIEW_TOL, OPT_LBS_SCALAR, OPT_COUNT };
int option(int id);

// ---- a frame's "valid" gate (include/riggs_hip.h: riggs_gate) as a kernel argument ----
struct GateArg {
  int n;
  const uint32_t* word[RIGGS_GATE_MAX];
  uint32_t mask[RIGGS_GATE_MAX];
};
static inline int gate_arg(GateArg& g, const riggs_gate* gate) {
  g.n = 0;
  for (int i = 0; i < RIGGS_GATE_MAX; i++) { g.word[i] = nullptr; g.mask[i] = 0u; }
  if (!gate) return 0;
  if (gate->n < 0 || gate->n > RIGGS_GATE_MAX) return 1;
  for (int i = 0; i < gate->n; i++) {
    if (!gate->word[i]) return 1;
    g.word[g.n] = gate->word[i]; g.mask[g.n] = gate->mask[i]; g.n++;
  }
  return 0;
}
#ifdef __HIPCC__
__device__ __forceinline__ bool gate_is_set(const GateArg& g) {
  uint32_t any = 0u;
#pragma unroll
  for (int i = 0; i < RIGGS_GATE_MAX; i++)
    if (i < g.n) any |= __hip_atomic_load(g.word[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & g.mask[i];
  return any != 0u;
}
#endif

// ---- in-library kernel timing (HIP events on the launch stream; see riggs_prof_* in the ABI) ----
enum ProfId {
  PROF_PREPROCESS_FWD = 0, PROF_DEPTH_SORT, PROF_TILE_SORT, PROF_RENDER_FWD,
  PROF_RENDER_BWD, PROF_PREPROCESS_BWD, PROF_FK_FWD, PROF_LBS_FWD, PROF_LBS_BWD, PROF_FK_BWD, PROF_KNN, PROF_POSE_FWD,
  PROF_POSE_BWD, PROF_ADAM, PROF_LOSS_FWD, PROF_LOSS_BWD, PROF_COUNT
};
void prof_begin(int id, hipStream_t s);
void prof_end(int id, hipStream_t s);
struct ProfScope {
  int id; hipStream_t s;
  ProfScope(int id_, hipStream_t s_) : id(id_), s(s_) { prof_begin(id, s); }
  ~ProfScope() { prof_end(id, s); }
};

// ---- arena layouts ---------------------------------------------------------
struct GeomLayout {
  size_t xyd, conic_o, rgb, cov3D, clamped, tiles, rect, depth_key, depth_key_sorted, order, block_tiles,
      color_job, sort_table, total;
};
GeomLayout geom_layout(int N);

struct ImageLayout {
  size_t final_T, n_contrib, ranges, final_acc, tile_max, slot_base, fwd_empty, fwd_ctr, total;
};
ImageLayout image_layout(int H, int W);

struct BinLayout {
  size_t point_list, tile_keys, ckpt, n_slots, table, work, fwd_items, walk_hist, total;
};
#define RIGGS_CKPT_FLOATS (5 * 256)  // floats per checkpoint slot
BinLayout bin_layout(int64_t cap, int N, int H, int W);

// ---- wave64 reductions -----------------------------------------------------
// Sum over the 64 lanes of a wave using DPP row operations + two cross-row steps.
__device__ __forceinline__ float wave_sum(float v) {
  // within rows of 16 lanes
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
  // lane 15 of each row now holds the row sum; combine rows
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));  // row_bcast:15
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, true));  // row_bcast:31
  return v;  // valid in lane 63
}
__device__ __forceinline__ float wave_sum_bcast(float v) {
  v = wave_sum(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

}  // namespace riggs
