// Gradient-row exchange for the frame-sharded (data-parallel) step: pack / unpack of the Gaussians that received a
// gradient this frame.
//
// The reference has no distributed path (utils/general_utils.py:207 pins cuda:0); SURVEY.md §8-e shards the per-frame
// step by frame and averages the parameter gradients.  The per-Gaussian gradients of one frame are SPARSE BY ROW: a
// Gaussian that is culled, hidden behind saturated pixels or never at alpha >= 1/255 inside a tile it overlaps gets an
// exactly zero gradient in every one of its 59 floats (93 % of the rows in the §8-d bench scene), and preprocess_bwd
// already knows which ones (it skips their arithmetic): it leaves one bit per Gaussian and a count per 256 in the
// backward workspace.  Instead of an all-reduce over N x 59 floats, a rank
//   1. PACKS its touched rows — (index, the row of every gradient tensor) — in ascending Gaussian order into one
//      segment [header | first row of every block of 256 | rows], pre-scaled by 1 / world,
//   2. all-gathers the segments (RCCL; each peer's segment travels once over the direct xGMI link: the time does not
//      grow with the number of GPUs, unlike the ring all-reduce of the dense bucket),
//   3. UNPACKS: the workgroup that owns 256 Gaussians walks the W segments IN RANK ORDER (its slice of each is found
//      through the block table, no search), overwrites a row at its first occurrence and adds the later ones — no
//      atomics, the same order of additions on every rank, so the replicas stay bit-identical.
// A segment that overflowed its capacity is flagged in its header; every rank then skips the unpack (gradients
// untouched) and reports it, so that the caller can fall back to the dense all-reduce for that step.
#include "common.h"
#include "raster_internal.h"

namespace riggs {

#define RIGGS_ROW_TENSORS 8

struct RowTensors {
  float* p[RIGGS_ROW_TENSORS];
  int w[RIGGS_ROW_TENSORS];
  int n, row_words;  // row_words = 1 (index) + sum of widths, rounded up to a multiple of 4 (16-byte rows)
};

// segment layout in 32-bit words: [0] rows stored, [1] rows needed, [2] N, [3] row_words, [4 .. 4 + nb] first row of every
// block of 256 Gaussians (nb + 1 entries, unclamped), then — 16-byte aligned — capacity rows
__host__ __device__ static inline size_t seg_rows_offset(int N) { return ((size_t)4 + (size_t)(N + 255) / 256 + 1 + 3) / 4 * 4; }

// lane's column -> (tensor, offset inside the row); column 0 is the index
__device__ inline bool column_source(const RowTensors& T, int c, const float*& base, int& width) {
  int o = c - 1;
#pragma unroll
  for (int t = 0; t < RIGGS_ROW_TENSORS; t++) {
    if (t < T.n) {
      if (o >= 0 && o < T.w[t]) { base = T.p[t] + o; width = T.w[t]; return true; }
      o -= T.w[t];
    }
  }
  return false;
}

#define RIGGS_SEG_INVALID 0xFFFFFFFFu  // "rows needed" of a segment whose frame was invalid (riggs_grad_rows_pack_gated)

__global__ __launch_bounds__(256) void rows_pack_kernel(int N, const unsigned long long* __restrict__ bits,
                                                        const uint32_t* __restrict__ block_touched, RowTensors T, float scale,
                                                        int capacity, uint32_t* __restrict__ seg, GateArg gate) {
  __shared__ uint32_t s_part[4];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nb = (N + 255) / 256;
  if (gate.n > 0 && gate_is_set(gate)) {
    // this rank's frame is invalid (NaN pose, truncated lists): nothing of it travels; the header says so to every rank
    if (t == 0) {
      seg[4 + b] = 0u;
      if (b == nb - 1) { seg[4 + nb] = 0u; seg[0] = 0u; seg[1] = RIGGS_SEG_INVALID; seg[2] = (uint32_t)N; seg[3] = (uint32_t)T.row_words; }
    }
    return;
  }
  // rows of the blocks before this one (<= 1172 counts at 300k: a few loads per thread)
  uint32_t sum = 0;
  for (int i = t; i < b; i += 256) sum += block_touched[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) s_part[wave] = sum;
  __syncthreads();
  const uint32_t prefix = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  if (t == 0) {
    seg[4 + b] = prefix;
    if (b == nb - 1) {
      const uint32_t need = prefix + block_touched[b];
      seg[4 + nb] = need;
      seg[0] = need < (uint32_t)capacity ? need : (uint32_t)capacity;
      seg[1] = need; seg[2] = (uint32_t)N; seg[3] = (uint32_t)T.row_words;
    }
  }
  const int nwords = (N + 63) / 64;
  uint32_t base = prefix;
  unsigned long long mine = 0ull;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const int word = b * 4 + w;
    const unsigned long long m = word < nwords ? bits[word] : 0ull;
    if (w < wave) base += (uint32_t)__builtin_popcountll(m);
    if (w == wave) mine = m;
  }
  float* rows = reinterpret_cast<float*>(seg + seg_rows_offset(N));
  for (int c = lane; c < T.row_words; c += 64) {
    const float* src = nullptr;
    int width = 0;
    const bool has = column_source(T, c, src, width);
    unsigned long long m = mine;
    uint32_t row = base;
    while (m) {
      const int bit = __builtin_ctzll(m);
      m &= m - 1ull;
      const int g = b * 256 + wave * 64 + bit;
      if (row < (uint32_t)capacity) {
        float v = 0.f;
        if (c == 0) v = __int_as_float(g);
        else if (has) v = src[(size_t)g * width] * scale;
        rows[(size_t)row * T.row_words + c] = v;
      }
      row++;
    }
  }
}

// Unpack.  Workgroup = the 256 Gaussians of one block of the segments' tables, 16 groups of 16 lanes; a group combines one
// row per step, lane k holding floats 4k .. 4k+3 of it (rows are 16-byte aligned multiples of 4 floats).  The sums are
// built in LDS — a slot of row_words floats per Gaussian that occurs in some segment — so that the chain per segment is
// LDS traffic only: the row a group needs from the NEXT segment is already in flight while it combines this one's.
// RIGGS_UNPACK_SLOTS slots (23 KB at 60 floats: every workgroup of a 300k launch is resident at once); the Gaussians
// beyond that (a block where more than 96 of 256 are touched) are combined in place in HBM, same order.  Segments
// are processed in rank order with a barrier between them (a Gaussian occurs at most once per segment, so inside a
// segment the groups never meet), and the finished slots are stored once at the end.
#define RIGGS_UNPACK_MAX_WORLD 64
#define RIGGS_UNPACK_SLOTS 96
__device__ inline void unpack_store4(const RowTensors& T, int g, int c0, const float4 v) {
  const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float* base = nullptr;
    int width = 0;
    if (column_source(T, c0 + j, base, width)) const_cast<float*>(base)[(size_t)g * width] = e[j];
  }
}
__device__ inline float4 unpack_load4(const RowTensors& T, int g, int c0) {
  float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float* base = nullptr;
    int width = 0;
    if (column_source(T, c0 + j, base, width)) e[j] = base[(size_t)g * width];
  }
  return make_float4(e[0], e[1], e[2], e[3]);
}

__global__ __launch_bounds__(256) void rows_unpack_kernel(int N, int world, int capacity, size_t seg_words,
                                                          const uint32_t* __restrict__ all, RowTensors T,
                                                          uint32_t* __restrict__ status,
                                                          unsigned long long* __restrict__ touched_bits) {
  extern __shared__ float s_acc[];                     // [RIGGS_UNPACK_SLOTS][row_words]
  __shared__ unsigned short s_slot[256];               // Gaussian (local) -> slot, 0xFFFF = none yet
  __shared__ unsigned short s_local[256];              // slot -> Gaussian (local)
  __shared__ uint32_t s_range[2 * RIGGS_UNPACK_MAX_WORLD];
  __shared__ uint32_t s_hdr[2];
  __shared__ uint32_t s_count;
  const int b = blockIdx.x, t = threadIdx.x, sub = t & 15, group = t >> 4;
  const int RW = T.row_words;
  if (t == 0) { s_hdr[0] = 0u; s_hdr[1] = 0u; s_count = 0u; }
  s_slot[t] = 0xFFFFu;
  __syncthreads();
  if (t < world) {  // every segment's header and this block's slice of it, all loads in flight at once
    const uint32_t* seg = all + (size_t)t * seg_words;
    const uint32_t need = seg[1];
    const bool invalid = need == RIGGS_SEG_INVALID;   // that rank's frame was invalid: the step is skipped everywhere
    const bool bad = !invalid && (need > (uint32_t)capacity || seg[2] != (uint32_t)N || seg[3] != (uint32_t)RW);
    const uint32_t first = seg[4 + b], last = seg[4 + b + 1];
    s_range[2 * t] = first; s_range[2 * t + 1] = last < (uint32_t)capacity ? last : (uint32_t)capacity;
    if (!invalid) atomicMax(&s_hdr[0], need);
    if (bad) atomicOr(&s_hdr[1], 1u);
    if (invalid) atomicOr(&s_hdr[1], 2u);
  }
  __syncthreads();
  // STICKY status (cleared by the host when it reads it, SparseRowExchange.check()): a caller that polls every k steps must
  // still learn that ONE of them overflowed — on that step every rank skipped the unpack and stepped its optimizer on local
  // gradients.  {largest need since the last check, overflow flag, unpack calls since the last check, the call (1-based)
  // that overflowed first}
  // status[4] is NOT sticky: this call's flag, rewritten by every call — the word the optimizers of this step gate on (gated on
  // the sticky word they would also skip every GOOD step up to the host's next look)
  if (b == 0 && t == 0) {
    atomicMax(&status[0], s_hdr[0]);
    const uint32_t call = atomicAdd(&status[2], 1u) + 1u;
    if (s_hdr[1]) { atomicOr(&status[1], s_hdr[1]); atomicCAS(&status[3], 0u, call); }
    status[4] = s_hdr[1];
  }
  if (s_hdr[1]) return;  // (every workgroup of every rank sees the same headers: all skip, the gradients stay as they were)
  const size_t rows_off = seg_rows_offset(N);
  const int c0 = 4 * sub;                              // this lane's floats of a row (rows wider than 64 floats loop)
  // the group's first row of segment 0 (index word + this lane's floats), loaded ahead
  float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
  int nxt_g = -1;
  {
    const uint32_t row = s_range[0] + group;
    if (row < s_range[1]) {
      const float* src = reinterpret_cast<const float*>(all + rows_off) + (size_t)row * RW;
      nxt_g = __float_as_int(src[0]);
      if (c0 < RW) nxt = *reinterpret_cast<const float4*>(src + c0);
    }
  }
  for (int r = 0; r < world; r++) {
    const uint32_t first = s_range[2 * r], last = s_range[2 * r + 1];
    const float* rows = reinterpret_cast<const float*>(all + (size_t)r * seg_words + rows_off);
    float4 cur = nxt;
    int cur_g = nxt_g;
    nxt_g = -1;
    if (r + 1 < world) {                               // next segment's first row: in flight across this one's work
      const uint32_t row = s_range[2 * r + 2] + group;
      if (row < s_range[2 * r + 3]) {
        const float* src = reinterpret_cast<const float*>(all + (size_t)(r + 1) * seg_words + rows_off) + (size_t)row * RW;
        nxt_g = __float_as_int(src[0]);
        if (c0 < RW) nxt = *reinterpret_cast<const float4*>(src + c0);
      }
    }
    for (uint32_t row = first + group; row < last; row += 16) {
      const float* src = rows + (size_t)row * RW;
      if (row != first + group) {
        cur_g = __float_as_int(src[0]);
        if (c0 < RW) cur = *reinterpret_cast<const float4*>(src + c0);
      }
      const int g = cur_g, l = g - b * 256;
      if (l < 0 || l >= 256 || g >= N) continue;       // (cannot happen with segments this library packed)
      unsigned slot = s_slot[l];
      const bool seen = slot != 0xFFFFu;               // (set by an EARLIER segment: a barrier lies in between)
      if (!seen) {
        if (sub == 0) { slot = atomicAdd(&s_count, 1u); s_slot[l] = (unsigned short)slot; s_local[slot] = (unsigned short)l; }
        slot = __shfl(slot, 0, 16);
      }
      for (int c = c0; c < RW; c += 64) {
        const float4 v = c == c0 ? cur : *reinterpret_cast<const float4*>(src + c);
        if (slot < RIGGS_UNPACK_SLOTS) {
          float4* a = reinterpret_cast<float4*>(s_acc + (size_t)slot * RW + c);
          if (seen) { float4 o = *a; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; *a = o; }
          else *a = v;
        } else {                                       // no LDS slot left: combine in place (same order, same additions)
          float4 o = v;
          if (seen) { const float4 d = unpack_load4(T, g, c); o.x = d.x + v.x; o.y = d.y + v.y; o.z = d.z + v.z; o.w = d.w + v.w; }
          unpack_store4(T, g, c, o);
        }
      }
    }
    __syncthreads();
  }
  // the rows this workgroup has written become the workspace's "rows that hold a gradient" (a superset of this rank's
  // own): the next riggs_raster_backward with cfg.sparse_zero zeroes the ones it does not touch itself
  if (touched_bits) {
    const unsigned long long wrote = __builtin_amdgcn_ballot_w64(s_slot[t] != 0xFFFFu);
    if ((t & 63) == 0 && b * 256 + (t >> 6) * 64 < N) touched_bits[b * 4 + (t >> 6)] = wrote;
  }
  // store: one slot per group and step; lane k scatters floats 4k .. 4k+3 to their tensors
  const int n_slots = min((int)s_count, RIGGS_UNPACK_SLOTS);
  for (int slot = group; slot < n_slots; slot += 16) {
    const int g = b * 256 + (int)s_local[slot];
    for (int c = c0; c < RW; c += 64) unpack_store4(T, g, c, *reinterpret_cast<const float4*>(s_acc + (size_t)slot * RW + c));
  }
}

// 1.0f / 0.0f into a slot of the buffer a dense all-reduce is about to sum: afterwards the slot is non-zero on EVERY rank when
// some rank's frame was invalid — the word the ranks' optimizers are gated on, so that all replicas skip that step together
__global__ void gate_flag_kernel(GateArg gate, float* flag) { flag[0] = gate_is_set(gate) ? 1.0f : 0.0f; }

static int fill_tensors(RowTensors& T, int n, float* const* grads, const int32_t* widths) {
  RIGGS_REQUIRE(n >= 1 && n <= RIGGS_ROW_TENSORS, "1..8 gradient tensors");
  int words = 1;
  for (int i = 0; i < RIGGS_ROW_TENSORS; i++) { T.p[i] = nullptr; T.w[i] = 0; }
  for (int i = 0; i < n; i++) {
    RIGGS_REQUIRE(grads[i] != nullptr && widths[i] >= 1, "gradient tensor missing / empty rows");
    T.p[i] = grads[i]; T.w[i] = widths[i]; words += widths[i];
  }
  T.n = n;
  T.row_words = (words + 3) / 4 * 4;
  return 0;
}


// ---- multi-GPU readiness on ONE GPU: a stand-in for the compute units a collective library's channel kernels hold while a
// rank's frame runs beside them (W = 8: RCCL keeps 16-64 workgroups resident for the length of a collective).  n workgroups,
// each claiming a whole CU's LDS (so they land on n distinct compute units), spin — a read of the stop word per ~2 us, nothing
// else — until the host raises `stop` or `max_ticks` of the 100 MHz wall clock have passed (bounded: it cannot hang the device).
__global__ __launch_bounds__(64) void cu_pin_kernel(const int32_t* stop, unsigned long long max_ticks, uint32_t* started) {
  extern __shared__ char pin_lds[];
  if (threadIdx.x == 0) {
    pin_lds[0] = 1;  // (the allocation is what matters)
    atomicAdd(started, 1u);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && wall_clock64() - t0 < max_ticks)
      __builtin_amdgcn_s_sleep(127);
  }
}

}  // namespace riggs

using namespace riggs;

extern "C" {

int32_t riggs_grad_rows_row_floats(int32_t n_tensors, const int32_t* widths) {
  int words = 1;
  for (int i = 0; i < n_tensors; i++) words += widths[i];
  return (words + 3) / 4 * 4;
}

size_t riggs_grad_rows_segment_bytes(int32_t N, int32_t row_floats, int32_t capacity) {
  return align_up((seg_rows_offset(N > 0 ? N : 1) + (size_t)(capacity > 0 ? capacity : 0) * (size_t)row_floats) * 4);
}

int riggs_gate_flag(const riggs_gate* gate, float* flag, riggs_stream stream_) {
  RIGGS_REQUIRE(flag != nullptr, "riggs_gate_flag: flag is NULL");
  GateArg g;
  RIGGS_REQUIRE(gate_arg(g, gate) == 0, "riggs_gate: 0..4 non-NULL words");
  hipLaunchKernelGGL(gate_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, g, flag);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_grad_rows_pack_gated(int32_t N, const void* backward_workspace, int32_t n_tensors, const float* const* grads,
                               const int32_t* widths, float scale, int32_t capacity, void* segment, const riggs_gate* gate,
                               riggs_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  RIGGS_REQUIRE(N > 0 && backward_workspace && segment && capacity >= 0, "riggs_grad_rows_pack: bad arguments");
  RowTensors T;
  int rc = fill_tensors(T, n_tensors, const_cast<float* const*>(grads), widths);
  if (rc) return rc;
  GateArg g;
  RIGGS_REQUIRE(gate_arg(g, gate) == 0, "riggs_gate: 0..4 non-NULL words");
  const char* ws = (const char*)backward_workspace;
  hipLaunchKernelGGL(rows_pack_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N,
                     (const unsigned long long*)(ws + ws_bits_offset(N)), (const uint32_t*)(ws + ws_blocks_offset(N)), T, scale,
                     capacity, (uint32_t*)segment, g);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_grad_rows_pack(int32_t N, const void* backward_workspace, int32_t n_tensors, const float* const* grads,
                         const int32_t* widths, float scale, int32_t capacity, void* segment, riggs_stream stream_) {
  return riggs_grad_rows_pack_gated(N, backward_workspace, n_tensors, grads, widths, scale, capacity, segment, nullptr, stream_);
}

int riggs_grad_rows_unpack(int32_t N, int32_t world, int32_t capacity, const void* segments, int32_t n_tensors,
                           float* const* grads, const int32_t* widths, uint32_t* status, void* backward_workspace,
                           riggs_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  RIGGS_REQUIRE(N > 0 && world >= 1 && segments && status && capacity >= 0, "riggs_grad_rows_unpack: bad arguments");
  RowTensors T;
  int rc = fill_tensors(T, n_tensors, grads, widths);
  if (rc) return rc;
  const size_t seg_words = riggs_grad_rows_segment_bytes(N, T.row_words, capacity) / 4;
  RIGGS_REQUIRE(world <= RIGGS_UNPACK_MAX_WORLD, "riggs_grad_rows_unpack: at most 64 segments");
  const size_t lds = (size_t)RIGGS_UNPACK_SLOTS * T.row_words * sizeof(float);
  RIGGS_REQUIRE(lds <= 60 * 1024, "riggs_grad_rows_unpack: rows wider than 160 floats");
  unsigned long long* bits = backward_workspace ? (unsigned long long*)((char*)backward_workspace + ws_bits_offset(N)) : nullptr;
  hipLaunchKernelGGL(rows_unpack_kernel, dim3((N + 255) / 256), dim3(256), lds, s, N, world, capacity, seg_words,
                     (const uint32_t*)segments, T, status, bits);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_debug_pin_cus(int32_t n_cus, const int32_t* stop_flag, uint32_t max_ms, uint32_t* started, riggs_stream stream_) {
  RIGGS_REQUIRE(n_cus >= 1 && n_cus <= 1024 && stop_flag && started && max_ms >= 1 && max_ms <= 600000, "riggs_debug_pin_cus: bad arguments");
  static unsigned long long attr_done = 0ull;
  const int lds = 160 * 1024;
  if (once_per_device(attr_done))
    RIGGS_HIP_CHECK(hipFuncSetAttribute((const void*)cu_pin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(cu_pin_kernel, dim3(n_cus), dim3(64), lds, (hipStream_t)stream_, stop_flag, (unsigned long long)max_ms * 100000ull, started);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
