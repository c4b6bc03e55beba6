// PoseMLP (time -> per-joint quaternions + root translation) as three launches instead of
// ~60 torch ops: /root/reference/skeleton_utils/network_utils.py:115-150 with the positional
// embedding of utils/time_utils.py:208-256 (input_dims 1, [t, sin(2^k t), cos(2^k t)]).
//
// The network is ONE row wide (batch 1): every layer is a GEMV over a <= 273 x 256 weight
// matrix that lives in L2 after the first iteration, so the job is pure latency.  One
// 1024-thread workgroup (16 wave64) walks the layers: a wave owns output rows, its lanes
// stride the row (coalesced 256-B reads), DPP reduces the 64 partial sums.  The backward chain
// (dz_l, dh_l) runs the same way with thread-per-column accumulation (coalesced along rows);
// the 0.5 M weight gradients are outer products written by a wide second kernel.
#include "fk_device.h"

namespace riggs {

#define PM_MAX_LAYERS 12
#define PM_MAX_W 256
#define PM_MAX_HEAD 320  // rows of the two heads together (4 J + 3): J <= 79 in the one-launch kernels
#define PM_MAX_IN 320
#define PM_MAX_EMB 64
#define PM_RPW 8       // rows per wave per pass (forward GEMV)
#define PM_NG 16       // row groups (= waves) of the transposed GEMV in the backward chain
#define PM_CPL 5       // columns per lane: ceil(PM_MAX_IN / 64)

struct PoseMlpDesc {
  int depth;        // hidden layers (8)
  int width;        // hidden width (<= 256)
  int multires;     // embedding frequencies (8) -> emb = 1 + 2*multires
  int skip;         // after layer `skip` the embedding is concatenated IN FRONT of h (network_utils.py:144-145)
  int n_rot;        // rotation head outputs (4 J)
  const float* W[PM_MAX_LAYERS];  // hidden layers, row-major (out, in)
  const float* b[PM_MAX_LAYERS];
  const float *W_rot, *b_rot, *W_tr, *b_tr;
};

__device__ __forceinline__ int pm_in_dim(const PoseMlpDesc& d, int l, int emb) {
  if (l == 0) return emb;
  return (l - 1 == d.skip) ? d.width + emb : d.width;
}

// acts layout (floats): [0, emb) embedding, then per layer l: width post-ReLU activations.
// A single workgroup can pull only ~25 GB/s through its CU, and the 2 MB of weights are evicted
// from L2 by the streaming kernels of every iteration, so each layer is spread over 64+
// workgroups (one wave per output row) and the layers are separate graph nodes.

// input element i of layer l (embedding / previous activations / [emb, h] after the skip layer)
__device__ __forceinline__ float pm_input(const PoseMlpDesc& d, const float* __restrict__ acts, int l, int emb, int i) {
  if (l == 0) return acts[i];
  if (l - 1 == d.skip) return (i < emb) ? acts[i] : acts[emb + (size_t)(l - 1) * d.width + (i - emb)];
  return acts[emb + (size_t)(l - 1) * d.width + i];
}

__global__ __launch_bounds__(64) void pm_embed_kernel(PoseMlpDesc d, const float* __restrict__ t, float* __restrict__ acts) {
  const int tid = threadIdx.x, emb = 1 + 2 * d.multires;
  if (tid < emb) {
    const float tv = t[0];
    float v = tv;
    if (tid > 0) {
      const int k = (tid - 1) >> 1;
      const float f = (float)(1 << k);
      v = ((tid - 1) & 1) ? cosf(tv * f) : sinf(tv * f);
    }
    acts[tid] = v;
  }
}

// layer l (l == depth: the two heads): one wave64 per output row
__global__ __launch_bounds__(256) void pm_layer_kernel(PoseMlpDesc d, int l, float* __restrict__ acts,
                                                       const float* __restrict__ rot_bias4,
                                                       float* __restrict__ rotation, float* __restrict__ translation) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int emb = 1 + 2 * d.multires;
  const bool heads = (l == d.depth);
  const int n_out = heads ? d.n_rot + 3 : d.width;
  const int r = blockIdx.x * 4 + wave;
  if (r >= n_out) return;
  const int in_dim = pm_in_dim(d, l, emb);  // for l == depth this is the heads' input width
  const float* row;
  float bias;
  if (!heads) { row = d.W[l] + (size_t)r * in_dim; bias = d.b[l][r]; }
  else if (r < d.n_rot) { row = d.W_rot + (size_t)r * in_dim; bias = d.b_rot[r]; }
  else { row = d.W_tr + (size_t)(r - d.n_rot) * in_dim; bias = d.b_tr[r - d.n_rot]; }
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < PM_CPL; k++) {
    const int i = lane + 64 * k;
    if (i < in_dim) acc += row[i] * pm_input(d, acts, l, emb, i);
  }
  acc = wave_sum(acc);
  if (lane == 63) {
    const float v = acc + bias;
    if (!heads) acts[emb + (size_t)l * d.width + r] = fmaxf(v, 0.f);
    else if (r < d.n_rot) rotation[r] = rot_bias4 ? v + rot_bias4[r & 3] : v;
    else translation[r - d.n_rot] = v;
  }
}

// Backward step for the consumer `l` (l == depth: heads, else hidden layer l):
//   v = dz of the consumer (for the heads: the incoming output gradients),
//   dh_in[c] = sum_r M[r][c] v[r]   for the consumer's input columns c.
// Each block owns 64 columns; its 4 waves split the rows (coalesced 256-B row segments), LDS combines.
// For hidden layers v = dz_l = dh_l * relu'(h_l) is computed by every block (block 0 stores it for the
// weight-gradient kernel).  dh buffers: dh[l] holds the gradient w.r.t. the INPUT of consumer l.
#define PM_RG 8  // row groups of the transposed GEMV (grid.y)
__global__ __launch_bounds__(256) void pm_backward_step_kernel(PoseMlpDesc d, int l, const float* __restrict__ acts,
                                                               const float* __restrict__ g_rot,
                                                               const float* __restrict__ g_tr,
                                                               const float* __restrict__ dh_out /* input-grad of consumer l+1 */,
                                                               float* __restrict__ dh_in /* zeroed */, float* __restrict__ dzs) {
  __shared__ float s_part[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int emb = 1 + 2 * d.multires;
  const bool heads = (l == d.depth);
  const int n_rows = heads ? d.n_rot + 3 : d.width;
  const int in_dim = pm_in_dim(d, l, emb);
  // dh of h_l lives in the consumer (l+1)'s input gradient, behind the embedding when that input was [emb, h]
  const int off = (l == d.skip) ? emb : 0;
  const int c = blockIdx.x * 64 + lane;
  const int rstep = 4 * PM_RG, r0 = blockIdx.y * 4 + wave;
  float acc = 0.f;
#pragma unroll 4
  for (int r = r0; r < n_rows; r += rstep) {
    float v;
    const float* row;
    if (heads) {
      v = (r < d.n_rot) ? g_rot[r] : g_tr[r - d.n_rot];
      row = (r < d.n_rot) ? d.W_rot + (size_t)r * in_dim : d.W_tr + (size_t)(r - d.n_rot) * in_dim;
    } else {
      const float h = acts[emb + (size_t)l * d.width + r];
      v = (h > 0.f) ? dh_out[off + r] : 0.f;  // dz_l
      row = d.W[l] + (size_t)r * in_dim;
      if (blockIdx.x == 0 && lane == 0) dzs[(size_t)l * d.width + r] = v;
    }
    if (l > 0 && c < in_dim) acc += row[c] * v;
  }
  if (l == 0) return;  // the embedding has no trainable input
  s_part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < in_dim) atomicAdd(&dh_in[c], s_part[0][lane] + s_part[1][lane] + s_part[2][lane] + s_part[3][lane]);
}

// Weight / bias gradients: one block per (parameter matrix, row): dW[r][:] = v[r] * input[:], db[r] = v[r].
// Flat layout: [W_0, b_0, ..., W_{depth-1}, b_{depth-1}, W_rot, b_rot, W_tr, b_tr]
struct PoseMlpGradDesc {
  int64_t w_off[PM_MAX_LAYERS + 2];  // offset of W block (hidden layers, rot head, tr head)
  int64_t b_off[PM_MAX_LAYERS + 2];
  int row_start[PM_MAX_LAYERS + 3];  // prefix of rows over the depth + 2 matrices
};
__global__ __launch_bounds__(64) void pm_backward_weights_kernel(PoseMlpDesc d, PoseMlpGradDesc g,
                                                                 const float* __restrict__ acts,
                                                                 const float* __restrict__ dzs,
                                                                 const float* __restrict__ g_rot,
                                                                 const float* __restrict__ g_tr,
                                                                 float* __restrict__ flat) {
  const int emb = 1 + 2 * d.multires;
  const int nmat = d.depth + 2;
  int m = 0;
  while (m + 1 < nmat && (int)blockIdx.x >= g.row_start[m + 1]) m++;
  const int r = blockIdx.x - g.row_start[m];
  const int l = (m < d.depth) ? m : d.depth;  // input of the heads = input index `depth`
  const int in_dim = pm_in_dim(d, l, emb);
  const float v = (m < d.depth) ? dzs[(size_t)m * d.width + r] : (m == d.depth ? g_rot[r] : g_tr[r]);
  float* out = flat + g.w_off[m] + (size_t)r * in_dim;
  for (int c = threadIdx.x; c < in_dim; c += 64) out[c] = v * pm_input(d, acts, l, emb, c);
  if (threadIdx.x == 0) flat[g.b_off[m] + r] = v;
}


// =====================================================================================================
// One-launch variants.  The layered kernels above cost one graph node (~4.5 us on MI355X, whatever the
// work) per layer: 10 forward + 11 backward nodes = 125 us of a 670 us frame.  Here the whole chain
// runs in ONE launch of width/8 workgroups (8 wave64 each, all co-resident): every wave keeps its row
// (forward) / column (backward) of every weight matrix in registers — loaded once, up front, all loads
// in flight together — and the 256-float vector that each layer hands to the next travels between
// workgroups as 8-byte {tag, value} granules written with one agent-scope (write-through) store and
// polled with agent-scope loads: the data is its own flag, so no fence and no counter (guide:
// "R2 granules", valid for <= 4 KB hand-offs on gfx950 whatever the workgroup -> XCD placement).
// tag = stage index + 1; the granule array is zeroed by a memset node in front of every launch, so a
// replayed hipGraph never sees tags of the previous replay.  Every spin is bounded: on time-out the error
// word is set and the outputs are poisoned with NaN instead of hanging the queue.
// =====================================================================================================
#define PMF_WAVES 8
// optional stage timestamps of workgroup 0 (riggs_pose_mlp_set_trace; 100 MHz wall clock)
#define PM_TRACE(i) do { if (trace && blockIdx.x == 0 && threadIdx.x == 0) trace[i] = wall_clock64(); } while (0)
#define PMF_SPIN_MAX (1u << 17)
typedef __attribute__((address_space(1))) unsigned long long pm_gu64;

// `local`: every workgroup of the chain runs on the same XCD (pm_chain_is_local): a plain 8-byte store lands in the L2 they
// share, where the consumers' agent-scope loads find it — half the latency of the write-through store (tools/scratch/
// pingpong.hip: 0.52 us round trip against 1.10 us), which takes the value to the memory side and drops the line from L2.
__device__ __forceinline__ void pm_store_granule(unsigned long long* g, uint32_t tag, float value, bool local) {
  const unsigned long long x = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(value);
  if (local) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(g), "v"(x) : "memory");
  else __hip_atomic_store((pm_gu64*)g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one wave re-reads its <= 4 granules per lane until every tag matches (wave-uniform exit).  A granule
// load is a round trip to the memory side (~1-2 us: the producers' write-through stores drop the line
// from L2), so several sweeps are kept in flight and checked oldest first (pm_sweep_n).
__device__ __forceinline__ void pm_sweep_issue(pm_gu64* g, int n, int lane, unsigned long long (&x)[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int r = lane + 64 * k;
    x[k] = (r < n) ? __hip_atomic_load(g + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  }
}
__device__ __forceinline__ bool pm_sweep_check(const unsigned long long (&x)[4], int n, int lane, uint32_t tag,
                                               float (&v)[4]) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    v[k] = __uint_as_float((uint32_t)x[k]);
    if (lane + 64 * k < n) ok &= ((uint32_t)(x[k] >> 32) == tag);
  }
  return __all(ok);
}
// NQ sweeps in flight: 4 when the granules come from the memory side (a load is 1-2 us: the hand-off is seen one load latency
// after it lands instead of up to two), 2 when the chain shares an L2 (pm_chain_is_local: the loads are 0.3 us, and what a
// hand-off then waits for is the queue of 32 pollers x 16 lines at that L2 — forward 15.8 -> 13.2 us, backward 19.0 -> 14.2 us
// on workgroup 0's stage stamps, tools/pose_mlp_trace.py)
template <int NQ>
__device__ __forceinline__ bool pm_sweep_n(const unsigned long long* gran, int n, uint32_t tag, float (&v)[4],
                                           uint32_t* err, uint32_t* sticky, int lane) {
  pm_gu64* g = (pm_gu64*)gran;
  unsigned long long x[NQ][4];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    pm_sweep_issue(g, n, lane, x[q]);
    __builtin_amdgcn_s_sleep(2);
  }
  for (uint32_t spins = 0;; spins++) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      if (pm_sweep_check(x[q], n, lane, tag, v)) return true;
      pm_sweep_issue(g, n, lane, x[q]);
    }
    if (spins > PMF_SPIN_MAX / NQ) {
      // per-call word (poisons this launch's outputs) and, when the caller keeps a persistent sync_state, its
      // STICKY status word: never cleared by a kernel, read by the host (riggs_pose_mlp_status_word)
      if (lane == 0) { atomicOr(err, 1u); if (sticky) atomicOr(sticky, 1u); }
      return false;
    }
  }
}
__device__ __forceinline__ bool pm_sweep(const unsigned long long* gran, int n, uint32_t tag, float (&v)[4],
                                         uint32_t* err, uint32_t* sticky, int lane, bool local = false) {
  return local ? pm_sweep_n<2>(gran, n, tag, v, err, sticky, lane) : pm_sweep_n<4>(gran, n, tag, v, err, sticky, lane);
}

// Placement of the chain.  The hardware deals a launch's workgroups round-robin over the eight XCDs (workgroup b -> XCD b % 8),
// so the chain's workgroups are the ones whose index is a multiple of eight: all on one XCD, one L2 — when one XCD can hold
// them all at once (pm_chain_stride; else they are the launch's first workgroups, on all XCDs).  Nothing RELIES on the
// placement: every chain workgroup publishes the XCD it finds itself on (hardware register XCC_ID) with the always-correct
// write-through granule, and only if all of them report the same one are the layer hand-offs stored the cheap way
// (pm_store_granule).
#define PM_CHAIN_STRIDE 8
#define PM_MAX_CHAIN 64   // chain workgroups (granules of the placement check)
// (`by`: the wave that does it — the backward's first wave is busy with the kinematic chain at that time)
__device__ __forceinline__ bool pm_chain_is_local(unsigned long long* xg, int rank, int n_chain, uint32_t tag, uint32_t* err,
                                                  uint32_t* sticky, int lane, int wave, int* s_word, int by = 0) {
  if (wave == by) {
    const uint32_t mine = __builtin_amdgcn_s_getreg(63508) & 0xFu;  // HW_REG_XCC_ID
    if (lane == 0) pm_store_granule(xg + rank, tag, __uint_as_float(mine), false);
    float v[4];
    const bool ok = pm_sweep(xg, n_chain, tag, v, err, sticky, lane);
    const bool same = ok && (lane >= n_chain || __float_as_uint(v[0]) == mine);  // (n_chain <= 64: one granule per lane)
    const bool all_same = __all(same);
    if (lane == 0) *s_word = all_same ? 1 : 0;
  }
  __syncthreads();
  return *s_word != 0;
}

#define PM_FAULT_BIT 1u
__global__ __launch_bounds__(PMF_WAVES * 64) void pm_forward_fused_kernel(PoseMlpDesc d, const float* __restrict__ t,
                                                                          const float* __restrict__ rot_bias4,
                                                                          float* __restrict__ acts,
                                                                          unsigned long long* gran, uint32_t* gen,
                                                                          uint32_t* sticky, uint32_t* bwd_state, int bwd_words,
                                                                          float* __restrict__ rotation,
                                                                          float* __restrict__ translation,
                                                                          float* __restrict__ wt, int n_chain, int stride,
                                                                          unsigned long long* xcc_gran,
                                                                          unsigned long long* trace) {
  __shared__ float s_in[PM_MAX_IN];
  __shared__ int s_failed, s_local;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rank = (int)blockIdx.x / stride;  // (chain workgroups: index in the chain)
  if ((int)blockIdx.x % stride != 0 || rank >= n_chain) {
    // ---- transposer workgroups (idle CUs, off the chain): WT_l[c][r] = M_l[r][off_l + c] for the backward,
    // whose waves own COLUMNS: read in place a column costs 64 cache lines per load and ~12 us of per-CU
    // miss latency at the head of that kernel; from WT it is one coalesced row like the forward's.
    __shared__ float s_t[32][33];
    const int emb_ = 1 + 2 * d.multires;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 16
    const int tiles_per = (PM_MAX_HEAD / 32) * (PM_MAX_W / 32);  // (rows: up to PM_MAX_HEAD for the heads) x (columns)
    const int n_side = (int)gridDim.x - n_chain, side = (int)blockIdx.x - min(n_chain, (int)blockIdx.x / stride + 1);
    for (int tile = side; tile < d.depth * tiles_per; tile += n_side) {
      const int l = 1 + tile / tiles_per, tt = tile % tiles_per;
      const int r0 = (tt / (PM_MAX_W / 32)) * 32, c0 = (tt % (PM_MAX_W / 32)) * 32;
      const bool heads = (l == d.depth);
      const int in_dim = pm_in_dim(d, l, emb_);
      const int off = (l - 1 == d.skip) ? emb_ : 0;
      const int n_rows = heads ? d.n_rot + 3 : d.width;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int r = r0 + ty + 16 * h, c = c0 + tx;
        float v = 0.f;
        if (r < n_rows && c < d.width) {
          const float* rp = !heads ? d.W[l] + (size_t)r * in_dim
                                   : (r < d.n_rot ? d.W_rot + (size_t)r * in_dim : d.W_tr + (size_t)(r - d.n_rot) * in_dim);
          v = rp[off + c];
        }
        s_t[ty + 16 * h][tx] = v;
      }
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 2; h++)
        wt[((size_t)(l - 1) * PM_MAX_W + c0 + ty + 16 * h) * PM_MAX_HEAD + r0 + tx] = s_t[tx][ty + 16 * h];
      __syncthreads();
    }
    return;
  }
  const int row = rank * PMF_WAVES + wave;
  const int emb = 1 + 2 * d.multires;
  uint32_t* err = bwd_state;  // {err, gen, pad, pad, backward granules...}: cleared here for the backward launch
  for (int i = rank * blockDim.x + threadIdx.x; i < bwd_words; i += n_chain * blockDim.x) bwd_state[i] = 0u;
  // `gen` (persistent across launches, bumped by workgroup 0 at the end) makes this launch's tags unique:
  // granules left by the previous launch — or replay — never match, so no per-launch memset is needed
  const uint32_t tag0 = __hip_atomic_load((__attribute__((address_space(1))) uint32_t*)gen, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) * 32u;
  // ---- every weight this wave will ever need, in flight at once
  float w[PM_MAX_LAYERS + 1][PM_CPL];
  float bias[PM_MAX_LAYERS + 1];
#pragma unroll
  for (int l = 0; l <= PM_MAX_LAYERS; l++) {
    bias[l] = 0.f;
#pragma unroll
    for (int k = 0; k < PM_CPL; k++) w[l][k] = 0.f;
    if (l <= d.depth) {
      const bool heads = (l == d.depth);
      const int n_out = heads ? d.n_rot + 3 : d.width;
      const int in_dim = pm_in_dim(d, l, emb);
      if (row < n_out) {
        const float* rp;
        if (!heads) { rp = d.W[l] + (size_t)row * in_dim; bias[l] = d.b[l][row]; }
        else if (row < d.n_rot) { rp = d.W_rot + (size_t)row * in_dim; bias[l] = d.b_rot[row]; }
        else { rp = d.W_tr + (size_t)(row - d.n_rot) * in_dim; bias[l] = d.b_tr[row - d.n_rot]; }
#pragma unroll
        for (int k = 0; k < PM_CPL; k++) {
          const int i = lane + 64 * k;
          if (i < in_dim) w[l][k] = rp[i];
        }
      }
    }
  }
  // ---- embedding (every workgroup computes its own copy; workgroup 0 keeps it for the backward)
  float embv = 0.f;
  if (lane < emb) {
    const float tv = t[0];
    embv = tv;
    if (lane > 0) {
      const int kf = (lane - 1) >> 1;
      const float f = (float)(1 << kf);
      embv = ((lane - 1) & 1) ? cosf(tv * f) : sinf(tv * f);
    }
    if (wave == 0) s_in[lane] = embv;
    if (wave == 0 && rank == 0) acts[lane] = embv;
  }
  if (threadIdx.x == 0) s_failed = 0;
  // TEST HOOK (tests/test_gpu_deform.py): the word behind the sticky status word of a persistent sync_state makes workgroup 1 keep
  // one layer's hand-off to itself — bit 0 in the forward, bit 1 in the backward — so that the consumers' bounded spins time out
  const bool fault = sticky != nullptr && rank == 1 && (sticky[1] & PM_FAULT_BIT) != 0u;
  PM_TRACE(0);
  // (behind the weight loads in the memory queue, off the chain: the first layer waits for the weights anyway)
  const bool local = pm_chain_is_local(xcc_gran, rank, n_chain, tag0 + 31u, err, sticky, lane, wave, &s_local);
  // (4 J + 3 > width, e.g. 64 joints: the head rows beyond `width` live in workgroups of their own, which only take part in
  // the last stage)
  const bool extra = rank * PMF_WAVES >= d.width;
#pragma unroll
  for (int l = 0; l <= PM_MAX_LAYERS; l++) {
    if (l <= d.depth && !(extra && l < d.depth)) {
      const bool heads = (l == d.depth);
      const int n_out = heads ? d.n_rot + 3 : d.width;
      const int in_dim = pm_in_dim(d, l, emb);
      PM_TRACE(1 + 2 * l);
      if (l > 0) {
        if (wave == 0) {
          // input of layer l = h_{l-1} (behind the embedding when layer l-1 was the skip layer)
          float v[4];
          bool ok = (s_failed == 0);
          if (ok) ok = pm_sweep(gran + (size_t)(l - 1) * d.width, d.width, tag0 + (uint32_t)l, v, err, sticky, lane, local);
          if (!ok && lane == 0) s_failed = 1;
          const int off = (l - 1 == d.skip) ? emb : 0;
          if (off > 0 && lane < emb) s_in[lane] = embv;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int r = lane + 64 * k;
            if (r < d.width) s_in[off + r] = v[k];
          }
        }
        __syncthreads();
      }
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < PM_CPL; k++) {
        const int i = lane + 64 * k;
        if (i < in_dim) acc += w[l][k] * s_in[i];
      }
      acc = wave_sum(acc);
      PM_TRACE(2 + 2 * l);
      if (lane == 63 && row < n_out) {
        float v = acc + bias[l];
        if (s_failed) v = __builtin_nanf("");
        if (!heads) {
          v = s_failed ? v : fmaxf(v, 0.f);
          if (!(fault && l == 1)) pm_store_granule(gran + (size_t)l * d.width + row, tag0 + (uint32_t)(l + 1), v, local);
          acts[emb + (size_t)l * d.width + row] = v;
        } else if (row < d.n_rot) {
          rotation[row] = rot_bias4 ? v + rot_bias4[row & 3] : v;
        } else {
          translation[row - d.n_rot] = v;
        }
      }
      __syncthreads();  // s_in is rewritten by wave 0 for the next layer
    }
  }
  // workgroup 0 has swept the last hidden layer, which every workgroup published after reading `gen`
  if (rank == 0 && threadIdx.x == 0)
    __hip_atomic_store((__attribute__((address_space(1))) uint32_t*)gen, tag0 / 32u + 1u, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// Backward chain in one launch.  Wave `col` owns hidden unit `col`: column (off_l + col) of every consumer
// matrix l >= 1 (registers), and row `col` of every weight-gradient matrix.  Stage l (depth .. 0):
//   v_l = dz_l (the heads: the incoming output gradients) -> LDS, from the granules of stage l + 1;
//   D_l[col] = sum_r M_l[r][off_l + col] v_l[r]  -> granule, the gradient w.r.t. h_{l-1};
//   dW_l[col][:] = v_l[col] * input_l[:], db_l[col] = v_l[col]  (plain stores, off the critical path: the
//   sweeping wave 0 postpones its own rows to the end, the others write while wave 0 polls).
// The granules live behind the forward's in `acts` (zeroed by the forward's memset node, so the backward
// needs none); tags carry a generation word that the last stage bumps, so a second backward over the same
// activations (retain_graph) never matches the first one's granules.
// The reverse sweep of the kinematic chain in front of the backward chain (riggs_pose_mlp_backward_fk): the gradients of the
// two heads' outputs are then not read but computed here — by EVERY chain workgroup, redundantly, while its weights are on
// their way: dL/dlocal_rot = FK^T (dL/dtransforms, dL/dd_nodes) (+ g_rot, if given), dL/dglobal_trans = g_tr + sum_j dL/dd_nodes_j.
// As a launch of its own in front of this one the sweep costs 10 us of a 385 us frame (one workgroup, 2 x 23 dependent steps).
struct PmFkArgs {
  int J;  // 0: no chain in front (g_rot / g_tr are the heads' gradients)
  const float *local_rot, *joints;
  const int32_t* parents;
  const float *dG, *g_nodes;
  const float* transforms;  // optional: the chain's forward result (riggs_lbs_forward_fk / riggs_fk_forward), so that only the reverse sweep runs
  float *dq_out, *dgt_out;  // optional: workgroup 0 leaves the two gradients here (NULL: not wanted)
  // optional: the template frame's pose regulariser of the stage-2 objective (train_rig.py:474-482: lambda_template_fixed *
  // mean((local_rotation - (1,0,0,0))^2), on the template camera only) as a cotangent —  dL/dlocal_rot += coef * (q - unit)
  // with coef = 2 lambda / (4 J) or 0, a device scalar the host refreshes between replays — and its value mean((q - unit)^2)
  const float* fixed_coef;
  float* fixed_loss_out;
};

// the one-launch-per-layer path's form of the same term (the fused kernel adds it while the chain's gradients sit in LDS)
__global__ __launch_bounds__(256) void pm_fixed_add_kernel(int J, const float* __restrict__ local_rot, const float* __restrict__ coef,
                                                           float* __restrict__ dq, float* __restrict__ loss_out) {
  __shared__ float s_p[4];
  const int t = threadIdx.x;
  float d = 0.f;
  if (t < 4 * J) {
    d = local_rot[t] - ((t & 3) == 0 ? 1.f : 0.f);
    if (coef) dq[t] += coef[0] * d;
  }
  float ss = d * d;
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((t & 63) == 0) s_p[t >> 6] = ss;
  __syncthreads();
  if (t == 0 && loss_out) loss_out[0] = ((s_p[0] + s_p[1]) + (s_p[2] + s_p[3])) / (float)(4 * J);
}
#undef PM_FAULT_BIT
#define PM_FAULT_BIT 2u
__global__ __launch_bounds__(PMF_WAVES * 64) void pm_backward_fused_kernel(PoseMlpDesc d, PoseMlpGradDesc g, PmFkArgs fk,
                                                                           const float* __restrict__ acts,
                                                                           const float* __restrict__ g_rot,
                                                                           const float* __restrict__ g_tr,
                                                                           unsigned long long* gran, uint32_t* err,
                                                                           uint32_t* sticky, uint32_t* gen, float* __restrict__ flat,
                                                                           const float* __restrict__ wt, int n_chain, int stride,
                                                                           unsigned long long* xcc_gran,
                                                                           unsigned long long* trace) {
  __shared__ float s_v[PM_MAX_HEAD];
  __shared__ int s_failed, s_local;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rank = (int)blockIdx.x / stride;  // (placement of the chain: see pm_chain_is_local)
  if ((int)blockIdx.x % stride != 0 || rank >= n_chain) return;
  const int col = rank * PMF_WAVES + wave;
  const bool extra = rank * PMF_WAVES >= d.width;  // (head rows beyond `width`: only the weight-gradient rows of the heads)
  const int emb = 1 + 2 * d.multires;
  const int n_head = d.n_rot + 3;
  const uint32_t tag0 = __hip_atomic_load((__attribute__((address_space(1))) uint32_t*)gen, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) * 32u;
  // ---- (with the kinematic chain in front) the first wave's loads for the reverse sweep go out ahead of its weights'
  __shared__ float s_gq[4 * MAX_J + 4];  // the heads' gradients: dL/dlocal_rot, then dL/dglobal_trans
  FkIn fin;
  if (fk.J > 0 && wave == 0) fk_load(fk.J, fk.local_rot, fk.joints, fk.parents, fk.dG, fk.g_nodes, fin, fk.transforms);
  // ---- this wave's column of every consumer matrix, from the transposed copy the forward launch left in
  // `wt` (one coalesced 1 KB row per matrix), and the forward activations (8 KB) into LDS: no global load
  // is left on the chain or in the weight-gradient rows
  float wc[PM_MAX_LAYERS + 1][4];
  float wc_head = 0.f;
#pragma unroll
  for (int l = 1; l <= PM_MAX_LAYERS; l++) {
#pragma unroll
    for (int k = 0; k < 4; k++) wc[l][k] = 0.f;
    if (l <= d.depth && col < d.width) {
      const int n_rows = (l == d.depth) ? n_head : d.width;
      const float* rp = wt + ((size_t)(l - 1) * PM_MAX_W + col) * PM_MAX_HEAD;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int r = lane + 64 * k;
        if (r < n_rows) wc[l][k] = rp[r];
      }
      if (l == d.depth && lane + 256 < n_rows) wc_head = rp[lane + 256];  // (head rows 256 ..)
    }
  }
  __shared__ float s_acts[PM_MAX_EMB + PM_MAX_LAYERS * PM_MAX_W];
  __shared__ float s_late[PM_MAX_LAYERS + 1];
  for (int i = threadIdx.x; i < emb + d.depth * d.width; i += blockDim.x) s_acts[i] = acts[i];
  if (threadIdx.x == 0) s_failed = 0;
  // TEST HOOK (tests/test_gpu_deform.py): the word behind the sticky status word of a persistent sync_state makes workgroup 1 keep
  // one layer's hand-off to itself — bit 0 in the forward, bit 1 in the backward — so that the consumers' bounded spins time out
  const bool fault = sticky != nullptr && rank == 1 && (sticky[1] & PM_FAULT_BIT) != 0u;
  PM_TRACE(0);
  // the sweep (one joint per lane of the first wave: fk_device.h) while the second wave checks the placement of the chain;
  // the check's barrier publishes both results
  if (fk.J > 0 && wave == 0) {
    FkLane f;
    fk_wave_forward(fk.J, fin, f, fk.transforms != nullptr);
    float dq[4];
    fk_wave_backward(fk.J, fin, f, dq);
    if (lane < fk.J) {
#pragma unroll
      for (int e = 0; e < 4; e++) s_gq[4 * lane + e] = dq[e];
    }
  }
  const bool local = pm_chain_is_local(xcc_gran, rank, n_chain, tag0 + 31u, err, sticky, lane, wave, &s_local, 1);
  if (fk.J > 0) {
    if (threadIdx.x < 3) {
      float sgt = g_tr ? g_tr[threadIdx.x] : 0.f;
      if (fk.g_nodes) for (int k = 0; k < fk.J; k++) sgt += fk.g_nodes[3 * k + threadIdx.x];
      s_gq[4 * fk.J + threadIdx.x] = sgt;
      if (rank == 0 && fk.dgt_out) fk.dgt_out[threadIdx.x] = sgt;
    }
    __syncthreads();
    if (g_rot && (int)threadIdx.x < 4 * fk.J) s_gq[threadIdx.x] += g_rot[threadIdx.x];
    if ((fk.fixed_coef || fk.fixed_loss_out) && wave == 0) {  // (4 J <= 256: four elements per lane of the first wave)
      float ss = 0.f;
      for (int i = lane; i < 4 * fk.J; i += 64) {
        const float dqf = fk.local_rot[i] - ((i & 3) == 0 ? 1.f : 0.f);
        ss += dqf * dqf;
      }
      for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
      if (rank == 0 && lane == 0 && fk.fixed_loss_out) fk.fixed_loss_out[0] = ss / (float)(4 * fk.J);
    }
    if (fk.fixed_coef && (int)threadIdx.x < 4 * fk.J)
      s_gq[threadIdx.x] += fk.fixed_coef[0] * (fk.local_rot[threadIdx.x] - ((threadIdx.x & 3) == 0 ? 1.f : 0.f));
    __syncthreads();
    if (rank == 0 && fk.dq_out && (int)threadIdx.x < 4 * fk.J) fk.dq_out[threadIdx.x] = s_gq[threadIdx.x];
  }
  PM_TRACE(1);
#pragma unroll
  for (int l = PM_MAX_LAYERS; l >= 0; l--) {
    if (l <= d.depth && !(extra && l < d.depth)) {
      const bool heads = (l == d.depth);
      const int n_rows = heads ? n_head : d.width;
      const int in_dim = pm_in_dim(d, l, emb);
      PM_TRACE(2 + 2 * (d.depth - l));
      if (wave == 0) {
        if (heads) {
#pragma unroll
          for (int k = 0; k < PM_MAX_HEAD / 64; k++) {
            const int r = lane + 64 * k;
            if (fk.J > 0) s_v[r] = (r < n_head) ? s_gq[r] : 0.f;  // (n_rot = 4 J: the translation's three follow the quaternions)
            else s_v[r] = (r < d.n_rot) ? g_rot[r] : (r < n_head ? g_tr[r - d.n_rot] : 0.f);
          }
        } else {
          float h[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int r = lane + 64 * k;
            h[k] = (r < d.width) ? s_acts[emb + (size_t)l * d.width + r] : 0.f;
          }
          float v[4];
          bool ok = (s_failed == 0);
          if (ok) ok = pm_sweep(gran + (size_t)l * d.width, d.width, tag0 + (uint32_t)(l + 1), v, err, sticky, lane, local);
          if (!ok && lane == 0) s_failed = 1;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int r = lane + 64 * k;
            if (r < PM_MAX_W) s_v[r] = (r < d.width && h[k] > 0.f) ? v[k] : 0.f;
          }
        }
      }
      __syncthreads();
      PM_TRACE(3 + 2 * (d.depth - l));
      float sv[4];
#pragma unroll
      for (int k = 0; k < 4; k++) sv[k] = s_v[lane + 64 * k];
      const float sv_head = heads ? s_v[lane + 256] : 0.f;
      float vr = (col < n_rows) ? s_v[col] : 0.f;
      if (s_failed) vr = __builtin_nanf("");
      __syncthreads();  // s_v may be rewritten by wave 0 from here on
      // (a) the chain: gradient w.r.t. h_{l-1}, published for stage l - 1   [granule slot l - 1]
      if (l >= 1) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int r = lane + 64 * k;
          if (r < n_rows) acc += wc[l][k] * sv[k];
        }
        if (heads && lane + 256 < n_rows) acc += wc_head * sv_head;
        acc = wave_sum(acc);
        if (lane == 63 && col < d.width && !(fault && l == 2)) pm_store_granule(gran + (size_t)(l - 1) * d.width + col, tag0 + (uint32_t)l, acc, local);
      }
      // (b) weight / bias gradients of matrix l, row `col`
      if (wave == 0) {
        if (lane == 0) s_late[l] = vr;  // written out after the chain, shared among the eight waves
      } else if (col < n_rows) {
        int m = l, r = col;
        if (heads && col >= d.n_rot) { m = d.depth + 1; r = col - d.n_rot; }
        float* out = flat + g.w_off[m] + (size_t)r * in_dim;
        for (int c = lane; c < in_dim; c += 64) out[c] = vr * pm_input(d, s_acts, l, emb, c);
        if (lane == 0) flat[g.b_off[m] + r] = vr;
      }
    }
  }
  PM_TRACE(4 + 2 * d.depth);
  __syncthreads();
  {
    // wave 0's own rows (it was busy polling): matrix l goes to wave l % 8
    const int col0 = rank * PMF_WAVES;
#pragma unroll
    for (int l = PM_MAX_LAYERS; l >= 0; l--) {
      if (l <= d.depth && (l % PMF_WAVES) == wave) {
        const bool heads = (l == d.depth);
        const int n_rows = heads ? n_head : d.width;
        const int in_dim = pm_in_dim(d, l, emb);
        if (col0 < n_rows) {
          int m = l, r = col0;
          if (heads && col0 >= d.n_rot) { m = d.depth + 1; r = col0 - d.n_rot; }
          const float vr = s_late[l];
          float* out = flat + g.w_off[m] + (size_t)r * in_dim;
          for (int c = lane; c < in_dim; c += 64) out[c] = vr * pm_input(d, s_acts, l, emb, c);
          if (lane == 0) flat[g.b_off[m] + r] = vr;
        }
      }
    }
  }
  if (wave == 0) {
    PM_TRACE(5 + 2 * d.depth);
    // workgroup 0 has swept the last stage, i.e. every workgroup has read `gen`: bump it for a later backward
    if (rank == 0 && lane == 0)
      __hip_atomic_store((__attribute__((address_space(1))) uint32_t*)gen, tag0 / 32u + 1u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace riggs

using namespace riggs;

extern "C" {

static int pm_fill(PoseMlpDesc& d, int32_t depth, int32_t width, int32_t multires, int32_t skip, int32_t n_rot,
                   const float* const* weights, const float* const* biases, const float* W_rot, const float* b_rot,
                   const float* W_tr, const float* b_tr) {
  RIGGS_REQUIRE(depth >= 1 && depth <= PM_MAX_LAYERS, "PoseMLP depth out of range");
  RIGGS_REQUIRE(width >= 1 && width <= PM_MAX_W, "PoseMLP width must be <= 256");
  RIGGS_REQUIRE(multires >= 0 && 1 + 2 * multires <= PM_MAX_EMB, "PoseMLP multires out of range");
  RIGGS_REQUIRE(n_rot >= 1 && n_rot + 3 <= PM_MAX_HEAD, "PoseMLP rotation head too wide");
  d.depth = depth; d.width = width; d.multires = multires; d.skip = skip; d.n_rot = n_rot;
  for (int l = 0; l < depth; l++) { d.W[l] = weights[l]; d.b[l] = biases[l]; }
  d.W_rot = W_rot; d.b_rot = b_rot; d.W_tr = W_tr; d.b_tr = b_tr;
  return 0;
}

// 8 when an eighth of the device (one XCD) holds `n_chain` workgroups of `kernel` at once, else 1
static int g_pm_one_xcd = 1;  // riggs_pose_mlp_set_placement
static int pm_chain_stride(const void* kernel, int n_chain) {
  int per_cu = 0, dev = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, PMF_WAVES * 64, 0) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 1;
  }
  return (per_cu * (cus / PM_CHAIN_STRIDE) >= n_chain) ? PM_CHAIN_STRIDE : 1;
}

int riggs_pose_mlp_set_placement(int32_t one_xcd) { g_pm_one_xcd = one_xcd ? 1 : 0; return 0; }

static unsigned long long* g_pm_trace = nullptr;  // 128 u64: forward stamps [0,64), backward stamps [64,128)
int riggs_pose_mlp_set_trace(void* dev_u64x128) { g_pm_trace = (unsigned long long*)dev_u64x128; return 0; }

// The one-launch kernels take the rotation + translation head when it fits one layer's width (n_rot + 3 <= width: up to 63
// joints at width 256) or, for the networks the reference builds (width >= 128), up to PM_MAX_HEAD rows (79 joints: the
// rows beyond `width` get workgroups of their own that join for the last stage — C5's 64 joints are 259 rows); narrow
// test networks with wide heads and anything larger run one launch per layer.
static bool pm_one_launch(int32_t width, int32_t n_rot) {
  if (option(OPT_POSE_MLP_LAYERED)) return false;  // (a host re-running a frame whose hand-off was lost)
  return n_rot + 3 <= width || (width >= 128 && n_rot + 3 <= PM_MAX_HEAD);
}
// acts: activations, then (256-byte aligned) the state of the one-launch kernels:
//   [forward granules + gen (used when the caller passes no persistent sync_state) | err, gen, pad, pad | backward granules]
static size_t pm_acts_core(int32_t depth, int32_t width, int32_t multires) {
  return ((size_t)(1 + 2 * multires) + (size_t)depth * width + 63) & ~(size_t)63;
}
static size_t pm_xcc_offset(int32_t depth, int32_t width) { return (2 * (size_t)depth * width + 4 + 63) & ~(size_t)63; }
static size_t pm_sync_floats(int32_t depth, int32_t width) {  // granules (2 floats each) + {gen, pad...} + the placement check's granules
  return pm_xcc_offset(depth, width) + 2 * PM_MAX_CHAIN;
}
size_t riggs_pose_mlp_sync_bytes(int32_t depth, int32_t width) { return pm_sync_floats(depth, width) * sizeof(float); }
// index (in 32-bit words) of the sticky status word inside sync_state: bit 0 = a hand-off spin of the one-launch
// kernels timed out (their outputs were poisoned with NaN)
size_t riggs_pose_mlp_status_word(int32_t depth, int32_t width) { return 2 * (size_t)depth * width + 1; }
// ... then the transposed consumer matrices (depth x 256 x 256) the forward launch prepares for the backward
size_t riggs_pose_mlp_acts_floats(int32_t depth, int32_t width, int32_t multires) {
  return pm_acts_core(depth, width, multires) + 2 * pm_sync_floats(depth, width) + (size_t)depth * PM_MAX_W * PM_MAX_HEAD;
}

int riggs_pose_mlp_forward(int32_t depth, int32_t width, int32_t multires, int32_t skip, int32_t n_rot,
                           const float* const* weights, const float* const* biases, const float* W_rot,
                           const float* b_rot, const float* W_tr, const float* b_tr, const float* t,
                           const float* rot_bias4, void* sync_state, float* acts, float* rotation,
                           float* translation, riggs_stream stream) {
  PoseMlpDesc d;
  int rc = pm_fill(d, depth, width, multires, skip, n_rot, weights, biases, W_rot, b_rot, W_tr, b_tr);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(PROF_POSE_FWD, s);
  if (pm_one_launch(width, n_rot)) {
    const size_t sf = pm_sync_floats(depth, width);
    float* own = acts + pm_acts_core(depth, width, multires);  // [own forward state | backward state]
    float* fs = (float*)sync_state;
    if (fs == nullptr) {
      // no persistent state from the caller: clear a private one every launch (one aligned memset node)
      fs = own;
      RIGGS_HIP_CHECK(hipMemsetAsync(fs, 0, sf * sizeof(float), s));
    }
    // the chain's workgroups are every eighth one of the launch (one XCD: pm_chain_is_local), the others transpose
    const int n_chain = ((width > n_rot + 3 ? width : n_rot + 3) + PMF_WAVES - 1) / PMF_WAVES;
    static int stride_f[PM_MAX_CHAIN + 1];
    if (stride_f[n_chain] == 0) stride_f[n_chain] = pm_chain_stride(reinterpret_cast<const void*>(pm_forward_fused_kernel), n_chain);
    const int stride = g_pm_one_xcd ? stride_f[n_chain] : 1;
    hipLaunchKernelGGL(pm_forward_fused_kernel, dim3(stride > 1 ? n_chain * stride : n_chain + 64), dim3(PMF_WAVES * 64), 0, s, d, t, rot_bias4, acts,
                       (unsigned long long*)fs, (uint32_t*)(fs + 2 * (size_t)depth * width),
                       sync_state ? (uint32_t*)(fs + 2 * (size_t)depth * width) + 1 : nullptr, (uint32_t*)(own + sf), (int)sf,
                       rotation, translation, own + 2 * sf, n_chain, stride, (unsigned long long*)(fs + pm_xcc_offset(depth, width)),
                       g_pm_trace);
    RIGGS_HIP_CHECK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(pm_embed_kernel, dim3(1), dim3(64), 0, s, d, t, acts);
  for (int l = 0; l <= depth; l++) {
    const int n_out = (l == depth) ? n_rot + 3 : width;
    hipLaunchKernelGGL(pm_layer_kernel, dim3((n_out + 3) / 4), dim3(256), 0, s, d, l, acts, rot_bias4, rotation, translation);
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

static size_t pm_bwd_core(int32_t depth, int32_t width) {
  return ((size_t)depth * width + (size_t)(depth + 1) * PM_MAX_IN + 1) & ~(size_t)1;
}
size_t riggs_pose_mlp_backward_workspace_floats(int32_t depth, int32_t width, int32_t multires) {
  (void)multires;
  return pm_bwd_core(depth, width);  // used by the layered (one launch per layer) variant only
}

static int pm_backward_impl(int32_t depth, int32_t width, int32_t multires, int32_t skip, int32_t n_rot,
                            const float* const* weights, const float* const* biases, const float* W_rot,
                            const float* b_rot, const float* W_tr, const float* b_tr, float* acts,
                            const float* g_rotation, const float* g_translation, float* workspace,
                            float* flat_grads, void* sync_state, riggs_stream stream, PmFkArgs fk) {
  PoseMlpDesc d;
  int rc = pm_fill(d, depth, width, multires, skip, n_rot, weights, biases, W_rot, b_rot, W_tr, b_tr);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int emb = 1 + 2 * multires;
  float* dzs = workspace;
  float* dh = workspace + (size_t)depth * width;  // (depth + 1) vectors of PM_MAX_IN: input-gradient of consumer l
  PoseMlpGradDesc g;
  int64_t o = 0;
  int rows = 0;
  for (int l = 0; l < depth; l++) {
    const int in_l = (l == 0) ? emb : ((l - 1 == skip) ? width + emb : width);
    g.w_off[l] = o; o += (int64_t)width * in_l;
    g.b_off[l] = o; o += width;
    g.row_start[l] = rows; rows += width;
  }
  const int in_h = (depth - 1 == skip) ? width + emb : width;
  g.w_off[depth] = o; o += (int64_t)n_rot * in_h;
  g.b_off[depth] = o; o += n_rot;
  g.row_start[depth] = rows; rows += n_rot;
  g.w_off[depth + 1] = o; o += (int64_t)3 * in_h;
  g.b_off[depth + 1] = o; o += 3;
  g.row_start[depth + 1] = rows; rows += 3;
  g.row_start[depth + 2] = rows;
  ProfScope ps(PROF_POSE_BWD, s);
  if (pm_one_launch(width, n_rot)) {
    float* tail = acts + pm_acts_core(depth, width, multires) + pm_sync_floats(depth, width);
    const int nr = width > n_rot + 3 ? width : n_rot + 3;
    const int n_chain = (nr + PMF_WAVES - 1) / PMF_WAVES;
    static int stride_b[PM_MAX_CHAIN + 1];
    if (stride_b[n_chain] == 0) stride_b[n_chain] = pm_chain_stride(reinterpret_cast<const void*>(pm_backward_fused_kernel), n_chain);
    const int stride = g_pm_one_xcd ? stride_b[n_chain] : 1;
    hipLaunchKernelGGL(pm_backward_fused_kernel, dim3(n_chain * stride), dim3(PMF_WAVES * 64), 0, s, d, g, fk,
                       acts, g_rotation, g_translation, (unsigned long long*)(tail + 4), (uint32_t*)tail,
                       sync_state ? (uint32_t*)sync_state + 2 * (size_t)depth * width + 1 : nullptr,
                       (uint32_t*)tail + 1, flat_grads, tail + pm_sync_floats(depth, width), n_chain, stride,
                       (unsigned long long*)(tail + pm_xcc_offset(depth, width)), g_pm_trace ? g_pm_trace + 64 : nullptr);
    RIGGS_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (fk.J > 0) {
    // (one launch per layer — networks the one-launch kernels do not take: the reverse sweep is a launch of its own as well)
    RIGGS_REQUIRE(fk.dq_out && fk.dgt_out, "riggs_pose_mlp_backward_fk needs dL_dlocal_rot and dL_dglobal_trans");
    if (g_translation) RIGGS_HIP_CHECK(hipMemcpyAsync(fk.dgt_out, g_translation, 12, hipMemcpyDeviceToDevice, s));
    else RIGGS_HIP_CHECK(hipMemsetAsync(fk.dgt_out, 0, 12, s));
    int rc2 = riggs_fk_backward(fk.J, fk.local_rot, fk.joints, fk.parents, fk.dG, fk.g_nodes, fk.dq_out, fk.dgt_out, stream);
    if (rc2) return rc2;
    RIGGS_REQUIRE(g_rotation == nullptr, "riggs_pose_mlp_backward_fk: an extra rotation gradient needs the one-launch kernels");
    if (fk.fixed_coef || fk.fixed_loss_out) {
      hipLaunchKernelGGL(pm_fixed_add_kernel, dim3(1), dim3(256), 0, s, fk.J, fk.local_rot, fk.fixed_coef, fk.dq_out, fk.fixed_loss_out);
      RIGGS_HIP_CHECK(hipGetLastError());
    }
    g_rotation = fk.dq_out; g_translation = fk.dgt_out;
  }
  RIGGS_HIP_CHECK(hipMemsetAsync(dh, 0, (size_t)(depth + 1) * PM_MAX_IN * sizeof(float), s));
  for (int l = depth; l >= 0; l--) {
    const int in_l = (l == 0) ? emb : ((l - 1 == skip) ? width + emb : width);
    const float* dh_out = (l == depth) ? nullptr : dh + (size_t)(l + 1) * PM_MAX_IN;
    hipLaunchKernelGGL(pm_backward_step_kernel, dim3(l == 0 ? 1 : (in_l + 63) / 64, PM_RG), dim3(256), 0, s, d, l, acts,
                       g_rotation, g_translation, dh_out, dh + (size_t)l * PM_MAX_IN, dzs);
  }
  hipLaunchKernelGGL(pm_backward_weights_kernel, dim3(rows), dim3(64), 0, s, d, g, acts, dzs, g_rotation,
                     g_translation, flat_grads);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_pose_mlp_backward(int32_t depth, int32_t width, int32_t multires, int32_t skip, int32_t n_rot,
                            const float* const* weights, const float* const* biases, const float* W_rot,
                            const float* b_rot, const float* W_tr, const float* b_tr, float* acts,
                            const float* g_rotation, const float* g_translation, float* workspace,
                            float* flat_grads, void* sync_state, riggs_stream stream) {
  PmFkArgs fk;
  memset(&fk, 0, sizeof(fk));
  return pm_backward_impl(depth, width, multires, skip, n_rot, weights, biases, W_rot, b_rot, W_tr, b_tr, acts, g_rotation,
                          g_translation, workspace, flat_grads, sync_state, stream, fk);
}

int riggs_pose_mlp_backward_fk(int32_t depth, int32_t width, int32_t multires, int32_t skip, int32_t n_rot,
                               const float* const* weights, const float* const* biases, const float* W_rot,
                               const float* b_rot, const float* W_tr, const float* b_tr, float* acts, int32_t num_joints,
                               const float* local_rot, const float* joints, const int32_t* parents, const float* transforms,
                               const float* dL_dtransforms, const float* dL_dd_nodes, const float* g_rotation,
                               const float* g_translation, float* dL_dlocal_rot, float* dL_dglobal_trans,
                               const float* template_fixed_coef, float* template_fixed_loss, float* workspace,
                               float* flat_grads, void* sync_state, riggs_stream stream) {
  RIGGS_REQUIRE(num_joints >= 1 && num_joints <= MAX_J && 4 * num_joints == n_rot, "the rotation head must predict one quaternion per joint");
  RIGGS_REQUIRE(local_rot && joints && parents && dL_dtransforms, "riggs_pose_mlp_backward_fk: missing chain input");
  PmFkArgs fk;
  fk.J = num_joints; fk.local_rot = local_rot; fk.joints = joints; fk.parents = parents; fk.dG = dL_dtransforms;
  fk.g_nodes = dL_dd_nodes; fk.dq_out = dL_dlocal_rot; fk.dgt_out = dL_dglobal_trans; fk.transforms = transforms;
  fk.fixed_coef = template_fixed_coef; fk.fixed_loss_out = template_fixed_loss;
  return pm_backward_impl(depth, width, multires, skip, n_rot, weights, biases, W_rot, b_rot, W_tr, b_tr, acts, g_rotation,
                          g_translation, workspace, flat_grads, sync_state, stream, fk);
}

}  // extern "C"
