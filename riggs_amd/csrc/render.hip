// Tile binning and per-tile alpha compositing, forward and backward.
//
// Workgroup = one 16x16 tile = 256 threads = 4 wave64 (a wave owns 4 rows x 16 columns).
// Instances of the tile are staged 256 at a time through LDS as three float4 records
// (48 B / instance, gathered as whole 16-B words), then every pixel walks the batch with
// wave-uniform LDS addresses (broadcast reads, no bank conflicts).
#include <stdlib.h>

#include "raster_internal.h"

namespace riggs {

#define ALPHA_MIN (1.0f / 255.0f)
#define ALPHA_MAX 0.99f
#define T_EPS 0.0001f
#define LOG2E 1.4426950408889634f

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }

// ------------------------------------------------------------------ binning helpers
__global__ __launch_bounds__(256) void gather_tiles_kernel(int N, const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ tiles,
                                                           uint32_t* __restrict__ tt_sorted) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < N) tt_sorted[s] = tiles[order[s]];
}

// duplicateWithKeys in DEPTH order: sorted slot s owns instances [offsets[s-1], offsets[s]) and
// writes (tile id, Gaussian index) row-major over its rectangle.  A later STABLE sort by tile id
// then yields exactly the order of a 64-bit (tile | depth-bits) key sort with ties in ascending
// Gaussian index.  Also pads [R, cap) with the sentinel tile id n_tiles.
__global__ __launch_bounds__(256) void emit_kernel(int N, int grid_x, int n_tiles, int64_t cap,
                                                   const uint32_t* __restrict__ order,
                                                   const uint32_t* __restrict__ offsets,
                                                   const uint32_t* __restrict__ tiles,
                                                   const ushort4* __restrict__ rect, uint32_t* __restrict__ keys,
                                                   uint32_t* __restrict__ vals, uint32_t* __restrict__ counters) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t R = N > 0 ? offsets[N - 1] : 0u;
  if (s == 0) { counters[0] = R; counters[1] = ((int64_t)R > cap) ? 1u : 0u; }
  if (s < N) {
    const uint32_t g = order[s];
    const uint32_t n = tiles[g];
    if (n != 0) {
      int64_t off = (s == 0) ? 0 : offsets[s - 1];
      const ushort4 rc = rect[g];
      for (int y = rc.y; y < rc.w; y++)
        for (int x = rc.x; x < rc.z; x++) {
          if (off < cap) { keys[off] = (uint32_t)(y * grid_x + x); vals[off] = g; }
          off++;
        }
    }
  }
  // pad region (grid covers max(N, cap) threads)
  if (s >= (int64_t)R && s < cap) { keys[s] = (uint32_t)n_tiles; vals[s] = 0u; }
}

__global__ __launch_bounds__(256) void ranges_kernel(int64_t n, int n_tiles, const uint32_t* __restrict__ keys,
                                                     const uint32_t* __restrict__ counters,
                                                     uint2* __restrict__ ranges) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t R = min((int64_t)counters[0], n);
  if (i >= R) return;
  const uint32_t k = keys[i];
  if (i == 0) ranges[k].x = 0;
  else {
    const uint32_t pk = keys[i - 1];
    if (pk != k) { ranges[pk].y = (uint32_t)i; ranges[k].x = (uint32_t)i; }
  }
  if (i == R - 1) ranges[k].y = (uint32_t)R;
}

int launch_gather_tiles(int N, const uint32_t* order, const uint32_t* tiles, uint32_t* tt_sorted, hipStream_t s) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(gather_tiles_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, order, tiles, tt_sorted);
  return 0;
}
int launch_emit(int N, int grid_x, int n_tiles, int64_t cap, const uint32_t* order, const uint32_t* offsets,
                const uint32_t* tiles, const ushort4* rect, uint32_t* keys, uint32_t* vals, uint32_t* counters,
                hipStream_t s) {
  int64_t n = N > cap ? N : cap;
  if (n < 1) n = 1;
  hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, N, grid_x, n_tiles, cap, order,
                     offsets, tiles, rect, keys, vals, counters);
  return 0;
}
int launch_ranges(int64_t n, int n_tiles, const uint32_t* keys_sorted, const uint32_t* counters, uint2* ranges,
                  hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(ranges_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, n_tiles, keys_sorted,
                     counters, ranges);
  return 0;
}

// ------------------------------------------------------------------ render forward
// Checkpoint slots: tile t, chunk c (= 64 consecutive instances of the tile list) -> slot
// slot_base[t] + c with slot_base[t] = (lower_bound(tile t) >> 6) + t  (monotone, <= R/64 + T).
__global__ __launch_bounds__(256) void slot_base_kernel(int64_t n, int n_tiles, const uint32_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ counters,
                                                        uint32_t* __restrict__ slot_base) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t > n_tiles) return;
  const int64_t R = min((int64_t)counters[0], n);
  int64_t lo = 0, hi = R;  // first index with key >= t
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < (uint32_t)t) lo = mid + 1; else hi = mid;
  }
  slot_base[t] = (uint32_t)(lo >> 6) + (uint32_t)t;
}
int launch_slot_base(int64_t n, int n_tiles, const uint32_t* keys_sorted, const uint32_t* counters,
                     uint32_t* slot_base, hipStream_t s) {
  hipLaunchKernelGGL(slot_base_kernel, dim3((n_tiles + 1 + 255) / 256), dim3(256), 0, s, n, n_tiles, keys_sorted,
                     counters, slot_base);
  return 0;
}

// Per-pixel front-to-back compositing.  The only true dependency between consecutive instances
// is the transmittance product, so the kernel evaluates the Gaussian falloff (LDS reads, conic
// form, exp) of FOUR instances at once and then applies the four T updates with selects
// (no exec-mask juggling): a deep tile is bound by the latency of one wave's instruction
// stream, not by throughput, and this shortens that stream ~4x.  Every 64 instances the
// running state (T, C, D) is checkpointed for the chunk-parallel backward.
#define FWD_ILP 4
__global__ __launch_bounds__(256) void render_fwd_kernel(RenderArgs a) {
  __shared__ float4 s_xyd[256 + FWD_ILP];
  __shared__ float4 s_con[256 + FWD_ILP];
  __shared__ float4 s_rgb[256 + FWD_ILP];
  __shared__ uint32_t s_max[4];
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const int tile = blockIdx.x;
  const int tid = threadIdx.x;
  const int pxi = (tile % gx) * RIGGS_TILE + (tid & 15);
  const int pyi = (tile / gx) * RIGGS_TILE + (tid >> 4);
  const bool inside = pxi < a.W && pyi < a.H;
  const float pfx = (float)pxi, pfy = (float)pyi;
  const uint2 range = a.ranges[tile];
  const int total = (int)(range.y - range.x);
  const uint32_t slot0 = a.slot_base[tile];
  bool done = !inside;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
  uint32_t last = 0;
  if (tid < FWD_ILP) {  // permanent null records behind a full batch
    s_xyd[256 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_con[256 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_rgb[256 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int base = 0; base < total; base += 256) {
    if (__syncthreads_count(done) == 256) break;
    const int nb = min(256, total - base);
    {
      float4 xy = make_float4(0.f, 0.f, 0.f, 0.f), co = xy, cc = xy;  // opacity 0 => never contributes
      if (tid < nb) {
        const uint32_t id = a.point_list[range.x + base + tid];
        xy = a.xyd[id]; co = a.conic_o[id]; cc = a.rgb[id];
      }
      s_xyd[tid] = xy; s_con[tid] = co; s_rgb[tid] = cc;
    }
    __syncthreads();
    for (int j = 0; j < nb; j += FWD_ILP) {
      if (__builtin_amdgcn_ballot_w64(!done) == 0) break;  // this wave's 64 pixels are finished
      if (((base + j) & 63) == 0 && !done) {
        float* ck = a.ckpt + ((size_t)(slot0 + ((base + j) >> 6)) * 5) * 256 + tid;
        ck[0] = T; ck[256] = C0; ck[512] = C1; ck[768] = C2; ck[1024] = D;
      }
      float alpha[FWD_ILP], depth[FWD_ILP];
      bool valid[FWD_ILP];
      bool any = false;
#pragma unroll
      for (int k = 0; k < FWD_ILP; k++) {
        const float4 xy = s_xyd[j + k];
        const float4 co = s_con[j + k];
        const float dx = xy.x - pfx, dy = xy.y - pfy;
        const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
        alpha[k] = fminf(ALPHA_MAX, co.w * fast_exp(power));
        valid[k] = (power <= 0.0f) && (alpha[k] >= ALPHA_MIN);
        depth[k] = xy.z;
        any = any || valid[k];
      }
      if (__builtin_amdgcn_ballot_w64(any && !done) == 0) continue;  // nothing lands on this wave's pixels
#pragma unroll
      for (int k = 0; k < FWD_ILP; k++) {
        const float4 c = s_rgb[j + k];
        const bool v = valid[k] && !done;
        const float test_T = T * (1.0f - alpha[k]);
        const bool stop = v && (test_T < T_EPS);
        const bool use = v && !stop;
        done = done || stop;
        const float w = use ? alpha[k] * T : 0.f;
        C0 += c.x * w; C1 += c.y * w; C2 += c.z * w;
        D += depth[k] * w; A += w;
        T = use ? test_T : T;
        last = use ? (uint32_t)(base + j + k + 1) : last;
      }
    }
  }
  // per-tile max of n_contrib bounds the work of the backward
  uint32_t m = inside ? last : 0u;
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  __syncthreads();
  if ((tid & 63) == 0) s_max[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) a.tile_max[tile] = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
  if (inside) {
    const size_t pid = (size_t)pyi * a.W + pxi, HW = (size_t)a.H * a.W;
    a.final_T[pid] = T;
    a.n_contrib[pid] = last;
    a.final_acc[pid] = make_float4(C0, C1, C2, D);
    a.out_color[pid] = C0 + T * a.bg[0];
    a.out_color[HW + pid] = C1 + T * a.bg[1];
    a.out_color[2 * HW + pid] = C2 + T * a.bg[2];
    a.out_depth[pid] = D;
    a.out_alpha[pid] = A;
  }
}

// ---- quad-lane forward ------------------------------------------------------------------------------
// Quad-lane forward (RIGGS_RENDER_FWD=4; the default is the eight-lane kernel below, which grew out of this one).
// The per-pixel chain of a deep tile is what bounds this kernel (thousands of
// contributing instances walked by ONE wave), so the wave is laid out as 16 pixels x 4 instance lanes:
// the four lanes of a DPP quad evaluate four CONSECUTIVE instances of the same pixel at once, their
// transmittances come from a 3-step exclusive product scan inside the quad, the T < 1e-4 stop is
// resolved with one ballot, and colour is accumulated per lane (folded over the quad only at
// checkpoints and at the end).  ~55 instructions per 4 instances instead of ~54 per instance.
// A tile is split over 4 workgroups (4 pixel rows each, one wave per row) so that the waves of a deep
// tile land on different CUs; the next batch of 256 instances is prefetched into registers while the
// current one is composited.
#define QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define QUAD_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (float)(v)), ctrl, 0xf, 0xf, true))
#define QUAD_U(v, ctrl) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(v), ctrl, 0xf, 0xf, true))

__device__ __forceinline__ float quad_sum(float v) {
  v += QUAD_F(v, QP(1, 0, 3, 2));
  v += QUAD_F(v, QP(2, 3, 0, 1));
  return v;
}

// Most instances of a tile's list never reach a given 16 x 4 pixel block (the lists are built from the
// 3-sigma tile rectangles; >90 % of the bench scene's pixels never saturate, so their waves walk the WHOLE
// list of up to 47k instances): while a batch is staged, every instance is tested against the workgroup's
// block with the axis-aligned extent of its alpha >= 1/255 ellipse (xyd.w / rgb.w, from the preprocess
// kernel; conservative) and only the survivors are kept, compacted per 64-instance chunk so that the
// checkpoints of the backward stay at multiples of 64 of the ORIGINAL list position.
template <int FQ_BATCH>
__global__ __launch_bounds__(256) void render_fwd_quad_kernel(RenderArgs a) {
  // A round stages FQ_BATCH = 1024 instances (16 chunks; thread t holds instances t, t+256, ...): the walk of a
  // 47k-entry list is a chain of rounds, each costing a global-load latency and a barrier whatever survives.
  __shared__ float4 s_xyd[FQ_BATCH];
  __shared__ float4 s_con[FQ_BATCH];
  __shared__ float4 s_rgb[FQ_BATCH];
  __shared__ unsigned short s_pos[FQ_BATCH];  // position of the survivor inside its batch
  __shared__ int s_cnt[FQ_BATCH / 64];        // survivors per chunk
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qx = lane >> 2, j = lane & 3;
  if (a.items) {
    // tiles without instances: background only, one pixel per thread
    const int n_empty = (int)a.item_ctr[2];
    for (int e = blockIdx.x; e < n_empty; e += gridDim.x) {
      const int t = (int)a.empties[e];
      const int px = (t % gx) * RIGGS_TILE + (tid & 15), py = (t / gx) * RIGGS_TILE + (tid >> 4);
      if (px < a.W && py < a.H) {
        const size_t pid = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;
        a.final_T[pid] = 1.0f; a.n_contrib[pid] = 0u; a.final_acc[pid] = make_float4(0.f, 0.f, 0.f, 0.f);
        a.out_color[pid] = a.bg[0]; a.out_color[HW + pid] = a.bg[1]; a.out_color[2 * HW + pid] = a.bg[2];
        a.out_depth[pid] = 0.f; a.out_alpha[pid] = 0.f;
      }
    }
  }
  const int n_items = a.items ? (int)a.item_ctr[0] * 4 : 0;  // four 16x4 pixel blocks per non-empty tile
  for (int turn = 0;; turn++) {
  int tile, sub;
  if (a.items) {
    // work list: the non-empty tiles in the order bin_offsets_kernel wrote them (longest lists first), dealt
    // round-robin to the resident workgroups (a shared dequeue word costs more than it balances: same-address
    // atomics from 8 XCDs serialise at ~10-60 ns each).  The four blocks of a tile get workgroup ids 8 apart =
    // the same XCD / L2.
    const int i = (int)blockIdx.x + turn * (int)gridDim.x, full = (n_items >> 5) << 5;
    if (i >= n_items) break;
    int p;
    if (i < full) { p = ((i >> 5) << 3) + (i & 7); sub = (i >> 3) & 3; }
    else { p = (full >> 2) + ((i - full) >> 2); sub = (i - full) & 3; }
    tile = (int)a.items[p];
    if (a.only_tile >= 0 && tile != a.only_tile) continue;  // diagnostics (tools/fwd_placement.py)
  } else {
    if (turn > 0) break;
    // XCD-aware mapping: workgroup b runs on XCD b % 8, so the four workgroups of a tile (which read the same
    // list and the same records) are given ids 8 apart — same L2 — instead of four neighbouring ids
    const int b = blockIdx.x, full = (int)(gridDim.x >> 5) << 5;
    if (!a.xcd_map) { tile = b >> 2; sub = b & 3; }
    else if (b < full) { tile = ((b >> 5) << 3) + (b & 7); sub = (b >> 3) & 3; }
    else { tile = (full >> 2) + ((b - full) >> 2); sub = (b - full) & 3; }
  }
  const int prow = sub * 4 + wave;                       // pixel row inside the tile
  const int pxi = (tile % gx) * RIGGS_TILE + qx;
  const int pyi = (tile / gx) * RIGGS_TILE + prow;
  const bool inside = pxi < a.W && pyi < a.H;
  const float pfx = (float)pxi, pfy = (float)pyi;
  const uint2 range = a.ranges[tile];
  const int total = (int)(range.y - range.x);
  const uint32_t slot0 = a.slot_base[tile];
  const int pix = prow * 16 + qx;                        // pixel index inside the tile (checkpoint layout)
  const float bx0 = (float)((tile % gx) * RIGGS_TILE), bx1 = bx0 + (float)(RIGGS_TILE - 1);
  const float by0 = (float)((tile / gx) * RIGGS_TILE + sub * 4), by1 = by0 + 3.0f;
  bool done = !inside;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f, Tstop = -1.0f;
  uint32_t last = 0;
  const unsigned long long t_begin = a.trace ? wall_clock64() : 0ull;  // optional statistics (riggs_raster_set_trace)
  uint32_t st_rounds = 0, st_surv = 0, st_iters = 0, st_full = 0;
  // prefetch registers for the next round (FQ_BATCH / 256 instances per thread), and the list entries of the round
  // after it, so that the gathers of a round never wait for their own addresses: a tile whose pixels never saturate
  // is a chain of rounds, and every global latency on that chain is kernel time (tools/fwd_placement.py)
  float4 n_xy[FQ_BATCH / 256], n_co[FQ_BATCH / 256], n_cc[FQ_BATCH / 256];
  uint32_t n_id[FQ_BATCH / 256];
#pragma unroll
  for (int k = 0; k < FQ_BATCH / 256; k++) {
    n_xy[k] = make_float4(0.f, 0.f, 0.f, 0.f); n_co[k] = n_xy[k]; n_cc[k] = n_xy[k];
    n_id[k] = 0u;
    if (FQ_BATCH + k * 256 + tid < total) n_id[k] = a.point_list[range.x + FQ_BATCH + k * 256 + tid];
    if (k * 256 + tid < total) {
      const uint32_t id = a.point_list[range.x + k * 256 + tid];
      n_xy[k] = a.xyd[id]; n_co[k] = a.conic_o[id]; n_cc[k] = a.rgb[id];
    }
  }
  // checkpoints of a round are held in registers (lane j of a pixel's quad keeps the one of chunk j) and stored at
  // the start of the NEXT round, ahead of that round's prefetch loads: vmcnt retires in order, so stores issued
  // after the loads would make the wait for the loads also wait for the stores' acknowledgements
  static_assert(FQ_BATCH == 256, "one held checkpoint per quad lane");
  float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f;
  bool hv = false;
  int hbase = 0;
  auto flush_ckpt = [&]() {
    if (hv) {
      float* ck = a.ckpt + ((size_t)(slot0 + (hbase >> 6) + j) * 5) * 256 + pix;
      ck[0] = h0; ck[256] = h1; ck[512] = h2; ck[768] = h3; ck[1024] = h4;
    }
    hv = false;
  };
  for (int base = 0; base < total; base += FQ_BATCH) {
    if (__syncthreads_count(done) == 256) break;
#pragma unroll
    for (int k = 0; k < FQ_BATCH / 256; k++) {
      const int chunk = k * 4 + wave, inb = k * 256 + tid;  // this wave's 64 lanes = one chunk of the batch
      const bool keep = (base + inb < total) &&
                        (!a.cull || ((n_xy[k].x + n_xy[k].w >= bx0) && (n_xy[k].x - n_xy[k].w <= bx1) &&
                                     (n_xy[k].y + n_cc[k].w >= by0) && (n_xy[k].y - n_cc[k].w <= by1)));
      const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
      const int cnt = __builtin_popcountll(mask);
      const int slot = chunk * 64 + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
      if (keep) { s_xyd[slot] = n_xy[k]; s_con[slot] = n_co[k]; s_rgb[slot] = n_cc[k]; s_pos[slot] = (unsigned short)inb; }
      if (lane >= cnt && lane < ((cnt + 7) & ~7)) {  // null records up to the next multiple of 8
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        s_xyd[chunk * 64 + lane] = z; s_con[chunk * 64 + lane] = z; s_rgb[chunk * 64 + lane] = z; s_pos[chunk * 64 + lane] = 0;
      }
      if (lane == 0) s_cnt[chunk] = cnt;
      st_surv += (uint32_t)cnt;
    }
    st_rounds++;
    flush_ckpt();
    hbase = base;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FQ_BATCH / 256; k++) {
      n_xy[k] = make_float4(0.f, 0.f, 0.f, 0.f); n_co[k] = n_xy[k]; n_cc[k] = n_xy[k];
      if (base + FQ_BATCH + k * 256 + tid < total) {
        const uint32_t id = n_id[k];
        n_xy[k] = a.xyd[id]; n_co[k] = a.conic_o[id]; n_cc[k] = a.rgb[id];
      }
      if (base + 2 * FQ_BATCH + k * 256 + tid < total) n_id[k] = a.point_list[range.x + base + 2 * FQ_BATCH + k * 256 + tid];
    }
    // one quad step: four consecutive instances (one per lane of the quad) of this lane's pixel
    auto quad_step = [&](float alpha, bool valid_in, float depth, const float4 c, int pos1) {
      const bool valid = valid_in && !done;
      const float om = valid ? 1.0f - alpha : 1.0f;
      // exclusive product scan over the quad: E_j = prod_{j' < j} om_j'
      float b1 = QUAD_F(om, QP(0, 0, 1, 2)); b1 = (j >= 1) ? b1 : 1.0f;            // [1, o0, o1, o2]
      float s1 = QUAD_F(b1, QP(0, 0, 1, 2)); s1 = (j >= 1) ? s1 : 1.0f;            // [1, 1, o0, o1]
      float s2 = QUAD_F(b1, QP(0, 0, 0, 1)); s2 = (j >= 2) ? s2 : 1.0f;            // [1, 1, 1, o0]
      const float E = b1 * s1 * s2;                                                // [1, o0, o0 o1, o0 o1 o2]
      const float Tj = T * E;
      const float test_T = Tj * om;
      const bool sc = valid && (test_T < T_EPS);
      const uint64_t bits = __builtin_amdgcn_ballot_w64(sc);
      const uint32_t qb = (uint32_t)(bits >> (lane & 60)) & 0xFu;
      const bool first_stop_before = (qb & ((1u << j) - 1u)) != 0u;
      const bool use = valid && !sc && !first_stop_before;
      const float w = use ? alpha * Tj : 0.f;
      C0 += c.x * w; C1 += c.y * w; C2 += c.z * w;
      D += depth * w; A += w;
      last = use ? (uint32_t)pos1 : last;
      if (sc && !first_stop_before) Tstop = Tj;  // transmittance in front of the instance that ends the pixel
      const float prod4 = QUAD_F(E * om, QP(3, 3, 3, 3));
      const bool nostop = (qb == 0u);
      T = (nostop && !done) ? T * prod4 : T;
      done = done || !nostop;
    };
    // two quad steps (8 instances) per iteration: the falloff of the second step overlaps the
    // dependent transmittance chain of the first
    for (int k = 0; k < FQ_BATCH / 64; k++) {
      const int cbase = base + 64 * k;
      if (cbase >= total) break;
      if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
      {
        // checkpoint of the state BEFORE instance cbase: fold the quad's partial sums
        const float k0 = quad_sum(C0), k1 = quad_sum(C1), k2 = quad_sum(C2), kd = quad_sum(D);
        if (j == k) { h0 = T; h1 = k0; h2 = k1; h3 = k2; h4 = kd; hv = !done; }
      }
      const int nk = s_cnt[k];
    for (int g = 64 * k; g < 64 * k + nk; g += 8) {
      if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
      const float4 xyA = s_xyd[g + j];
      const float4 xyB = s_xyd[g + 4 + j];
      const float dxA = xyA.x - pfx, dyA = xyA.y - pfy, dxB = xyB.x - pfx, dyB = xyB.y - pfy;
      const float4 coA = s_con[g + j], coB = s_con[g + 4 + j];
      const float pwA = -0.5f * (coA.x * dxA * dxA + coA.z * dyA * dyA) - coA.y * dxA * dyA;
      const float pwB = -0.5f * (coB.x * dxB * dxB + coB.z * dyB * dyB) - coB.y * dxB * dyB;
      const float alA = fminf(ALPHA_MAX, coA.w * fast_exp(pwA));
      const float alB = fminf(ALPHA_MAX, coB.w * fast_exp(pwB));
      const bool vA = (pwA <= 0.0f) && (alA >= ALPHA_MIN);
      const bool vB = (pwB <= 0.0f) && (alB >= ALPHA_MIN);
      st_iters++;
      if (__builtin_amdgcn_ballot_w64((vA || vB) && !done) == 0) continue;
      st_full++;
      const float4 cA = s_rgb[g + j], cB = s_rgb[g + 4 + j];
      quad_step(alA, vA, xyA.z, cA, base + (int)s_pos[g + j] + 1);
      quad_step(alB, vB, xyB.z, cB, base + (int)s_pos[g + 4 + j] + 1);
    }
    }
  }
  if (a.trace && lane == 0) {  // per wave: {100 MHz ticks, rounds, survivors of this wave's chunk, iterations, iterations with a contribution, list length}
    unsigned long long* tr = a.trace + ((size_t)(tile * 4 + sub) * 4 + wave) * 6;
    tr[0] = wall_clock64() - t_begin; tr[1] = st_rounds; tr[2] = (unsigned long long)st_surv | ((unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xFFFFu) << 32) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 0xFu) << 48) | ((unsigned long long)(t_begin & 0xFFFull) << 52); tr[3] = st_iters; tr[4] = st_full; tr[5] = (unsigned long long)total;
  }
  flush_ckpt();
  // fold the quad
  const float k0 = quad_sum(C0), k1 = quad_sum(C1), k2 = quad_sum(C2), kd = quad_sum(D), ka = quad_sum(A);
  float ts = Tstop;
  ts = fmaxf(ts, QUAD_F(ts, QP(1, 0, 3, 2)));
  ts = fmaxf(ts, QUAD_F(ts, QP(2, 3, 0, 1)));
  uint32_t lm = last;
  lm = max(lm, QUAD_U(lm, QP(1, 0, 3, 2)));
  lm = max(lm, QUAD_U(lm, QP(2, 3, 0, 1)));
  const float Tfin = (ts >= 0.f) ? ts : T;
  uint32_t m = inside ? lm : 0u;
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if (lane == 0 && m > 0) atomicMax(&a.tile_max[tile], m);  // tile_max is zeroed before the launch
  if (inside && j == 0) {
    const size_t pid = (size_t)pyi * a.W + pxi, HW = (size_t)a.H * a.W;
    a.final_T[pid] = Tfin;
    a.n_contrib[pid] = lm;
    a.final_acc[pid] = make_float4(k0, k1, k2, kd);
    a.out_color[pid] = k0 + Tfin * a.bg[0];
    a.out_color[HW + pid] = k1 + Tfin * a.bg[1];
    a.out_color[2 * HW + pid] = k2 + Tfin * a.bg[2];
    a.out_depth[pid] = kd;
    a.out_alpha[pid] = ka;
  }
  __syncthreads();  // the staging buffers are reused by the next item
  }
}

// ---- eight instance-lanes per pixel -------------------------------------------------------------------------------------
// Same algorithm as the quad-lane kernel with the wave turned the other way: 8 pixels x 8 instance lanes, workgroup =
// 4 waves = an 8 x 4 pixel block (eight per tile).  A wave needs ONE scan step per 8 instances instead of two quad steps
// over 16 pixels: about the same number of wave instructions in total, spread over twice the waves.  That matters
// because a wave alone on its SIMD is already issue-bound (one VALU instruction per 4 cycles), and the slowest 16 x 4
// block of the quad-lane kernel (a tile whose pixels never saturate walks its whole list: 12 rounds, ~130 steps) takes
// as long ALONE on the chip as the whole launch (tools/fwd_placement.py).  Halving the pixels per workgroup halves that
// chain, and the smaller block culls more of the list.  A round stages 256 instances (4 chunks, one per wave); lane i < 4
// of a pixel's eight holds the checkpoint of chunk i of the round.
#define ODPP_F(old, v, ctrl, bank) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(v)), ctrl, 0xf, bank, false))
#define ODPP_U(old, v, ctrl, bank) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, 0xf, bank, false))
#define DPP_ROW_SHL4 0x104
#define DPP_ROW_SHR4 0x114
#define DPP_HALF_MIRROR 0x141
__device__ __forceinline__ float oct_sum(float v) {
  v += QUAD_F(v, QP(1, 0, 3, 2));
  v += QUAD_F(v, QP(2, 3, 0, 1));
  v += ODPP_F(0.f, v, DPP_HALF_MIRROR, 0xf);
  return v;
}
template <int B, bool TRACE>
__global__ __launch_bounds__(256) void render_fwd_oct_kernel(RenderArgs a) {
  static_assert(B == 256 || B == 512, "one held checkpoint per lane of a pixel's eight");
  constexpr int K = B / 256;  // instances per thread and round
  __shared__ float4 s_xyd[B];
  __shared__ float4 s_con[B];
  __shared__ float4 s_rgb[B];
  __shared__ unsigned short s_pos[B];  // position of the survivor inside its batch
  __shared__ int s_cnt[B / 64];        // survivors per chunk
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pl = lane >> 3, i = lane & 7;
  if (a.items) {
    // tiles without instances: background only, one pixel per thread
    const int n_empty = (int)a.item_ctr[2];
    for (int e = blockIdx.x; e < n_empty; e += gridDim.x) {
      const int t = (int)a.empties[e];
      const int px = (t % gx) * RIGGS_TILE + (tid & 15), py = (t / gx) * RIGGS_TILE + (tid >> 4);
      if (px < a.W && py < a.H) {
        const size_t pid = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;
        a.final_T[pid] = 1.0f; a.n_contrib[pid] = 0u; a.final_acc[pid] = make_float4(0.f, 0.f, 0.f, 0.f);
        a.out_color[pid] = a.bg[0]; a.out_color[HW + pid] = a.bg[1]; a.out_color[2 * HW + pid] = a.bg[2];
        a.out_depth[pid] = 0.f; a.out_alpha[pid] = 0.f;
      }
    }
  }
  const int n_items = a.items ? (int)a.item_ctr[0] * 8 : 0;  // eight 8x4 pixel blocks per non-empty tile
  for (int turn = 0;; turn++) {
  int tile, sub;
  if (a.items) {
    // the eight blocks of a tile get workgroup ids 8 apart = the same XCD / L2
    const int it = (int)blockIdx.x + turn * (int)gridDim.x, full = (n_items >> 6) << 6;
    if (it >= n_items) break;
    int p;
    if (it < full) { p = ((it >> 6) << 3) + (it & 7); sub = (it >> 3) & 7; }
    else { p = (full >> 3) + ((it - full) >> 3); sub = (it - full) & 7; }
    tile = (int)a.items[p];
    if (a.only_tile >= 0 && tile != a.only_tile) continue;  // diagnostics (tools/fwd_placement.py)
  } else {
    if (turn > 0) break;
    const int b = blockIdx.x, full = (int)(gridDim.x >> 6) << 6;
    if (!a.xcd_map) { tile = b >> 3; sub = b & 7; }
    else if (b < full) { tile = ((b >> 6) << 3) + (b & 7); sub = (b >> 3) & 7; }
    else { tile = (full >> 3) + ((b - full) >> 3); sub = (b - full) & 7; }
  }
  const int prow = (sub >> 1) * 4 + wave;                 // pixel row inside the tile
  const int pcol = (sub & 1) * 8 + pl;                    // pixel column inside the tile
  const int pxi = (tile % gx) * RIGGS_TILE + pcol;
  const int pyi = (tile / gx) * RIGGS_TILE + prow;
  const bool inside = pxi < a.W && pyi < a.H;
  const float pfx = (float)pxi, pfy = (float)pyi;
  const uint2 range = a.ranges[tile];
  const int total = (int)(range.y - range.x);
  const uint32_t slot0 = a.slot_base[tile];
  const int pix = prow * 16 + pcol;                       // pixel index inside the tile (checkpoint layout)
  const float bx0 = (float)((tile % gx) * RIGGS_TILE + (sub & 1) * 8), bx1 = bx0 + 7.0f;
  const float by0 = (float)((tile / gx) * RIGGS_TILE + (sub >> 1) * 4), by1 = by0 + 3.0f;
  bool done = !inside;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f, Tstop = -1.0f;
  uint32_t last = 0;
  const unsigned long long t_begin = (TRACE && a.trace) ? wall_clock64() : 0ull;
  uint32_t st_rounds = 0, st_surv = 0, st_iters = 0, st_full = 0;
  // prefetch registers for the next round (one instance per thread) and the list entry of the round after it
  float4 n_xy[K], n_co[K], n_cc[K];
  uint32_t n_id[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    n_xy[k] = make_float4(0.f, 0.f, 0.f, 0.f); n_co[k] = n_xy[k]; n_cc[k] = n_xy[k];
    n_id[k] = 0u;
    if (B + k * 256 + tid < total) n_id[k] = a.point_list[range.x + B + k * 256 + tid];
    if (k * 256 + tid < total) {
      const uint32_t id = a.point_list[range.x + k * 256 + tid];
      n_xy[k] = a.xyd[id]; n_co[k] = a.conic_o[id]; n_cc[k] = a.rgb[id];
    }
  }
  // checkpoints are held one round (lane i keeps chunk i's) and stored ahead of the next round's loads
  float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f;
  bool hv = false;
  int hbase = 0;
  auto flush_ckpt = [&]() {
    if (hv) {
      float* ck = a.ckpt + ((size_t)(slot0 + (hbase >> 6) + i) * 5) * 256 + pix;
      ck[0] = h0; ck[256] = h1; ck[512] = h2; ck[768] = h3; ck[1024] = h4;
    }
    hv = false;
  };
  for (int base = 0; base < total; base += B) {
    if (__syncthreads_count(done) == 256) break;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int chunk = k * 4 + wave, inb = k * 256 + tid;  // this wave's 64 lanes = one chunk of the batch
      const bool keep = (base + inb < total) &&
                        (!a.cull || ((n_xy[k].x + n_xy[k].w >= bx0) && (n_xy[k].x - n_xy[k].w <= bx1) &&
                                     (n_xy[k].y + n_cc[k].w >= by0) && (n_xy[k].y - n_cc[k].w <= by1)));
      const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
      const int cnt = __builtin_popcountll(mask);
      const int slot = chunk * 64 + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
      if (keep) { s_xyd[slot] = n_xy[k]; s_con[slot] = n_co[k]; s_rgb[slot] = n_cc[k]; s_pos[slot] = (unsigned short)inb; }
      if (lane >= cnt && lane < ((cnt + 7) & ~7)) {  // null records up to the next multiple of 8
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        s_xyd[chunk * 64 + lane] = z; s_con[chunk * 64 + lane] = z; s_rgb[chunk * 64 + lane] = z; s_pos[chunk * 64 + lane] = 0;
      }
      if (lane == 0) s_cnt[chunk] = cnt;
      if constexpr (TRACE) st_surv += (uint32_t)cnt;
    }
    if constexpr (TRACE) st_rounds++;
    flush_ckpt();
    hbase = base;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
      n_xy[k] = make_float4(0.f, 0.f, 0.f, 0.f); n_co[k] = n_xy[k]; n_cc[k] = n_xy[k];
      if (base + B + k * 256 + tid < total) {
        const uint32_t id = n_id[k];
        n_xy[k] = a.xyd[id]; n_co[k] = a.conic_o[id]; n_cc[k] = a.rgb[id];
      }
      if (base + 2 * B + k * 256 + tid < total) n_id[k] = a.point_list[range.x + base + 2 * B + k * 256 + tid];
    }
    for (int k = 0; k < B / 64; k++) {
      const int cbase = base + 64 * k;
      if (cbase >= total) break;
      if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
      {
        // checkpoint of the state BEFORE instance cbase: fold the eight lanes' partial sums
        const float k0 = oct_sum(C0), k1 = oct_sum(C1), k2 = oct_sum(C2), kd = oct_sum(D);
        if (i == k) { h0 = T; h1 = k0; h2 = k1; h3 = k2; h4 = kd; hv = !done; }
      }
      const int nk = s_cnt[k];
      for (int g = 64 * k; g < 64 * k + nk; g += 8) {
        // (no 'all pixels finished' test here: it costs eight instructions per step of every wave to save a few steps
        // once per wave; the chunk loop above has it)
        const float4 xy = s_xyd[g + i];
        const float dx = xy.x - pfx, dy = xy.y - pfy;
        const float4 co = s_con[g + i];
        const float pw = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
        const float alpha = fminf(ALPHA_MAX, co.w * fast_exp(pw));
        const bool valid = (pw <= 0.0f) && (alpha >= ALPHA_MIN) && !done;
        if constexpr (TRACE) st_iters++;
        if (__builtin_amdgcn_ballot_w64(valid) == 0) continue;
        if constexpr (TRACE) st_full++;
        const float4 c = s_rgb[g + i];
        const int pos1 = base + (int)s_pos[g + i] + 1;
        // one step: eight consecutive instances (one per lane) of this lane's pixel
        const float om = valid ? 1.0f - alpha : 1.0f;
        // exclusive product scan over the eight lanes: inside each quad as in the quad-lane kernel ...
        float b1 = QUAD_F(om, QP(0, 0, 1, 2)); b1 = ((i & 3) >= 1) ? b1 : 1.0f;     // [1, o0, o1, o2]
        float s1 = QUAD_F(b1, QP(0, 0, 1, 2)); s1 = ((i & 3) >= 1) ? s1 : 1.0f;     // [1, 1, o0, o1]
        float s2 = QUAD_F(b1, QP(0, 0, 0, 1)); s2 = ((i & 3) >= 2) ? s2 : 1.0f;     // [1, 1, 1, o0]
        float E = b1 * s1 * s2;                                                     // [1, o0, o0 o1, o0 o1 o2] per quad
        // ... then the upper quad takes the lower quad's total (row_shr:4 written to banks 1 and 3 only)
        const float Pq = QUAD_F(E * om, QP(3, 3, 3, 3));                            // product of the own quad
        E *= ODPP_F(1.0f, Pq, DPP_ROW_SHR4, 0xA);
        const float Tj = T * E;
        const float test_T = Tj * om;
        const bool sc = valid && (test_T < T_EPS);
        const uint64_t bits = __builtin_amdgcn_ballot_w64(sc);
        const uint32_t ob = (uint32_t)(bits >> (lane & 56)) & 0xFFu;
        const bool first_stop_before = (ob & ((1u << i) - 1u)) != 0u;
        const bool use = valid && !sc && !first_stop_before;
        const float w = use ? alpha * Tj : 0.f;
        C0 += c.x * w; C1 += c.y * w; C2 += c.z * w;
        D += xy.z * w; A += w;
        last = use ? (uint32_t)pos1 : last;
        if (sc && !first_stop_before) Tstop = Tj;  // transmittance in front of the instance that ends the pixel
        // product of all eight: the upper quad's running total, handed down to the lower quad (row_shl:4, banks 0 and 2)
        const float X3 = QUAD_F(E * om, QP(3, 3, 3, 3));
        const float prod8 = ODPP_F(X3, X3, DPP_ROW_SHL4, 0x5);
        const bool nostop = (ob == 0u);
        T = (nostop && !done) ? T * prod8 : T;
        done = done || !nostop;
      }
    }
  }
  if (TRACE && a.trace && lane == 0) {
    unsigned long long* tr = a.trace + ((size_t)(tile * 8 + sub) * 4 + wave) * 6;
    tr[0] = wall_clock64() - t_begin; tr[1] = st_rounds;
    tr[2] = (unsigned long long)st_surv | ((unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xFFFFu) << 32) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 0xFu) << 48);
    tr[3] = st_iters; tr[4] = st_full; tr[5] = (unsigned long long)total;
  }
  flush_ckpt();
  // fold the eight lanes
  const float k0 = oct_sum(C0), k1 = oct_sum(C1), k2 = oct_sum(C2), kd = oct_sum(D), ka = oct_sum(A);
  float ts = Tstop;
  ts = fmaxf(ts, QUAD_F(ts, QP(1, 0, 3, 2)));
  ts = fmaxf(ts, QUAD_F(ts, QP(2, 3, 0, 1)));
  ts = fmaxf(ts, ODPP_F(-1.0f, ts, DPP_HALF_MIRROR, 0xf));
  uint32_t lm = last;
  lm = max(lm, QUAD_U(lm, QP(1, 0, 3, 2)));
  lm = max(lm, QUAD_U(lm, QP(2, 3, 0, 1)));
  lm = max(lm, ODPP_U(0u, lm, DPP_HALF_MIRROR, 0xf));
  const float Tfin = (ts >= 0.f) ? ts : T;
  uint32_t m = inside ? lm : 0u;
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if (lane == 0 && m > 0) atomicMax(&a.tile_max[tile], m);  // tile_max is zeroed before the launch
  if (inside && i == 0) {
    const size_t pid = (size_t)pyi * a.W + pxi, HW = (size_t)a.H * a.W;
    a.final_T[pid] = Tfin;
    a.n_contrib[pid] = lm;
    a.final_acc[pid] = make_float4(k0, k1, k2, kd);
    a.out_color[pid] = k0 + Tfin * a.bg[0];
    a.out_color[HW + pid] = k1 + Tfin * a.bg[1];
    a.out_color[2 * HW + pid] = k2 + Tfin * a.bg[2];
    a.out_depth[pid] = kd;
    a.out_alpha[pid] = ka;
  }
  __syncthreads();  // the staging buffers are reused by the next item
  }
}

// Variant that keeps the instance stream in SGPRs: a tile's instance records are wave-uniform, so
// they are fetched with scalar loads (s_load_dwordx4 through the scalar cache) straight from the
// per-Gaussian arrays — no LDS staging, no workgroup barriers, and the vector pipe only sees the
// per-pixel arithmetic.  Each wave walks the list on its own and stops as soon as its 64 pixels
// are finished.
__global__ __launch_bounds__(256) void render_fwd_sgpr_kernel(RenderArgs a) {
  __shared__ uint32_t s_max[4];
  // constant address space (4): these arrays are read-only for the whole launch, which lets the
  // compiler use the scalar unit for the wave-uniform loads.
  typedef const uint32_t __attribute__((address_space(4))) c_u32;
  typedef float f4v __attribute__((ext_vector_type(4)));
  typedef const f4v __attribute__((address_space(4))) c_f4;
  c_u32* plist = (c_u32*)(uintptr_t)a.point_list;
  c_f4* gxyd = (c_f4*)(uintptr_t)a.xyd;
  c_f4* gcon = (c_f4*)(uintptr_t)a.conic_o;
  c_f4* grgb = (c_f4*)(uintptr_t)a.rgb;
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const int tile = blockIdx.x;
  const int tid = threadIdx.x;
  const int pxi = (tile % gx) * RIGGS_TILE + (tid & 15);
  const int pyi = (tile / gx) * RIGGS_TILE + (tid >> 4);
  const bool inside = pxi < a.W && pyi < a.H;
  const float pfx = (float)pxi, pfy = (float)pyi;
  const uint2 range = a.ranges[tile];
  const int total = (int)(range.y - range.x);
  const uint32_t slot0 = a.slot_base[tile];
  bool done = !inside;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
  uint32_t last = 0;
  for (int j = 0; j < total; j += FWD_ILP) {
    if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
    if ((j & 63) == 0 && !done) {
      float* ck = a.ckpt + ((size_t)(slot0 + (j >> 6)) * 5) * 256 + tid;
      ck[0] = T; ck[256] = C0; ck[512] = C1; ck[768] = C2; ck[1024] = D;
    }
    float alpha[FWD_ILP], depth[FWD_ILP];
    f4v col[FWD_ILP];
    bool valid[FWD_ILP];
    bool any = false;
#pragma unroll
    for (int k = 0; k < FWD_ILP; k++) {
      const int jj = min(j + k, total - 1);
      const uint32_t id = plist[range.x + jj];
      const f4v xy = gxyd[id];
      const f4v co = gcon[id];
      col[k] = grgb[id];
      const float dx = xy.x - pfx, dy = xy.y - pfy;
      const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
      alpha[k] = fminf(ALPHA_MAX, co.w * fast_exp(power));
      valid[k] = (j + k < total) && (power <= 0.0f) && (alpha[k] >= ALPHA_MIN);
      depth[k] = xy.z;
      any = any || valid[k];
    }
    if (__builtin_amdgcn_ballot_w64(any && !done) == 0) continue;
#pragma unroll
    for (int k = 0; k < FWD_ILP; k++) {
      const bool v = valid[k] && !done;
      const float test_T = T * (1.0f - alpha[k]);
      const bool stop = v && (test_T < T_EPS);
      const bool use = v && !stop;
      done = done || stop;
      const float w = use ? alpha[k] * T : 0.f;
      C0 += col[k].x * w; C1 += col[k].y * w; C2 += col[k].z * w;
      D += depth[k] * w; A += w;
      T = use ? test_T : T;
      last = use ? (uint32_t)(j + k + 1) : last;
    }
  }
  uint32_t m = inside ? last : 0u;
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((tid & 63) == 0) s_max[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) a.tile_max[tile] = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
  if (inside) {
    const size_t pid = (size_t)pyi * a.W + pxi, HW = (size_t)a.H * a.W;
    a.final_T[pid] = T;
    a.n_contrib[pid] = last;
    a.final_acc[pid] = make_float4(C0, C1, C2, D);
    a.out_color[pid] = C0 + T * a.bg[0];
    a.out_color[HW + pid] = C1 + T * a.bg[1];
    a.out_color[2 * HW + pid] = C2 + T * a.bg[2];
    a.out_depth[pid] = D;
    a.out_alpha[pid] = A;
  }
}

int launch_render_fwd(const RenderArgs& a, hipStream_t s) {
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE, gy = (a.H + RIGGS_TILE - 1) / RIGGS_TILE;
  if (gx * gy == 0) return 0;
  // A/B switch: 0 = eight-lane kernel (default), 1 = SGPR-stream variant, 2 = pixel-per-lane ILP kernel (no cull),
  // 4 = quad-lane kernel (16 x 4 pixel blocks)
  static const int variant = getenv("RIGGS_RENDER_FWD") ? atoi(getenv("RIGGS_RENDER_FWD")) : 0;
  // (tile_max was cleared together with ranges by riggs_raster_render)
  if (variant == 1) hipLaunchKernelGGL(render_fwd_sgpr_kernel, dim3(gx * gy), dim3(256), 0, s, a);
  else if (variant == 2) hipLaunchKernelGGL(render_fwd_kernel, dim3(gx * gy), dim3(256), 0, s, a);
  else if (variant == 4) hipLaunchKernelGGL(render_fwd_quad_kernel<256>, dim3(gx * gy * 4), dim3(256), 0, s, a);
  else {
    // one workgroup per 8 x 4 block of every tile; with the work list the ones past the non-empty tiles only help
    // with the background of the empty tiles and leave
    static const int ob = getenv("RIGGS_FWD_OCT_BATCH") ? atoi(getenv("RIGGS_FWD_OCT_BATCH")) : 256;  // (512-instance rounds lose)
    if (ob == 512) hipLaunchKernelGGL((render_fwd_oct_kernel<512, true>), dim3(gx * gy * 8), dim3(256), 0, s, a);
    else if (a.trace) hipLaunchKernelGGL((render_fwd_oct_kernel<256, true>), dim3(gx * gy * 8), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((render_fwd_oct_kernel<256, false>), dim3(gx * gy * 8), dim3(256), 0, s, a);
  }
  return 0;
}

// ------------------------------------------------------------------ render backward
// Back-to-front walk per pixel; per-instance gradient contributions are summed over the 64
// pixels of a wave with DPP, the 4 waves meet in LDS, and ONE atomicAdd per value per
// (tile, instance) goes to the per-Gaussian accumulator (48-B record).
__global__ __launch_bounds__(256) void render_bwd_v1_kernel(RenderBwdArgs a) {
  __shared__ float4 s_xyd[256];
  __shared__ float4 s_con[256];
  __shared__ float4 s_rgb[256];
  __shared__ uint32_t s_id[256];
  __shared__ float s_part[4][10];
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const int tile = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int pxi = (tile % gx) * RIGGS_TILE + (tid & 15);
  const int pyi = (tile / gx) * RIGGS_TILE + (tid >> 4);
  const bool inside = pxi < a.W && pyi < a.H;
  const float pfx = (float)pxi, pfy = (float)pyi;
  const uint2 range = a.ranges[tile];
  const int total = (int)(range.y - range.x);
  const size_t pid = (size_t)pyi * a.W + pxi, HW = (size_t)a.H * a.W;
  const float T_final = inside ? a.final_T[pid] : 0.f;
  float T = T_final;
  const int last = inside ? (int)a.n_contrib[pid] : 0;
  float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f, gA = 0.f;
  if (inside) {
    gC0 = a.dL_dcolor[pid]; gC1 = a.dL_dcolor[HW + pid]; gC2 = a.dL_dcolor[2 * HW + pid];
    if (a.dL_ddepth) gD = a.dL_ddepth[pid];
    if (a.dL_dalpha) gA = a.dL_dalpha[pid];
  }
  const float bg_dot = a.bg[0] * gC0 + a.bg[1] * gC1 + a.bg[2] * gC2;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accd = 0.f, acca = 0.f;
  float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, ld = 0.f, last_alpha = 0.f;
  const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
  // the block only needs instances [0, max_last) of the tile
  int max_last = last;
  for (int o = 32; o > 0; o >>= 1) max_last = max(max_last, __shfl_xor(max_last, o));
  __shared__ int s_max[4];
  if (lane == 0) s_max[wave] = max_last;
  __syncthreads();
  max_last = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
  max_last = min(max_last, total);

  for (int hi = max_last; hi > 0; hi -= 256) {
    const int lo = max(0, hi - 256);
    const int nb = hi - lo;
    __syncthreads();
    if (tid < nb) {
      // slot j holds instance position hi-1-j (back-to-front)
      const uint32_t id = a.point_list[range.x + (hi - 1 - tid)];
      s_id[tid] = id;
      s_xyd[tid] = a.xyd[id];
      s_con[tid] = a.conic_o[id];
      s_rgb[tid] = a.rgb[id];
    }
    __syncthreads();
    for (int j = 0; j < nb; j++) {
      const int pos = hi - 1 - j;  // 0-based position in the tile list
      float v_mx = 0.f, v_my = 0.f, v_ca = 0.f, v_cb = 0.f, v_cc = 0.f, v_op = 0.f, v_r = 0.f, v_g = 0.f, v_b = 0.f,
            v_d = 0.f;
      bool active = pos < last;
      if (active) {
        const float4 xy = s_xyd[j];
        const float4 co = s_con[j];
        const float dx = xy.x - pfx, dy = xy.y - pfy;
        const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
        const float G = fast_exp(power);
        const float alpha = fminf(ALPHA_MAX, co.w * G);
        active = (power <= 0.0f) && (alpha >= ALPHA_MIN);
        if (active) {
          const float4 c = s_rgb[j];
          T = T / (1.0f - alpha);
          const float w = alpha * T;
          acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = c.x;
          acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = c.y;
          acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = c.z;
          accd = last_alpha * ld + (1.f - last_alpha) * accd; ld = xy.z;
          acca = last_alpha + (1.f - last_alpha) * acca;
          float dL_dalpha = (c.x - acc0) * gC0 + (c.y - acc1) * gC1 + (c.z - acc2) * gC2 + (xy.z - accd) * gD +
                            (1.0f - acca) * gA;
          dL_dalpha *= T;
          last_alpha = alpha;
          dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
          const float dL_dG = co.w * dL_dalpha;
          const float gdx = G * dx, gdy = G * dy;
          v_mx = dL_dG * (-gdx * co.x - gdy * co.y) * ddelx_dx;
          v_my = dL_dG * (-gdy * co.z - gdx * co.y) * ddely_dy;
          v_ca = -0.5f * gdx * dx * dL_dG;
          v_cb = -gdx * dy * dL_dG;
          v_cc = -0.5f * gdy * dy * dL_dG;
          v_op = G * dL_dalpha;
          v_r = w * gC0; v_g = w * gC1; v_b = w * gC2;
          v_d = w * gD;
        }
      }
      // wave-uniform skip when no pixel of the wave touched this instance
      if (__builtin_amdgcn_ballot_w64(active) != 0) {
        v_mx = wave_sum(v_mx); v_my = wave_sum(v_my); v_ca = wave_sum(v_ca); v_cb = wave_sum(v_cb);
        v_cc = wave_sum(v_cc); v_op = wave_sum(v_op); v_r = wave_sum(v_r); v_g = wave_sum(v_g);
        v_b = wave_sum(v_b); v_d = wave_sum(v_d);
        if (lane == 63) {
          float* g = a.gacc + (size_t)s_id[j] * RIGGS_GACC;
          atomicAdd(g + 0, v_mx); atomicAdd(g + 1, v_my); atomicAdd(g + 2, v_ca); atomicAdd(g + 3, v_cb);
          atomicAdd(g + 4, v_cc); atomicAdd(g + 5, v_op); atomicAdd(g + 6, v_r); atomicAdd(g + 7, v_g);
          atomicAdd(g + 8, v_b);
          if (a.dL_ddepth) atomicAdd(g + 9, v_d);
        }
      }
    }
  }
  (void)s_part;
}

// ---- wave64 scans (DPP).  Inclusive Hillis-Steele inside rows of 16, then row broadcasts.
#define DPP_F(old, v, ctrl, rmask) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(v)), ctrl, rmask, 0xf, false))

__device__ __forceinline__ float wave_excl_prod_scan(float v) {
  v *= DPP_F(1.0f, v, 0x111, 0xf);  // row_shr:1
  v *= DPP_F(1.0f, v, 0x112, 0xf);  // row_shr:2
  v *= DPP_F(1.0f, v, 0x114, 0xf);  // row_shr:4
  v *= DPP_F(1.0f, v, 0x118, 0xf);  // row_shr:8
  v *= DPP_F(1.0f, v, 0x142, 0xa);  // row_bcast:15 -> rows 1,3
  v *= DPP_F(1.0f, v, 0x143, 0xc);  // row_bcast:31 -> rows 2,3
  return DPP_F(1.0f, v, 0x138, 0xf);  // wave_shr:1 : inclusive -> exclusive
}
__device__ __forceinline__ float wave_excl_sum_scan(float v) {
  v += DPP_F(0.0f, v, 0x111, 0xf);
  v += DPP_F(0.0f, v, 0x112, 0xf);
  v += DPP_F(0.0f, v, 0x114, 0xf);
  v += DPP_F(0.0f, v, 0x118, 0xf);
  v += DPP_F(0.0f, v, 0x142, 0xa);
  v += DPP_F(0.0f, v, 0x143, 0xc);
  return DPP_F(0.0f, v, 0x138, 0xf);
}

// Two independent wave64 exclusive scans at once, hand-scheduled.  The compiler expands every step of the generic
// version above into v_mov (identity) + v_mov_dpp + v_op and pads the DPP read-after-write hazard (2 wait states)
// with s_nop, one chain after the other: ~21 slots per product scan.  Here a step is ONE v_op_dpp in place — lanes
// whose DPP source is invalid or masked off are simply not written (bound_ctrl:0), which IS the identity — and the
// second chain fills the first one's hazard slots: 6 x (2 ops + s_nop 0) + 2 shifts for both scans.
#define DUAL_SCAN_STEP(OP, CTRL) \
  OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
#define DUAL_SCAN_BODY(OP)                                                   \
  "s_nop 1\n\t"                                                              \
  DUAL_SCAN_STEP(OP, "row_shr:1 row_mask:0xf bank_mask:0xf")                 \
  DUAL_SCAN_STEP(OP, "row_shr:2 row_mask:0xf bank_mask:0xf")                 \
  DUAL_SCAN_STEP(OP, "row_shr:4 row_mask:0xf bank_mask:0xf")                 \
  DUAL_SCAN_STEP(OP, "row_shr:8 row_mask:0xf bank_mask:0xf")                 \
  DUAL_SCAN_STEP(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf")              \
  DUAL_SCAN_STEP(OP, "row_bcast:31 row_mask:0xc bank_mask:0xf")              \
  "v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"           \
  "v_mov_b32_dpp %3, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"           \
  "s_nop 0"
__device__ __forceinline__ void dual_excl_prod_scan(float& a, float& b) {
  float ea = 1.0f, eb = 1.0f;  // lane 0 has no source in the final shift: it keeps the identity
  asm volatile(DUAL_SCAN_BODY("v_mul_f32_dpp") : "+v"(a), "+v"(b), "+v"(ea), "+v"(eb));
  a = ea; b = eb;
}
__device__ __forceinline__ void dual_excl_sum_scan(float& a, float& b) {
  float ea = 0.0f, eb = 0.0f;
  asm volatile(DUAL_SCAN_BODY("v_add_f32_dpp") : "+v"(a), "+v"(b), "+v"(ea), "+v"(eb));
  a = ea; b = eb;
}

// Chunk-parallel, instance-major backward.  One wave64 owns one chunk of 64 consecutive
// instances of one tile: lane <-> instance, and the wave walks the tile's 256 pixels.  For a
// pixel the 64 transmittances are an exclusive product scan over the lanes seeded with the
// forward's checkpoint, and dL/dalpha needs one more scan (prefix of the projected colour
//   k = gC.c + gD.z + gA), the suffix being  total - prefix.  Per-instance gradients
// accumulate in registers over the 256 pixels — no cross-lane reduction, no atomic
// contention (one atomic per value per (tile, instance) at the end).  Chunks are independent,
// so a tile with thousands of contributing instances spreads over the whole chip instead of
// serialising on four waves.
// Work list of the backward: one entry (tile << 16 | chunk) per chunk that holds a contributing instance
// (chunk * 64 < min(list length, tile_max)).  Workgroup 0 builds it (tiles are scanned 1024 at a time);
// the other workgroups clear the per-Gaussian gradient accumulators meanwhile (this replaces a memset node).
__global__ __launch_bounds__(1024) void render_bwd_worklist_kernel(RenderBwdArgs a) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x > 0) {
    float4* z = reinterpret_cast<float4*>(a.gacc);  // N * RIGGS_GACC floats, RIGGS_GACC % 4 == 0, 256-byte aligned
    const size_t n4 = (size_t)a.n_points * (RIGGS_GACC / 4);
    for (size_t i = (size_t)(blockIdx.x - 1) * 1024 + tid; i < n4; i += (size_t)(gridDim.x - 1) * 1024)
      z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  if (tid == 0) s_carry = 0u;
  __syncthreads();
  for (int base = 0; base < a.n_tiles; base += 1024) {
    const int t = base + tid;
    uint32_t c = 0, limit = 0, sb = 0, rx = 0;
    if (t < a.n_tiles) {
      const uint2 rg = a.ranges[t];
      limit = min(rg.y - rg.x, a.tile_max[t]);
      c = (limit + 63u) >> 6;
      sb = a.slot_base[t]; rx = rg.x;
    }
    uint32_t v = c;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)v, o);
      if (lane >= o) v += u;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    uint32_t off = s_carry;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    // self-contained entries: (tile << 16 | chunk, checkpoint slot, start of the tile's list, instances to walk)
    uint4* out = a.work + (off + v - c);
    for (uint32_t k = 0; k < c; k++) out[k] = make_uint4(((uint32_t)t << 16) | k, sb + k, rx, limit);
    __syncthreads();
    if (tid == 1023) s_carry = off + v;
    __syncthreads();
  }
  if (tid == 0) { a.work_ctr[0] = s_carry * 4u; a.work_ctr[1] = 0u; }  // four pixel-quarters per chunk
}

// Persistent workgroups over the device-built list of (tile, 64-instance chunk) items, dealt round-robin: deep tiles
// (dozens of fully active chunks) spread over the chip.
template <int NW>  // waves per workgroup = parts the tile's 256 pixels are split into
__global__ __launch_bounds__(64 * NW) void render_bwd_kernel(RenderBwdArgs a) {
  constexpr int PPW = 256 / NW;    // pixels per wave
  constexpr int RSTEP = NW;        // a wave's rows are part, part + NW, ...
  __shared__ float4 s_pa[NW][PPW];  // (T_start, Pre_start, Qb, n_contrib as float bits)
  __shared__ float4 s_pb[NW][PPW];  // (gC0, gC1, gC2, gD)
  __shared__ float s_pc[NW][PPW];   // gA
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t n_items = a.work_ctr[0];
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const size_t HW = (size_t)a.H * a.W;
  const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
  const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
  // static interleaving (a single dequeue word saturates at ~90 dequeues/us on this chip): workgroup b takes
  // chunks b, b + #workgroups, ...; its four waves are the four pixel-quarters of the chunk and meet in
  // LDS so that each (chunk, instance) still issues ONE set of atomics (float atomics are the scarce
  // resource: ~30 ns each once millions are in flight)
  __shared__ float s_red[NW - 1][64][11];  // ten partial sums and the 'has a contribution' flag
  const uint32_t n_chunks = n_items >> 2;
  const int quarter = wave;
  // a wave's pixels are rows part, part + NW, ... of the tile: every wave then sees the same mix of instances and
  // live pixels, and they meet at the barrier at about the same time
  const int spix = ((lane >> 4) * RSTEP + quarter) * 16 + (lane & 15);  // this lane's pixel while staging (lane < PPW)
  // A chunk costs about as much arithmetic as a few global round trips, and its inputs hang off a chain of them
  // (work entry -> list entry / n_contrib -> records / per-pixel state), so the chain is software-pipelined across
  // the workgroup's chunks: the work entry (self-contained: tile|chunk, checkpoint slot, list start, limit; read
  // through the scalar cache) is fetched three chunks ahead, the list entry of this lane's instance and the
  // n_contrib of this lane's pixel two ahead, and the records / per-pixel state of the NEXT chunk are requested
  // right after this chunk's arithmetic and consumed after the fold of the partial sums — BEFORE this chunk's float
  // atomics are issued: vmcnt retires in order, so loads queued behind the atomics would wait for every
  // acknowledgement (microseconds once millions are in flight) with all the workgroup's waves at the barrier.
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  typedef const u4v __attribute__((address_space(4))) c_u4;
  c_u4* work = (c_u4*)(uintptr_t)a.work;
  const uint32_t stride = gridDim.x;
  auto fetch_level2 = [&](const u4v wk, uint32_t& id_out, uint32_t& n_out) {
    const int tile = (int)(wk.x >> 16), pos = (int)(wk.x & 0xFFFFu) * 64 + lane;
    id_out = 0u; n_out = 0u;
    if (pos < (int)wk.w) id_out = a.point_list[wk.z + pos];
    const int pxi = (tile % gx) * RIGGS_TILE + (spix & 15), pyi = (tile / gx) * RIGGS_TILE + (spix >> 4);
    if (lane < PPW && pxi < a.W && pyi < a.H) n_out = a.n_contrib[(size_t)pyi * a.W + pxi];
  };
  struct Level3 {  // what a lane loads for a chunk: its instance's records, its pixel's state
    float4 xy, co, cc, acc;
    float Tn, g0, g1, g2, gD, gA, Ts, S0, S1, S2, Ds;
  };
  auto issue_level3 = [&](const u4v wk, uint32_t id, uint32_t n, Level3& r) {
    const int tile = (int)(wk.x >> 16), pos0 = (int)(wk.x & 0xFFFFu) * 64;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    r.xy = z; r.co = z; r.cc = z; r.acc = z;
    r.Tn = 0.f; r.g0 = 0.f; r.g1 = 0.f; r.g2 = 0.f; r.gD = 0.f; r.gA = 0.f; r.Ts = 1.f; r.S0 = 0.f; r.S1 = 0.f; r.S2 = 0.f; r.Ds = 0.f;
    if (pos0 + lane < (int)wk.w) { r.xy = a.xyd[id]; r.co = a.conic_o[id]; r.cc = a.rgb[id]; }
    if ((int)n > pos0) {  // (n is 0 for the lanes without a pixel and for the pixels outside the image)
      const int pxi = (tile % gx) * RIGGS_TILE + (spix & 15), pyi = (tile / gx) * RIGGS_TILE + (spix >> 4);
      const size_t pid = (size_t)pyi * a.W + pxi;
      const float* ck = a.ckpt + ((size_t)wk.y * 5) * 256;
      r.Tn = a.final_T[pid];
      r.acc = a.final_acc[pid];
      r.g0 = a.dL_dcolor[pid]; r.g1 = a.dL_dcolor[HW + pid]; r.g2 = a.dL_dcolor[2 * HW + pid];
      r.gD = a.dL_ddepth ? a.dL_ddepth[pid] : 0.f;
      r.gA = a.dL_dalpha ? a.dL_dalpha[pid] : 0.f;
      r.Ts = ck[spix]; r.S0 = ck[256 + spix]; r.S1 = ck[512 + spix]; r.S2 = ck[768 + spix]; r.Ds = ck[1024 + spix];
    }
  };
  auto stage_pixels = [&](const u4v wk, uint32_t n, const Level3& r) {  // this wave's pixels (one per lane) -> LDS
    const int pos0 = (int)(wk.x & 0xFFFFu) * 64;
    float4 pa = make_float4(1.f, 0.f, 0.f, 0.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
    float pc = 0.f;
    if ((int)n > pos0) {
      const float pre = r.g0 * r.S0 + r.g1 * r.S1 + r.g2 * r.S2 + r.gD * r.Ds + r.gA * (1.0f - r.Ts);
      const float qb = (r.g0 * r.acc.x + r.g1 * r.acc.y + r.g2 * r.acc.z + r.gD * r.acc.w + r.gA * (1.0f - r.Tn)) +
                       r.Tn * (bg0 * r.g0 + bg1 * r.g1 + bg2 * r.g2);
      pa = make_float4(r.Ts, pre, qb, __uint_as_float(n));
      pb = make_float4(r.g0, r.g1, r.g2, r.gD);
      pc = r.gA;
    }
    if (lane < PPW) { s_pa[wave][lane] = pa; s_pb[wave][lane] = pb; s_pc[wave][lane] = pc; }
  };
  const u4v wk_zero = {0u, 0u, 0u, 0u};
  u4v wk_a = wk_zero, wk_b = wk_zero, wk_c = wk_zero;  // this chunk, the next, the one after
  uint32_t id_a = 0u, n_a = 0u, id_b = 0u, n_b = 0u;
  float4 xy = make_float4(0.f, 0.f, 0.f, 0.f), co = xy, cc = xy;
  if (blockIdx.x < n_chunks) {
    wk_a = work[blockIdx.x];
    fetch_level2(wk_a, id_a, n_a);
    if (blockIdx.x + stride < n_chunks) { wk_b = work[blockIdx.x + stride]; fetch_level2(wk_b, id_b, n_b); }
    if (blockIdx.x + 2 * stride < n_chunks) wk_c = work[blockIdx.x + 2 * stride];
    Level3 r;
    issue_level3(wk_a, id_a, n_a, r);
    stage_pixels(wk_a, n_a, r);
    xy = r.xy; co = r.co; cc = r.cc;
  }
  for (uint32_t chunk_item = blockIdx.x; chunk_item < n_chunks; chunk_item += stride) {
    const u4v wk = wk_a;
    const uint32_t id = id_a;
    const unsigned long long t_begin = a.trace ? wall_clock64() : 0ull;
    const int tile = (int)(wk.x >> 16), chunk = (int)(wk.x & 0xFFFFu);
    const int limit = (int)wk.w;
    const int pos0 = chunk * 64;
    const int tx0 = (tile % gx) * RIGGS_TILE, ty0 = (tile / gx) * RIGGS_TILE;
    const int pos = pos0 + lane;
    const bool active = pos < limit;
    float m_x = 0.f, m_y = 0.f;
    bool touched = false;  // this lane's instance contributes to at least one pixel of the tile
    float a_mx = 0.f, a_my = 0.f, a_ca = 0.f, a_cb = 0.f, a_cc = 0.f, a_op = 0.f, a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f;
    // (the wave only touches its own LDS region: LDS operations of one wave are ordered, no barrier needed)
    // two pixels per iteration: their scans are independent chains that the scheduler interleaves (a lone
    // chain leaves the SIMD idle through every DPP / transcendental latency)
    for (int pl = 0; pl < PPW; pl += 2) {
      const float4 paA = s_pa[wave][pl], paB = s_pa[wave][pl + 1];
      const int nA = (int)__float_as_uint(paA.w), nB = (int)__float_as_uint(paB.w);
      if (nA <= pos0 && nB <= pos0) continue;  // wave-uniform: this chunk lies behind both pixels' last contributors
      const int pix = ((pl >> 4) * RSTEP + quarter) * 16 + (pl & 15);  // pl is even: both pixels are in the same row
      const float pfx = (float)(tx0 + (pix & 15)), pfy = (float)(ty0 + (pix >> 4));
      const float dxA = xy.x - pfx, dxB = dxA - 1.0f, dy = xy.y - pfy;
      // cheap conservative reject (alpha >= 1/255 extents, as in the forward's cull) before the exponentials
      if (a.cull && __builtin_amdgcn_ballot_w64(active && fabsf(dy) <= cc.w && fminf(fabsf(dxA), fabsf(dxB)) <= xy.w) == 0) continue;
      const float cyy = co.z * dy * dy;
      const float powA = -0.5f * (co.x * dxA * dxA + cyy) - co.y * dxA * dy;
      const float powB = -0.5f * (co.x * dxB * dxB + cyy) - co.y * dxB * dy;
      const float GA_ = fast_exp(powA), GB_ = fast_exp(powB);
      float alA = fminf(ALPHA_MAX, co.w * GA_), alB = fminf(ALPHA_MAX, co.w * GB_);
      const bool vA = active && (pos < nA) && (powA <= 0.0f) && (alA >= ALPHA_MIN);
      const bool vB = active && (pos < nB) && (powB <= 0.0f) && (alB >= ALPHA_MIN);
      if (__builtin_amdgcn_ballot_w64(vA || vB) == 0) continue;
      touched = touched || vA || vB;
      alA = vA ? alA : 0.f; alB = vB ? alB : 0.f;
      const float GA = vA ? GA_ : 0.f, GB = vB ? GB_ : 0.f;
      const float omA = 1.0f - alA, omB = 1.0f - alB;
      float scA = omA, scB = omB;
      dual_excl_prod_scan(scA, scB);
      const float TlA = paA.x * scA, TlB = paB.x * scB;
      const float4 pbA = s_pb[wave][pl], pbB = s_pb[wave][pl + 1];
      const float gAA = s_pc[wave][pl], gAB = s_pc[wave][pl + 1];
      const float wA = alA * TlA, wB = alB * TlB;
      const float kA = pbA.x * cc.x + pbA.y * cc.y + pbA.z * cc.z + pbA.w * xy.z + gAA;
      const float kB = pbB.x * cc.x + pbB.y * cc.y + pbB.z * cc.z + pbB.w * xy.z + gAB;
      const float wkA = wA * kA, wkB = wB * kB;
      float ssA = wkA, ssB = wkB;
      dual_excl_sum_scan(ssA, ssB);
      const float preA = paA.y + ssA, preB = paB.y + ssB;
      // dL/dalpha = T k - (suffix + T_final * bg.g) / (1 - alpha),  suffix = total - prefix - own
      // (no select on dLa: an invalid pair has G = 0 and every use below is multiplied by G)
      const float dLaA = TlA * kA - (paA.z - preA - wkA) * __builtin_amdgcn_rcpf(omA);
      const float dLaB = TlB * kB - (paB.z - preB - wkB) * __builtin_amdgcn_rcpf(omB);
      // raw moments of q = dL/dalpha * G; opacity and the conic factors of dL/dmean2D are applied once per chunk
      // (co is the lane's)
      const float qA = dLaA * GA, qB = dLaB * GB;
      const float qxA = qA * dxA, qxB = qB * dxB, qyA = qA * dy, qyB = qB * dy;
      m_x += qxA + qxB; m_y += qyA + qyB;
      a_ca += qxA * dxA + qxB * dxB;
      a_cb += qxA * dy + qxB * dy;
      a_cc += qyA * dy + qyB * dy;
      a_op += qA + qB;
      a_r += wA * pbA.x + wB * pbB.x; a_g += wA * pbA.y + wB * pbB.y; a_b += wA * pbA.z + wB * pbB.z;
      a_d += wA * pbA.w + wB * pbB.w;
    }
    a_mx = -co.w * (co.x * m_x + co.y * m_y);
    a_my = -co.w * (co.z * m_y + co.y * m_x);
    a_ca *= co.w; a_cb *= co.w; a_cc *= co.w;
    // ---- request the next chunk's records / pixel state, the list entry and n_contrib of the one after, and the
    // work entry after that
    const bool has_next = chunk_item + stride < n_chunks;
    Level3 nr;
    if (has_next) issue_level3(wk_b, id_b, n_b, nr);
    uint32_t id_c = 0u, n_c = 0u;
    if (chunk_item + 2 * stride < n_chunks) fetch_level2(wk_c, id_c, n_c);
    u4v wk_d = wk_zero;
    if (chunk_item + 3 * stride < n_chunks) wk_d = work[chunk_item + 3 * stride];
    // ---- fold the pixel parts
    __syncthreads();
    if (wave > 0) {
      float* r = s_red[wave - 1][lane];
      r[0] = a_mx; r[1] = a_my; r[2] = a_ca; r[3] = a_cb; r[4] = a_cc; r[5] = a_op; r[6] = a_r; r[7] = a_g; r[8] = a_b; r[9] = a_d; r[10] = touched ? 1.f : 0.f;
    }
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
      for (int q = 0; q < NW - 1; q++) {
        const float* r = s_red[q][lane];
        a_mx += r[0]; a_my += r[1]; a_ca += r[2]; a_cb += r[3]; a_cc += r[4]; a_op += r[5]; a_r += r[6]; a_g += r[7]; a_b += r[8]; a_d += r[9];
        touched = touched || (r[10] != 0.f);
      }
    }
    // ---- the next chunk's pixels go to LDS (each wave only touches its own region, and is done with it)
    if (has_next) { stage_pixels(wk_b, n_b, nr); xy = nr.xy; co = nr.co; cc = nr.cc; }
    // ---- only now the atomics, and only for the instances that reach a pixel of this tile: the lists are built
    // from 3-sigma rectangles, most of a tile's instances never get to alpha >= 1/255 inside it, and adding their
    // exact zeros cost a quarter of the kernel (the memory-side atomic units were its one saturated resource)
    if (wave == 0 && active && touched) {
      float* g = a.gacc + (size_t)id * RIGGS_GACC;
      atomicAdd(g + 0, a_mx * ddelx_dx); atomicAdd(g + 1, a_my * ddely_dy);
      atomicAdd(g + 2, -0.5f * a_ca); atomicAdd(g + 3, -a_cb); atomicAdd(g + 4, -0.5f * a_cc);
      atomicAdd(g + 5, a_op); atomicAdd(g + 6, a_r); atomicAdd(g + 7, a_g); atomicAdd(g + 8, a_b);
      if (a.dL_ddepth) atomicAdd(g + 9, a_d);
    }
    wk_a = wk_b; id_a = id_b; n_a = n_b;
    wk_b = wk_c; id_b = id_c; n_b = n_c;
    wk_c = wk_d;
    if (a.trace && threadIdx.x == 0) {  // per chunk: {start, end (100 MHz ticks), hardware id, workgroup}
      unsigned long long* tr = a.trace + (size_t)chunk_item * 4;
      tr[0] = t_begin; tr[1] = wall_clock64();
      tr[2] = (unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xFFFFu) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 0xFu) << 16);
      tr[3] = ((unsigned long long)blockIdx.x << 32) | wk.x;
    }
  }
}

int launch_render_bwd(const RenderBwdArgs& a, hipStream_t s) {
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE, gy = (a.H + RIGGS_TILE - 1) / RIGGS_TILE;
  if (gx * gy == 0) return 0;
  static const bool v1 = getenv("RIGGS_RENDER_BWD_V1") != nullptr;  // A/B switch: pixel-major reference kernel
  if (v1) {
    (void)hipMemsetAsync(a.gacc, 0, (size_t)a.n_points * RIGGS_GACC * 4, s);
    hipLaunchKernelGGL(render_bwd_v1_kernel, dim3(gx * gy), dim3(256), 0, s, a);
  }
  else {
    hipLaunchKernelGGL(render_bwd_worklist_kernel, dim3(1 + 512), dim3(1024), 0, s, a);
    static const int per_cu = getenv("RIGGS_BWD_WG_PER_CU") ? atoi(getenv("RIGGS_BWD_WG_PER_CU")) : 8;
    const int64_t max_blocks = 256 * per_cu;  // 8 workgroups of 4 waves per CU: every SIMD holds 8 pulling waves
    const unsigned blocks = (unsigned)((a.n_slots < max_blocks) ? a.n_slots : max_blocks);
    static const int nw = getenv("RIGGS_BWD_WAVES") ? atoi(getenv("RIGGS_BWD_WAVES")) : 4;
    if (nw == 8) hipLaunchKernelGGL(render_bwd_kernel<8>, dim3(blocks), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(render_bwd_kernel<4>, dim3(blocks), dim3(256), 0, s, a);
  }
  return 0;
}

}  // namespace riggs
