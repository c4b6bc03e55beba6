// Tile binning as a STABLE COUNTING SORT, replacing emit + radix sort + ranges.
//
// The Gaussians are already in depth order (stable sort of the depth bits), so the per-tile lists of
// upstream's 64-bit (tile | depth) key sort are exactly a stable partition of the emitted instances
// by tile id.  With T <= a few thousand tiles that is a counting sort:
//   bin_count   : per chunk of depth-ordered Gaussians, a tile histogram in LDS        -> table[chunk][tile]
//   bin_scan    : per tile, exclusive scan over the chunks (table rewritten in place)   -> tile_count[tile]
//   bin_scatter : per chunk: LDS cursors = tile start (own exclusive scan over the tile counts) + chunk offset + offset of
//                 the preceding waves,
//                 then every wave walks ITS Gaussians in depth order (four per step, lanes = (slot, tile of the
//                 slot's rectangle), ranks from rectangle tests) and writes point_list — order-preserving by
//                 construction.  One extra workgroup of the same launch does the once-per-frame part (bin_offsets_body):
//                 ranges[tile], slot_base[tile], R, overflow flag, the forward's work list (non-empty tiles, longest first).
// HBM traffic: N*(4+8+4) read per pass, R*4(+4) written once, 2*chunks*T*4 for the table — against
// ~R*32 B for two radix passes over (key, value) pairs plus the emit pass.
#include <stdlib.h>
#include <type_traits>
#include "raster_internal.h"
#include "color_job.h"

namespace riggs {


__device__ __forceinline__ int rect_tile(const ushort4 rc, int l, int grid_x) {
  const int w = rc.z - rc.x;
  // l / w for small non-negative ints without the integer-division sequence (exact: |error| << 0.5/w)
  const int ry = (int)(((float)l + 0.5f) * __builtin_amdgcn_rcpf((float)w));
  const int rx = l - ry * w;
  return (rc.y + ry) * grid_x + rc.x + rx;
}

__global__ __launch_bounds__(1024) void bin_count_kernel(int N, int T, int grid_x, int g_per_block,
                                                        const uint32_t* __restrict__ order,
                                                        const ushort4* __restrict__ rect,
                                                        ushort4* __restrict__ srect /* [N]: the rectangles in depth order (empty if invisible) */,
                                                        uint32_t* __restrict__ table) {
  extern __shared__ uint32_t s_hist[];  // [T]
  for (int t = threadIdx.x; t < T; t += blockDim.x) s_hist[t] = 0u;
  __syncthreads();
  const int first = blockIdx.x * g_per_block;
  const int end = min(N, first + g_per_block);
  for (int s = first + threadIdx.x; s < end; s += blockDim.x) {
    const uint32_t g = order[s];
    // (the one gather of the rectangles by depth order — a 64-byte line per Gaussian for 8 bytes; the scatter kernel reads
    // this sorted copy, coalesced.  A culled Gaussian's rectangle is empty: preprocess_fwd writes it for every Gaussian)
    const ushort4 rc = rect[g];
    srect[s] = rc;
    if (rc.z == rc.x) continue;
    for (int y = rc.y; y < rc.w; y++)
      for (int x = rc.x; x < rc.z; x++) atomicAdd(&s_hist[y * grid_x + x], 1u);
  }
  __syncthreads();
  uint32_t* row = table + (size_t)blockIdx.x * T;
  for (int t = threadIdx.x; t < T; t += blockDim.x) row[t] = s_hist[t];
}

// per tile: exclusive scan over the chunks, in place; tile_count[t] = total.
// Workgroup = 64 tiles x 16 chunk segments: every thread sums its segment (loads independent and
// coalesced across the 64 tiles), LDS prefix over the 16 segments, then the exclusive prefixes are
// written back — no serial walk over hundreds of chunks.
#define SCAN_SEG 16
#define SCAN_KEEP 32
__device__ __forceinline__ void bin_scan_body(int T, int n_chunks, uint32_t* __restrict__ table,
                                              uint32_t* __restrict__ tile_count) {
  __shared__ uint32_t s_seg[SCAN_SEG][64];
  const int tl = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tl;
  const int per = (n_chunks + SCAN_SEG - 1) / SCAN_SEG;
  const int b0 = seg * per, b1 = min(n_chunks, b0 + per);
  uint32_t sum = 0;
  if (per <= SCAN_KEEP) {
    // the usual case: the whole segment stays in registers (one batch of independent loads, no re-read)
    uint32_t v[SCAN_KEEP];
#pragma unroll
    for (int i = 0; i < SCAN_KEEP; i++) v[i] = (t < T && b0 + i < b1) ? table[(size_t)(b0 + i) * T + t] : 0u;
#pragma unroll
    for (int i = 0; i < SCAN_KEEP; i++) sum += v[i];
    s_seg[seg][tl] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int s2 = 0; s2 < seg; s2++) run += s_seg[s2][tl];
    if (t < T) {
#pragma unroll
      for (int i = 0; i < SCAN_KEEP; i++) {
        if (b0 + i < b1) table[(size_t)(b0 + i) * T + t] = run;
        run += v[i];
      }
      if (seg == SCAN_SEG - 1) tile_count[t] = run;
    }
    return;
  }
  if (t < T) for (int b = b0; b < b1; b++) sum += table[(size_t)b * T + t];
  s_seg[seg][tl] = sum;
  __syncthreads();
  uint32_t run = 0;
  for (int s2 = 0; s2 < seg; s2++) run += s_seg[s2][tl];
  if (t < T) {
    for (int b = b0; b < b1; b++) { const uint32_t c = table[(size_t)b * T + t]; table[(size_t)b * T + t] = run; run += c; }
    if (seg == SCAN_SEG - 1) tile_count[t] = run;
  }
}
__global__ __launch_bounds__(1024) void bin_scan_kernel(int T, int n_chunks, uint32_t* __restrict__ table,
                                                        uint32_t* __restrict__ tile_count) {
  bin_scan_body(T, n_chunks, table, tile_count);
}

// exclusive scan over the tiles (one workgroup — the extra, last one of bin_scatter_kernel's grid): ranges, checkpoint slot
// bases, cleared per-tile words, R and the overflow flag, and the forward's work list: the non-empty tiles, longest lists first
// (by floor(log2(length))), the empty ones in a list of their own.  In front of them all, fwd_ctr[1] entries: the tiles the
// forward composites WIDE (render.hip) — the ones whose walk went o.wide_min instances deep in the PREVIOUS frame of this arena
// (o.walk_hist: max n_contrib per tile, written by the forward; how deep a walk goes is a property of the scene and the view —
// a list's length says nothing about it: the longest lists of the bench scene, 48 000 instances, saturate within 800) and
// whose list is that long now; the first o.wide_tiles of them in tile order.  Without a history (the stamp behind the last
// tile — a function of the tile and Gaussian counts, o.hist_stamp — is missing: the arena's owner zero-fills the words when
// it allocates the arena or changes the scene / image size, riggs_raster_binning_reset_history) or with cfg.deterministic
// (wide_tiles = 0) no tile is wide.
__device__ void bin_offsets_body(int T, int64_t cap, const uint32_t* __restrict__ tile_count, const BinOut o,
                                 const uint32_t* __restrict__ total_src = nullptr, int n_total_src = 0) {
  uint2* __restrict__ ranges = o.ranges;
  uint32_t* __restrict__ slot_base = o.slot_base;
  uint32_t* __restrict__ tile_max = o.tile_max;
  uint32_t* __restrict__ counters = o.counters;
  uint32_t* __restrict__ fwd_items = o.fwd_items;
  uint32_t* __restrict__ fwd_empty = o.fwd_empty;
  uint32_t* __restrict__ fwd_ctr = o.fwd_ctr;
  __shared__ uint32_t s_wave[16], s_wwave[16];
  __shared__ uint32_t s_carry, s_wcarry;
  __shared__ uint32_t s_hist[33], s_cur[33], s_nempty;  // (class 32: the wide tiles)
  const int nthr = (int)blockDim.x;  // <= 1024
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int s_slot;
  if (tid == 0) { s_carry = 0u; s_nempty = 0u; s_wcarry = 0u; s_slot = RIGGS_HIST_SLOTS; }
  if (tid < 33) s_hist[tid] = 0u;
  // ... nor is there a history when the VIEW is another one: a trainer draws a new camera every iteration, and the walk depths of
  // a frame seen from elsewhere pick the wrong tiles for the wide form (eight cameras on a circle: 0.395 ms with the previous
  // camera's history, 0.388 without any).  Histories are kept PER VIEW (raster_internal.h): the slot whose view matrix is this
  // frame's — no entry further off than o.view_tol (riggs_set_option("fwd_hist_view_tol"), default 0.2: a static camera, an orbit
  // of a few degrees per frame, the same training view coming back) — or, without one, the next slot round robin, whose history
  // starts with this frame.
  const uint32_t slot_words = riggs_hist_slot_words((uint32_t)T);
  uint32_t* const hist_hdr = o.walk_hist;
  __syncthreads();
  if (o.viewmatrix) {
    for (int sl = tid; sl < RIGGS_HIST_SLOTS; sl += nthr) {
      const uint32_t* hs = hist_hdr + RIGGS_HIST_HDR + (size_t)sl * slot_words;
      bool ok = hs[T] == o.hist_stamp;
      for (int e = 0; e < 16 && ok; e++) ok = fabsf(o.viewmatrix[e] - __uint_as_float(hs[T + 2 + e])) <= o.view_tol;  // (NaN: no)
      if (ok) atomicMin(&s_slot, sl);
    }
  } else if (tid == 0 && hist_hdr[RIGGS_HIST_HDR + T] == o.hist_stamp) s_slot = 0;
  __syncthreads();
  const bool slot_found = s_slot < RIGGS_HIST_SLOTS;
  const uint32_t cursor = hist_hdr[0];
  const int slot = slot_found ? s_slot : (o.viewmatrix ? (int)(cursor % RIGGS_HIST_SLOTS) : 0);
  uint32_t* __restrict__ walk_hist = hist_hdr + RIGGS_HIST_HDR + (size_t)slot * slot_words;
  const bool have_hist = o.wide_tiles > 0u && slot_found;
  __syncthreads();  // (every thread has read the cursor)
  if (tid == 0) {
    hist_hdr[1] = (uint32_t)slot;  // where the forward writes this frame's depths
    if (!slot_found && o.viewmatrix) hist_hdr[0] = cursor + 1u;
  }
  uint32_t my_len[8];  // list lengths of this thread's tiles (the first 8 passes; beyond that they are re-read)
#pragma unroll
  for (int k = 0; k < 8; k++) my_len[k] = 0u;
  for (int base = 0, pass = 0; base < T; base += nthr, pass++) {
    const int t = base + tid;
    const uint32_t c = (t < T) ? tile_count[t] : 0u;
    // inclusive scan inside the wave
    uint32_t v = c;
    for (int o2 = 1; o2 < 64; o2 <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)v, o2);
      if (lane >= o2) v += u;
    }
    // candidates for the wide form, counted in tile order
    const uint32_t prev = (have_hist && t < T) ? walk_hist[t] : 0u;
    const bool cand = prev >= o.wide_min && prev < 0x80000000u && c >= o.wide_min;
    const uint64_t cmask = __builtin_amdgcn_ballot_w64(cand);
    if (lane == 63) { s_wave[wave] = v; s_wwave[wave] = (uint32_t)__builtin_popcountll(cmask); }
    __syncthreads();
    uint32_t wave_off = 0, wrank = s_wcarry + (uint32_t)__builtin_popcountll(cmask & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) { wave_off += s_wave[w]; wrank += s_wwave[w]; }
    const uint32_t carry = s_carry;
    const uint32_t start = carry + wave_off + v - c;
    if (t < T) {
      // clamp to the arena: on overflow (flagged below) the frame is invalid but every access stays in bounds
      const uint32_t lo = (uint32_t)min((int64_t)start, cap), hi = (uint32_t)min((int64_t)start + c, cap);
      // empty tiles get (0, 0) like upstream's identifyTileRanges (whose ranges buffer is zeroed before)
      ranges[t] = (hi > lo) ? make_uint2(lo, hi) : make_uint2(0u, 0u);
      tile_max[t] = 0u;  // atomicMax target of the forward
      tile_max[T + 1 + t] = 0u;  // ... and the arrival ticket of the tile's forward blocks
      slot_base[t] = (lo >> 6) + (uint32_t)t;
      // forward work queue, step 1: histogram of floor(log2(list length)); the empty tiles go to their own list
      const bool wide = cand && wrank < o.wide_tiles && hi - lo >= o.wide_min;
      walk_hist[t] = wide ? 0x80000000u : 0u;  // (the forward writes this frame's depth over it; an empty tile's stays 0)
      // the order of the work list is by how long a tile's blocks are EXPECTED to walk: the depth of the previous frame's walk
      // (doubled: the view moves) where the arena has a history, capped by the list; the list's length alone otherwise
      uint32_t key = hi - lo;
      if (have_hist && prev > 0u && prev < 0x80000000u && pass < 8) key = min(key, 2u * prev + 64u);  // (pass >= 8: emitted by length below)
      if (hi > lo) atomicAdd(&s_hist[wide ? 32 : 31 - __builtin_clz(key)], 1u);
      else fwd_empty[atomicAdd(&s_nempty, 1u)] = (uint32_t)t;
#pragma unroll
      for (int k = 0; k < 8; k++) if (pass == k) my_len[k] = key;
    }
    __syncthreads();
    if (tid == nthr - 1) { s_carry = carry + wave_off + v; s_wcarry = wrank + (cand ? 1u : 0u); }
    __syncthreads();
  }
  if (tid == 0) walk_hist[T] = o.hist_stamp;
  if (tid < 16 && o.viewmatrix && !slot_found) walk_hist[T + 2 + tid] = __float_as_uint(o.viewmatrix[tid]);  // (a matched slot keeps ITS view: no drift)
  if (total_src) {
    // (the grouped binning counts only the instances that fit the arena per tile: the true total comes from its group counts)
    __shared__ uint32_t s_tot;
    if (tid == 0) s_tot = 0u;
    __syncthreads();
    uint32_t part = 0;
    for (int k = tid; k < n_total_src; k += nthr) part += total_src[k];
    for (int o2 = 32; o2 > 0; o2 >>= 1) part += (uint32_t)__shfl_xor((int)part, o2);
    if (lane == 0) atomicAdd(&s_tot, part);
    __syncthreads();
    if (tid == 0) s_carry = s_tot;
    __syncthreads();
  }
  if (tid == 0) {
    const uint32_t R = s_carry;
    ranges[T] = make_uint2(0u, 0u);
    tile_max[T] = 0u;
    slot_base[T] = ((uint32_t)min((int64_t)R, cap) >> 6) + (uint32_t)T;
    counters[0] = R;
    counters[1] = (counters[1] & 2u) | (((int64_t)R > cap) ? 1u : 0u);  // (bit 1: the depth sort's third pass failed to synchronise)
  }
  __syncthreads();
  if (tid < 32) {
    const uint32_t n_wide = s_hist[32];
    const uint32_t h = s_hist[31 - tid];  // lane k: length class 31 - k
    uint32_t v = h;
    for (int o2 = 1; o2 < 32; o2 <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)v, o2);
      if (tid >= o2) v += u;
    }
    s_cur[31 - tid] = n_wide + v - h;
    if (tid == 0) s_cur[32] = 0u;
    if (tid == 31) { fwd_ctr[0] = n_wide + v; fwd_ctr[1] = n_wide; fwd_ctr[2] = s_nempty; fwd_ctr[64] = 0u; }  // ([64]: backward work-list size, on its own cache line)
  }
  __syncthreads();
  auto emit = [&](uint32_t t, uint32_t len) { fwd_items[atomicAdd(&s_cur[(walk_hist[t] >> 31) ? 32 : 31 - __builtin_clz(len)], 1u)] = t; };
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int t = k * nthr + tid;
    if (t < T && my_len[k] > 0u) emit((uint32_t)t, my_len[k]);
  }
  for (int t = 8 * nthr + tid; t < T; t += nthr) {  // (same thread that wrote ranges[t] above; beyond 8 passes: by length)
    const uint2 r = ranges[t];
    if (r.y > r.x) emit((uint32_t)t, r.y - r.x);
  }
}

// Scatter.  LDS: s_base[T] (u32 absolute start of this block's segment in each tile) and
// s_rel[W][T] (u16 offsets of each wave's sub-segment, then used as that wave's running cursor).
__global__ __launch_bounds__(1024) void bin_scatter_kernel(int N, int T, int grid_x, int64_t cap, int g_per_block, int g_per_wave,
                                                          const uint32_t* __restrict__ order,
                                                          const ushort4* __restrict__ srect /* rectangles in depth order (bin_count_kernel) */,
                                                          const uint32_t* __restrict__ table,
                                                          const uint32_t* __restrict__ tile_count,
                                                          uint32_t* __restrict__ point_list,
                                                          uint32_t* __restrict__ tile_keys, const BinOut out,
                                                          const int n_main, const ColorJob* __restrict__ job_rec, const int job_gpb) {
  extern __shared__ uint32_t s_mem[];
  if ((int)blockIdx.x == n_main) {  // the extra workgroup: what used to be a single-workgroup launch of its own
    bin_offsets_body(T, cap, tile_count, out);
    return;
  }
  if ((int)blockIdx.x > n_main) {
    // the workgroups behind: the SH colours of the frame (color_job.h), job_gpb Gaussians each; the record is preprocess_fwd's
    // (N = 0: it evaluated the colours itself).  They start behind the sort's own workgroups (block index order).
    const ColorJob job = *job_rec;
    color_block(job, (int)blockIdx.x - n_main - 1, job_gpb, reinterpret_cast<float*>(s_mem));
    return;
  }
  const int W = blockDim.x >> 6;
  uint32_t* s_base = s_mem;                                            // [T]
  const int Tpad = (T + 1) & ~1;
  unsigned short* s_rel = reinterpret_cast<unsigned short*>(s_mem + T);  // [W][Tpad]
  uint32_t* s_rel32 = s_mem + T;                                        // same storage as packed pairs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < W * (Tpad >> 1); e += blockDim.x) s_rel32[e] = 0u;
  __syncthreads();
  // a wave walks g_per_wave Gaussians in batches of 64 (one batch in the headline configuration; more when N is large,
  // so that the O(T) set-up of a workgroup — cursor tables, tile bases, its row of the chunk table — is paid per thousands
  // of Gaussians and the chunk table stays small: bin_plan)
  const int first = blockIdx.x * g_per_block + wave * g_per_wave;
  const int end = min(N, min(first + g_per_wave, (blockIdx.x + 1) * g_per_block));
  auto load_batch = [&](int s0, uint32_t& g, uint32_t& n, ushort4& rc) {
    const int s = s0 + lane;
    g = 0u; n = 0u; rc = make_ushort4(0, 0, 0, 0);
    if (s < end) {
      g = order[s];
      rc = srect[s];
      n = (uint32_t)((int)(rc.z - rc.x) * (int)(rc.w - rc.y));
    }
  };
  // (i) per-wave tile histogram (16-bit counters packed in pairs; a wave adds at most g_per_wave <= 1024 per tile)
  uint32_t nx_g = 0u, nx_n = 0u;          // the first batch stays in registers for the walk below
  ushort4 nx_rc = make_ushort4(0, 0, 0, 0);
  for (int s0 = first; s0 < end; s0 += 64) {
    uint32_t g, n;
    ushort4 rc;
    load_batch(s0, g, n, rc);
    if (s0 == first) { nx_g = g; nx_n = n; nx_rc = rc; }
    if (n) {
      uint32_t* hist = s_rel32 + (size_t)wave * (Tpad >> 1);
      for (int y = rc.y; y < rc.w; y++)
        for (int x = rc.x; x < rc.z; x++) {
          const int t = y * grid_x + x;
          atomicAdd(&hist[t >> 1], 1u << (16 * (t & 1)));
        }
    }
  }
  __syncthreads();
  // (ii) counts -> offsets of the waves inside the block's segment; absolute base of the segment = start of the tile's
  // list (every workgroup scans the tile counts itself: T words from L2, instead of a launch that does it once) + the
  // instances of the earlier chunks
  const uint32_t* row = table + (size_t)blockIdx.x * T;
  {
    __shared__ uint32_t s_wsum[16];
    __shared__ uint32_t s_run;
    if (tid == 0) s_run = 0u;
    __syncthreads();
    // (the counts and the chunk's row for FOUR rounds of the scan are asked for at once: taken a round at a time — a load, the
    // scan, a barrier, the second load — the workgroup stood through eight dependent round trips at 2 500 tiles)
    for (int base0 = 0; base0 < T; base0 += 4 * blockDim.x) {
      uint32_t cq[4], rq[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int t = base0 + k * (int)blockDim.x + tid;
        cq[k] = (t < T) ? tile_count[t] : 0u;
        rq[k] = (t < T) ? row[t] : 0u;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int base = base0 + k * (int)blockDim.x;
        if (base >= T) break;
        const int t = base + tid;
        const uint32_t c = cq[k];
        uint32_t v = c;
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t u = (uint32_t)__shfl_up((int)v, o);
          if (lane >= o) v += u;
        }
        if (lane == 63) s_wsum[wave] = v;
        __syncthreads();
        uint32_t off = s_run;
        for (int w = 0; w < wave; w++) off += s_wsum[w];
        if (t < T) s_base[t] = off + v - c + rq[k];
        __syncthreads();
        if (tid == (int)blockDim.x - 1) s_run = off + v;
        __syncthreads();
      }
    }
  }
  for (int t = tid; t < T; t += blockDim.x) {
    uint32_t run = 0;
    for (int w = 0; w < W; w++) {
      const unsigned short c = s_rel[(size_t)w * Tpad + t];
      s_rel[(size_t)w * Tpad + t] = (unsigned short)run;
      run += c;
    }
  }
  __syncthreads();
  // (iii) ordered walk.  Four Gaussians per step — lanes 4q .. 4q+3 of the wave, q = 0 .. 15 — when their rectangles
  // have at most 16 tiles each (all but ~0 % of the bench scene): lane = (slot, tile of the slot's rectangle).  The
  // order inside a tile's segment must be the Gaussians' order, so a lane's position is the cursor plus the number
  // of EARLIER slots of the step whose rectangle contains the tile (rectangle tests against wave-uniform bounds), and
  // the LAST slot that contains the tile advances the cursor — deterministic by construction, no atomics.  A group
  // with a larger rectangle falls back to one Gaussian per step (lanes = tiles).  Fixed groups keep the loop free of
  // 64-bit mask arithmetic: the CU's one scalar unit was what bounded the walk (7.8 M scalar instructions).
  unsigned short* cur = s_rel + (size_t)wave * Tpad;
  const int slot4 = lane >> 4, l16 = lane & 15;
  for (int s0 = first; s0 < end; s0 += 64) {
    const uint32_t cur_g = nx_g, cur_n = nx_n;
    const ushort4 cur_rc = nx_rc;
    if (s0 + 64 < end) load_batch(s0 + 64, nx_g, nx_n, nx_rc);  // (the next batch's loads fly during this one's walk)
    const uint64_t live = __builtin_amdgcn_ballot_w64(cur_n != 0u);
    const uint64_t large = __builtin_amdgcn_ballot_w64(cur_n > 16u);
    const uint64_t huge = __builtin_amdgcn_ballot_w64(cur_n > 32u);
    const int pk_xy = (int)cur_rc.x | ((int)cur_rc.y << 16), pk_zw = (int)cur_rc.z | ((int)cur_rc.w << 16);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      if (((live >> (4 * q)) & 0xFull) == 0ull) continue;
      if (((large >> (4 * q)) & 0xFull) != 0ull && ((huge >> (4 * q)) & 0xFull) == 0ull) {
        // rectangles of 17 .. 32 tiles (the usual size at 1080p with millions of Gaussians): TWO Gaussians per step, lanes =
        // (slot, tile) with 32 tiles per slot, the same rank-by-earlier-slot rule
        const int slot2 = lane >> 5, l32 = lane & 31;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int sa = 4 * q + 2 * h, sb = sa + 1;
          const int a0 = __builtin_amdgcn_readlane(pk_xy, sa), b0 = __builtin_amdgcn_readlane(pk_zw, sa);
          const int a1 = __builtin_amdgcn_readlane(pk_xy, sb), b1 = __builtin_amdgcn_readlane(pk_zw, sb);
          const int axy = slot2 ? a1 : a0, bzw = slot2 ? b1 : b0;
          const uint32_t g = (uint32_t)__shfl((int)cur_g, sa + slot2);
          const int rx0 = axy & 0xFFFF, ry0 = (int)((uint32_t)axy >> 16), rx1 = bzw & 0xFFFF, ry1 = (int)((uint32_t)bzw >> 16);
          const int w = rx1 - rx0;
          const int ry = (int)(((float)l32 + 0.5f) * __builtin_amdgcn_rcpf((float)w));  // (as rect_tile)
          const int x = rx0 + (l32 - ry * w), y = ry0 + ry;
          const bool valid = l32 < w * (ry1 - ry0);
          const bool in0 = x >= (a0 & 0xFFFF) && y >= (int)((uint32_t)a0 >> 16) && x < (b0 & 0xFFFF) && y < (int)((uint32_t)b0 >> 16);
          const bool in1 = x >= (a1 & 0xFFFF) && y >= (int)((uint32_t)a1 >> 16) && x < (b1 & 0xFFFF) && y < (int)((uint32_t)b1 >> 16);
          const int before = (slot2 && in0) ? 1 : 0;
          const bool last = slot2 || !in1;  // no later slot of the step holds this tile
          if (valid) {
            const int t = y * grid_x + x;
            const unsigned short rel = cur[t];
            if (last) cur[t] = (unsigned short)(rel + before + 1);
            const int64_t pos = (int64_t)s_base[t] + rel + before;
            if (pos < cap) { point_list[pos] = g; if (tile_keys) tile_keys[pos] = (uint32_t)t; }
          }
        }
        continue;
      }
      if (((large >> (4 * q)) & 0xFull) != 0ull) {
        // one Gaussian per step
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int src = 4 * q + j;
          const int n = __builtin_amdgcn_readlane((int)cur_n, src);
          if (n == 0) continue;
          const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)cur_g, src);
          const int r0 = __builtin_amdgcn_readlane(pk_xy, src), r1 = __builtin_amdgcn_readlane(pk_zw, src);
          ushort4 rc;
          rc.x = (unsigned short)(r0 & 0xFFFF); rc.y = (unsigned short)((uint32_t)r0 >> 16);
          rc.z = (unsigned short)(r1 & 0xFFFF); rc.w = (unsigned short)((uint32_t)r1 >> 16);
          for (int l = lane; l < n; l += 64) {
            const int t = rect_tile(rc, l, grid_x);
            const unsigned short rel = cur[t];
            cur[t] = rel + 1;
            const int64_t pos = (int64_t)s_base[t] + rel;
            if (pos < cap) { point_list[pos] = g; if (tile_keys) tile_keys[pos] = (uint32_t)t; }
          }
        }
        continue;
      }
      // wave-uniform rectangles of the four (an invisible Gaussian has the empty rectangle 0,0,0,0)
      const int a0 = __builtin_amdgcn_readlane(pk_xy, 4 * q), b0 = __builtin_amdgcn_readlane(pk_zw, 4 * q);
      const int a1 = __builtin_amdgcn_readlane(pk_xy, 4 * q + 1), b1 = __builtin_amdgcn_readlane(pk_zw, 4 * q + 1);
      const int a2 = __builtin_amdgcn_readlane(pk_xy, 4 * q + 2), b2 = __builtin_amdgcn_readlane(pk_zw, 4 * q + 2);
      const int a3 = __builtin_amdgcn_readlane(pk_xy, 4 * q + 3), b3 = __builtin_amdgcn_readlane(pk_zw, 4 * q + 3);
      const int axy = (slot4 == 0) ? a0 : (slot4 == 1) ? a1 : (slot4 == 2) ? a2 : a3;
      const int bzw = (slot4 == 0) ? b0 : (slot4 == 1) ? b1 : (slot4 == 2) ? b2 : b3;
      const uint32_t g = (uint32_t)__shfl((int)cur_g, 4 * q + slot4);
      const int rx0 = axy & 0xFFFF, ry0 = (int)((uint32_t)axy >> 16), rx1 = bzw & 0xFFFF, ry1 = (int)((uint32_t)bzw >> 16);
      const int w = rx1 - rx0;
      const int ry = (int)(((float)l16 + 0.5f) * __builtin_amdgcn_rcpf((float)w));  // (as rect_tile)
      const int x = rx0 + (l16 - ry * w), y = ry0 + ry;
      const bool valid = l16 < w * (ry1 - ry0);
      auto inside = [&](int axy_, int bzw_) {
        return x >= (axy_ & 0xFFFF) && y >= (int)((uint32_t)axy_ >> 16) && x < (bzw_ & 0xFFFF) && y < (int)((uint32_t)bzw_ >> 16);
      };
      const int in0 = inside(a0, b0) ? 1 : 0, in1 = inside(a1, b1) ? 1 : 0, in2 = inside(a2, b2) ? 1 : 0, in3 = inside(a3, b3) ? 1 : 0;
      const int before = (slot4 > 0 ? in0 : 0) + (slot4 > 1 ? in1 : 0) + (slot4 > 2 ? in2 : 0);
      const int after = (slot4 < 1 ? in1 : 0) + (slot4 < 2 ? in2 : 0) + (slot4 < 3 ? in3 : 0);
      if (valid) {
        const int t = y * grid_x + x;
        const unsigned short rel = cur[t];
        if (after == 0) cur[t] = (unsigned short)(rel + before + 1);
        const int64_t pos = (int64_t)s_base[t] + rel + before;
        if (pos < cap) { point_list[pos] = g; if (tile_keys) tile_keys[pos] = (uint32_t)t; }
      }
    }
  }
}

// =====================================================================================================
// GROUPED (two-level) tile sort: for tile grids beyond 25 600 tiles (T > BIN_GROUPED_MIN_T; e.g. 3840 x 2160 = 32 400
// tiles), where the direct counting sort above no longer fits (it keeps a cursor per tile and wave in LDS), and for many
// Gaussians over many tiles (bin_grouped below), where it is the faster one.  The same stable sort in TWO levels:
//   (1) by GROUP of 8 horizontally adjacent tiles — gbin_count (per chunk and wave: instances per group), bin_scan (over the
//       chunks), gbin_scatter (the ordered walk) — writing instance words  gaussian | tile-in-group << 29  into a scratch
//       list (the checkpoint area, which the compositing only fills later);
//   (2) every group's contiguous segment — its tiles' lists back to back — is partitioned stably by the three tile bits, in
//       parts of BIN_PART instances (one workgroup each): gbin_tcount (per part and tile), gbin_tscatter.
// Order: groups by (row, column block) = ascending tile id blocks; inside a group the level-1 order (depth) is kept by both
// levels: the same list as the direct sort, bit for bit (tests/test_gpu_raster.py: 32 400 tiles against the oracle's key
// sort; tests/test_gpu_configs.py: C4 and C5 take this path; the whole raster suite also passes with riggs_set_option("bin_grouped", 1)).
// Why it pays at 2 M Gaussians / 8160 tiles / 38.8 M instances (0.40 ms against 0.70 ms): the direct sort's walk is 38.8 M
// 4-byte stores to as many different lines; here level 1 writes a SPAN — a rectangle's columns inside one group, up to 32
// bytes — with one to four stores, and level 2 moves whole parts through LDS with coalesced loads and stores.  What it took
// (first version: 66 + 30 + 301 + 71 + 329 us, slower than the direct sort — profiles/round3_C5_grouped_binning_timeline.txt;
// now 72 + 6 + 175 + 48 + 99 — profiles/round3_final_C5_graph_timeline.txt):
//   * level 1 counted twice (per chunk in gbin_count, per wave in gbin_scatter: LDS atomics on a few hot groups, 27 - 45 us
//     per workgroup): gbin_count counts per wave and hands the waves' starts over;
//   * level 1's walk is bound by instruction issue, not by memory: lanes = spans instead of tiles, eight Gaussians per step,
//     their order kept by letting the slots claim the group cursors one after the other instead of rectangle tests;
//   * level 2 found its part by scanning all group counts in every workgroup (3.5 us, 5 700 times, twice): a part table
//     written once by an extra workgroup of gbin_scatter; extra workgroups go FIRST in their grids (the last of 8 000
//     workgroups starts when the kernel is almost over);
//   * level 2 ranked 64 instances at a time with ballots and three barriers per 512: now a thread takes 16 consecutive
//     instances with packed counters, and the part goes through LDS in and out; its registers were what bound it last (128:
//     two workgroups per CU, phases of 3 - 5 us latency each that did not overlap; 50: four, 157 -> 99 us);
//   * bin_scan over 977 chunks ran its slow path on 16 workgroups: <= 512 chunks (30 -> 6 us).
// =====================================================================================================
#define BIN_GROUPED_MIN_T 25600
#define BIN_GROUPED_AUTO_T 4096
#define BIN_GROUPED_AUTO_N 500000
#define BIN_PART 8192  // instances per workgroup of the second level
#ifndef GTS_THREADS
#define GTS_THREADS 512  // threads of gbin_tscatter_kernel
#define GTS_WAVES 8     // (its waves per SIMD: four workgroups per CU, LDS 4 x 35.5 KB)
#endif

struct __attribute__((packed, aligned(4))) GU4 { uint32_t a, b, c, d; };  // (dword-aligned 16- and 8-byte stores)
struct __attribute__((packed, aligned(4))) GU2 { uint32_t a, b; };

// (1a) per chunk of depth-ordered Gaussians: instances per group, counted PER WAVE (wave w of the scatter kernel below walks
// the chunk's Gaussians [w * g_per_wave, (w + 1) * g_per_wave) — the same split here): the chunk's totals go to the table,
// the waves' starts inside the chunk's run of every group to wave_start[chunk][wave][group] (u16), so that the scatter kernel
// does not count again (its own histogram pass cost it 27 - 45 us per workgroup at 2 M Gaussians: LDS atomics on a few hot
// groups).  Block 0 also clears the per-tile counts of level 2.  LDS: [W][Gpad] u16, packed pairs while counting.
// (Lane = Gaussian, looping over its spans.  Lanes = spans as in the scatter kernel's walk — one atomic instruction per
// eight Gaussians — took 138 us instead of 73: the spans of eight depth-adjacent Gaussians hit the same hot groups in the
// same instruction, and the LDS serialises equal addresses.)
__global__ __launch_bounds__(1024) void gbin_count_kernel(int N, int T, int G, int gxg, int g_per_block, int g_per_wave,
                                                         const uint32_t* __restrict__ order,
                                                         const ushort4* __restrict__ rect, ushort4* __restrict__ srect,
                                                         uint32_t* __restrict__ table,
                                                         uint32_t* __restrict__ wave_start, uint32_t* __restrict__ tile_count) {
  extern __shared__ uint32_t s_rel32[];
  const int W = blockDim.x >> 6;
  const int Gpad = (G + 1) & ~1;
  unsigned short* s_rel = reinterpret_cast<unsigned short*>(s_rel32);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < W * (Gpad >> 1); e += blockDim.x) s_rel32[e] = 0u;
  if (blockIdx.x == 0) for (int t = tid; t <= T; t += blockDim.x) tile_count[t] = 0u;
  __syncthreads();
  const int first = blockIdx.x * g_per_block + wave * g_per_wave;
  const int end = min(N, min(first + g_per_wave, (blockIdx.x + 1) * g_per_block));
  uint32_t* hist = s_rel32 + (size_t)wave * (Gpad >> 1);
  for (int s0 = first; s0 < end; s0 += 64) {
    const int s = s0 + lane;
    if (s < end) {
      const uint32_t g = order[s];
      const ushort4 rc = rect[g];
      srect[s] = rc;  // (the rectangles in depth order: see bin_count_kernel)
      if (rc.z > rc.x) {
        for (int y = rc.y; y < rc.w; y++)
          for (int xg = rc.x >> 3; xg <= (rc.z - 1) >> 3; xg++) {
            const int gi = y * gxg + xg;
            atomicAdd(&hist[gi >> 1], (uint32_t)(min((int)rc.z, xg * 8 + 8) - max((int)rc.x, xg * 8)) << (16 * (gi & 1)));
          }
      }
    }
  }
  __syncthreads();
  uint32_t* row = table + (size_t)blockIdx.x * G;
  for (int g = tid; g < G; g += blockDim.x) {
    uint32_t run = 0;
    for (int w = 0; w < W; w++) {
      const unsigned short c = s_rel[(size_t)w * Gpad + g];
      s_rel[(size_t)w * Gpad + g] = (unsigned short)run;
      run += c;
    }
    row[g] = run;
  }
  __syncthreads();
  uint32_t* ws = wave_start + (size_t)blockIdx.x * W * (Gpad >> 1);
  for (int e = tid; e < W * (Gpad >> 1); e += blockDim.x) ws[e] = s_rel32[e];
}

__device__ void gbin_parts_body(int G, int64_t cap, uint32_t n_parts, const uint32_t* __restrict__ group_count, uint4* __restrict__ part_tab);

// (1b) ordered scatter into the groups' segments.  LDS: s_base[G] + per-wave cursors [W][G] (u32)
__global__ __launch_bounds__(1024) void gbin_scatter_kernel(int N, int G, int gxg, int64_t cap, int g_per_block, int g_per_wave,
                                                           const uint32_t* __restrict__ order, const ushort4* __restrict__ srect,
                                                           const uint32_t* __restrict__ table,
                                                           const uint32_t* __restrict__ wave_start,
                                                           const uint32_t* __restrict__ group_count, uint32_t* __restrict__ inter,
                                                           uint4* __restrict__ part_tab, uint32_t n_parts) {
  if (blockIdx.x == 0) {  // (the extra workgroup, first in the grid: the second level's parts)
    gbin_parts_body(G, cap, n_parts, group_count, part_tab);
    return;
  }
  const int chunk = (int)blockIdx.x - 1;
  extern __shared__ uint32_t s_mem[];
  const int W = blockDim.x >> 6;
  uint32_t* s_base = s_mem;         // [G]     start of this chunk's run in every group's segment
  uint32_t* s_cur = s_mem + G;      // [W][G]  the waves' cursors: ABSOLUTE places in the scratch list
  const int Gpad = (G + 1) & ~1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int first = chunk * g_per_block + wave * g_per_wave;
  const int end = min(N, min(first + g_per_wave, (chunk + 1) * g_per_block));
  // (ii) start of this chunk's run in every group's segment: exclusive scan of the group counts + the earlier chunks
  const uint32_t* row = table + (size_t)chunk * G;
  {
    __shared__ uint32_t s_wsum[16];
    __shared__ uint32_t s_run;
    if (tid == 0) s_run = 0u;
    __syncthreads();
    for (int base = 0; base < G; base += blockDim.x) {
      const int g = base + tid;
      const uint32_t c = (g < G) ? group_count[g] : 0u;
      uint32_t v = c;
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)v, o);
        if (lane >= o) v += u;
      }
      if (lane == 63) s_wsum[wave] = v;
      __syncthreads();
      uint32_t off = s_run;
      for (int w = 0; w < wave; w++) off += s_wsum[w];
      if (g < G) s_base[g] = off + v - c + row[g];
      __syncthreads();
      if (tid == (int)blockDim.x - 1) s_run = off + v;
      __syncthreads();
    }
  }
  // (i) cursor of wave w in group g = the chunk's start + the wave's start inside the chunk's run (counted by gbin_count_kernel)
  {
    const unsigned short* ws = reinterpret_cast<const unsigned short*>(wave_start + (size_t)chunk * W * (Gpad >> 1));
    for (int w = 0; w < W; w++)
      for (int g = tid; g < G; g += blockDim.x) s_cur[(size_t)w * G + g] = s_base[g] + ws[(size_t)w * Gpad + g];
  }
  __syncthreads();
  // (iii) ordered walk.  The unit is a SPAN: the columns of a Gaussian's rectangle that fall into one group, in one tile
  // row — up to eight instances that go to consecutive places of the group's segment.  Lane = (slot, span): eight Gaussians
  // share a step when each has at most 8 spans (a 5 x 4 rectangle has 4 - 8), four with at most 16; a larger one takes a
  // step of its own (lanes = its spans).  The slots CLAIM their places one after the other — slot j's lanes advance the
  // group cursors by their span lengths with ONE LDS atomic that returns the old value, then slot j + 1's: the LDS keeps
  // a wave's operations in order, so the instances of a group land in the Gaussians' order whatever overlaps whatever,
  // without a rectangle test, and the eight atomics of a step are in flight together (inside a slot no two spans share a
  // group, so their order does not matter).  History: lane = tile with geometric tests of the earlier slots, 75
  // instructions per step of two Gaussians, bound by instruction issue: 208 us; span lanes with a read and a write of a
  // 16-bit cursor per slot — 22 LDS instructions per step, each waiting for the one before: 175 us, bound by the LDS.
  // A span is then written with up to four stores (4 + 4, 4 + 2 + 1, ... dwords; the places are only dword-aligned, which
  // global stores of any width accept).
  uint32_t* cur = s_cur + (size_t)wave * G;
  auto put_span = [&](const int64_t pos, const uint32_t g, const int span0, const int L) {
    const uint32_t v0 = g | ((uint32_t)(span0 & 7) << 29);  // (a span stays inside its group: no wrap of the three tile bits)
    if (pos + L <= cap) {
      uint32_t* q = inter + pos;
      int off = 0;
      if (L >= 4) { GU4 v; v.a = v0; v.b = v0 + (1u << 29); v.c = v0 + (2u << 29); v.d = v0 + (3u << 29); *reinterpret_cast<GU4*>(q) = v; off = 4; }
      if (L == 8) { GU4 v; v.a = v0 + (4u << 29); v.b = v0 + (5u << 29); v.c = v0 + (6u << 29); v.d = v0 + (7u << 29); *reinterpret_cast<GU4*>(q + 4) = v; off = 8; }
      const int rem = L - off;  // 0 .. 3
      if (rem & 2) { GU2 v; v.a = v0 + ((uint32_t)off << 29); v.b = v0 + ((uint32_t)(off + 1) << 29); *reinterpret_cast<GU2*>(q + off) = v; off += 2; }
      if (rem & 1) q[off] = v0 + ((uint32_t)off << 29);
    } else {
      for (int k = 0; k < L; k++) if (pos + k < cap) inter[pos + k] = v0 + ((uint32_t)k << 29);  // (the arena overflows: the frame is flagged)
    }
  };
  for (int s0 = first; s0 < end; s0 += 64) {
    const int s = s0 + lane;
    uint32_t my_g = 0u;
    int pk_xy = 0, pk_zw = 0, my_ns = 0;
    if (s < end) {
      my_g = order[s];
      const ushort4 rc = srect[s];
      if (rc.z > rc.x) {
        pk_xy = (int)rc.x | ((int)rc.y << 16); pk_zw = (int)rc.z | ((int)rc.w << 16);
        my_ns = ((int)rc.w - (int)rc.y) * ((((int)rc.z - 1) >> 3) - ((int)rc.x >> 3) + 1);
      }
    }
    const uint64_t live = __builtin_amdgcn_ballot_w64(my_ns != 0);
    const uint64_t over8 = __builtin_amdgcn_ballot_w64(my_ns > 8);
    const uint64_t over16 = __builtin_amdgcn_ballot_w64(my_ns > 16);
    // one step of SLOTS Gaussians (lanes base .. base + SLOTS - 1 of the batch), 64 / SLOTS lanes each
    auto step = [&](const int base, auto slots_tag) {
      constexpr int SLOTS = decltype(slots_tag)::value, LN = 64 / SLOTS;
      const int slot = lane / LN, l = lane % LN;
      const int axy = __shfl(pk_xy, base + slot), bzw = __shfl(pk_zw, base + slot);
      const uint32_t g = (uint32_t)__shfl((int)my_g, base + slot);
      const int rx0 = axy & 0xFFFF, ry0 = (int)((uint32_t)axy >> 16), rx1 = bzw & 0xFFFF, ry1 = (int)((uint32_t)bzw >> 16);
      const int gc0 = rx0 >> 3, ncol = ((rx1 - 1) >> 3) - gc0 + 1;  // (the empty rectangle 0,0,0,0 of an invisible Gaussian: ncol = 0)
      const bool valid = l < (ry1 - ry0) * ncol;
      const int row = (int)(((float)l + 0.5f) * __builtin_amdgcn_rcpf((float)ncol));  // (l / ncol, as rect_tile)
      const int col = gc0 + (l - row * ncol), y = ry0 + row;
      const int span0 = max(rx0, col << 3), L = min(rx1, (col << 3) + 8) - span0;
      const int gi = valid ? y * gxg + col : 0;
      // (a result register per slot, OR-ed afterwards: into one register the compiler waits for every atomic before the next)
      uint32_t got[SLOTS];
#pragma unroll
      for (int jj = 0; jj < SLOTS; jj++) got[jj] = 0u;
#pragma unroll
      for (int jj = 0; jj < SLOTS; jj++)
        if (valid && slot == jj) got[jj] = atomicAdd(&cur[gi], (uint32_t)L);
      uint32_t pos = 0u;
#pragma unroll
      for (int jj = 0; jj < SLOTS; jj++) pos |= got[jj];
      if (valid) put_span((int64_t)pos, g, span0, L);
    };
#pragma unroll 1
    for (int q = 0; q < 8; q++) {  // (not unrolled: the three kinds of step, eight times over, would not fit the instruction cache)
      if (((live >> (8 * q)) & 0xFFull) == 0ull) continue;
      if (((over8 >> (8 * q)) & 0xFFull) == 0ull) { step(8 * q, std::integral_constant<int, 8>{}); continue; }
#pragma unroll 1
      for (int hh = 0; hh < 2; hh++) {
        const int base = 8 * q + 4 * hh;
        if (((live >> base) & 0xFull) == 0ull) continue;
        if (((over16 >> base) & 0xFull) == 0ull) { step(base, std::integral_constant<int, 4>{}); continue; }
#pragma unroll 1
        for (int jj = 0; jj < 4; jj++) {
          // a step of its own: lanes = the Gaussian's spans, 64 at a time (every group at most once per Gaussian)
          const int src = base + jj;
          const int ns = __builtin_amdgcn_readlane(my_ns, src);
          if (ns == 0) continue;
          const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)my_g, src);
          const int r0 = __builtin_amdgcn_readlane(pk_xy, src), r1 = __builtin_amdgcn_readlane(pk_zw, src);
          const int rx0 = r0 & 0xFFFF, ry0 = (int)((uint32_t)r0 >> 16), rx1 = r1 & 0xFFFF;
          const int gc0 = rx0 >> 3, ncol = ((rx1 - 1) >> 3) - gc0 + 1;
          const float rn = __builtin_amdgcn_rcpf((float)ncol);
          for (int l = lane; l < ns; l += 64) {
            const int row = (int)(((float)l + 0.5f) * rn);
            const int col = gc0 + (l - row * ncol), y = ry0 + row;
            const int span0 = max(rx0, col << 3), L = min(rx1, (col << 3) + 8) - span0;
            const int gi = y * gxg + col;
            put_span((int64_t)atomicAdd(&cur[gi], (uint32_t)L), g, span0, L);
          }
        }
      }
    }
  }
}

// The parts of the second level: workgroup b of its kernels takes part_tab[b] = {group, first part of the group, start in
// the scratch list, length}; group = 0xFFFFFFFF beyond the last part.  Written by the EXTRA (last) workgroup of
// gbin_scatter_kernel from the group counts (every workgroup scanning the <= 8192 counts itself cost the second level's
// kernels 3.5 us each, 5 700 times over).
struct GPart { int group; uint32_t first_part; uint32_t start; uint32_t len; uint32_t part; };
__device__ void gbin_parts_body(int G, int64_t cap, uint32_t n_parts, const uint32_t* __restrict__ group_count, uint4* __restrict__ part_tab) {
  __shared__ uint32_t s_wc[16], s_wp[16], s_carry_c, s_carry_p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  if (tid == 0) { s_carry_c = 0u; s_carry_p = 0u; }
  __syncthreads();
  for (int base = 0; base < G; base += nthr) {
    const int g = base + tid;
    const uint32_t c = (g < G) ? group_count[g] : 0u;
    const uint32_t np = (c + BIN_PART - 1) / BIN_PART;
    uint32_t vc = c, vp = np;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t uc = (uint32_t)__shfl_up((int)vc, o), up = (uint32_t)__shfl_up((int)vp, o);
      if (lane >= o) { vc += uc; vp += up; }
    }
    if (lane == 63) { s_wc[wave] = vc; s_wp[wave] = vp; }
    __syncthreads();
    uint32_t oc = s_carry_c, op = s_carry_p;
    for (int w = 0; w < wave; w++) { oc += s_wc[w]; op += s_wp[w]; }
    const uint32_t cstart = oc + vc - c, pstart = op + vp - np;
    for (uint32_t part = 0; part < np; part++) {
      const int64_t lo = (int64_t)cstart + (int64_t)part * BIN_PART;
      const int64_t hi = min((int64_t)cstart + min((int64_t)c, (int64_t)(part + 1) * BIN_PART), cap);
      if (pstart + part < n_parts) part_tab[pstart + part] = make_uint4((uint32_t)g, pstart, (uint32_t)min(lo, cap), hi > lo ? (uint32_t)(hi - lo) : 0u);
    }
    __syncthreads();
    if (tid == nthr - 1) { s_carry_c = oc + vc; s_carry_p = op + vp; }
    __syncthreads();
  }
  for (uint32_t b = s_carry_p + (uint32_t)tid; b < n_parts; b += (uint32_t)nthr) part_tab[b] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
}
__device__ __forceinline__ GPart gbin_part(int b, const uint4* __restrict__ part_tab) {
  const uint4 t = part_tab[b];
  GPart p;
  p.group = (int)t.x; p.first_part = t.y; p.start = t.z; p.len = t.w; p.part = (uint32_t)b - t.y;
  return p;
}

// (2a) per part: instances per tile of the group -> part_hist[part][8]; totals into tile_count
__global__ __launch_bounds__(512) void gbin_tcount_kernel(int gxg, int grid_x, const uint4* __restrict__ part_tab,
                                                         const uint32_t* __restrict__ inter, uint32_t* __restrict__ part_hist,
                                                         uint32_t* __restrict__ tile_count) {
  __shared__ uint32_t s_h[8];
  const GPart p = gbin_part((int)blockIdx.x, part_tab);
  if (p.group < 0) return;
  if (threadIdx.x < 8) s_h[threadIdx.x] = 0u;
  __syncthreads();
  // eight 8-bit counters in one 64-bit word (a thread sees at most BIN_PART / 512 = 16 instances)
  uint64_t packed = 0ull;
#pragma unroll
  for (int it = 0; it < BIN_PART / 512; it++) {
    const uint32_t k = (uint32_t)it * 512u + threadIdx.x;
    if (k < p.len) packed += 1ull << ((inter[p.start + k] >> 29) * 8u);
  }
#pragma unroll
  for (int q = 0; q < 8; q++) {
    uint32_t v = (uint32_t)(packed >> (8 * q)) & 0xFFu;
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_h[q], v);
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const uint32_t v = s_h[threadIdx.x];
    part_hist[(size_t)blockIdx.x * 8 + threadIdx.x] = v;
    const int x = (p.group % gxg) * 8 + (int)threadIdx.x, y = p.group / gxg;
    if (v && x < grid_x) atomicAdd(&tile_count[y * grid_x + x], v);
  }
}

// (2b) per part: stable partition by tile into the final lists; the extra (last) workgroup writes ranges, counters, work lists.
// The second level is VALU work, not memory work (38.8 M instances x every instruction spent per instance; the version that
// ranked 64 instances at a time with ballots took 85 wave instructions per 64: 239 us), so a thread takes SIXTEEN
// CONSECUTIVE instances: counts them in packed counters, one packed scan over the threads, places them from packed running
// offsets — and the part goes through LDS twice: in (coalesced loads -> a thread's 16 in a row) and out (final order ->
// long coalesced runs).
__global__ __launch_bounds__(GTS_THREADS, GTS_WAVES) void gbin_tscatter_kernel(int T, int G, int gxg, int grid_x, int64_t cap,
                                                              const uint32_t* __restrict__ group_count, const uint4* __restrict__ part_tab,
                                                              const uint32_t* __restrict__ inter,
                                                              const uint32_t* __restrict__ part_hist, const uint32_t* __restrict__ tile_count,
                                                              uint32_t* __restrict__ point_list, uint32_t* __restrict__ tile_keys,
                                                              const BinOut out) {
  if (blockIdx.x == 0) {  // (the extra workgroup FIRST: it runs for ~20 us, and the last of 8 000 workgroups would start it at the very end)
    bin_offsets_body(T, cap, tile_count, out, group_count, G);
    return;
  }
  constexpr int EPT = BIN_PART / GTS_THREADS, NW = GTS_THREADS / 64;  // instances per thread, waves
  static_assert(EPT * 64 < 65536 && BIN_PART < 65536, "packed 16-bit counters below");
  __shared__ uint32_t s_cur[8];            // start of the group's tiles' lists for this part (global)
  __shared__ uint32_t s_wtot[NW][4];        // [wave] packed pairs of 16-bit counts: tiles (0,1) (2,3) (4,5) (6,7)
  __shared__ uint32_t s_wbase[NW][4];       // [wave] packed pairs: place (in the part's final order) of the wave's first instance of each tile
  __shared__ uint32_t s_lstart[9];         // start of each tile in the part's final order
  __shared__ uint32_t s_stage[BIN_PART + BIN_PART / EPT];
  const GPart p = gbin_part((int)blockIdx.x - 1, part_tab);
  if (p.group < 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gx0 = (p.group % gxg) * 8, gy = p.group / gxg;
  if (wave == 0) {
    // start of tile q's list = start of the group's segment + the group's earlier tiles + tile q's share of the earlier parts
    // (lane = (stride s, tile q): the earlier parts s, s + 8, ... — a dense group has over a hundred parts)
    const int q = lane & 7, sl = lane >> 3;
    uint32_t share = 0u;
    for (uint32_t k = (uint32_t)sl; k < p.part; k += 8u) share += part_hist[(size_t)(p.first_part + k) * 8 + q];
    share += (uint32_t)__shfl_xor((int)share, 8);
    share += (uint32_t)__shfl_xor((int)share, 16);
    share += (uint32_t)__shfl_xor((int)share, 32);
    const uint32_t tc = (gx0 + q < grid_x) ? tile_count[gy * grid_x + gx0 + q] : 0u;
    uint32_t v = tc;
    for (int o = 1; o < 8; o <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)v, o);
      if (q >= o) v += u;
    }
    if (lane < 8) s_cur[q] = (p.start - p.part * BIN_PART) + (v - tc) + share;  // (p.start - ...: the group's segment; the clamp to `cap` only bites on overflow)
  }
  // in: coalesced loads, padded rows of 16 in LDS, a thread's row into registers
#pragma unroll
  for (int it = 0; it < EPT; it++) {
    const uint32_t k = (uint32_t)it * GTS_THREADS + (uint32_t)tid;
    s_stage[k + k / EPT] = (k < p.len) ? inter[p.start + k] : 0u;
  }
  __syncthreads();
  uint32_t ev[EPT];
#pragma unroll
  for (int i = 0; i < EPT; i++) ev[i] = s_stage[tid * (EPT + 1) + i];
  const uint32_t k_first = (uint32_t)tid * EPT;
  // packed counts: eight 8-bit counters in a 64-bit word while counting (a thread has 16 instances), then word w = tiles
  // (2w, 2w+1) in 16-bit halves for the scan over the threads
  uint64_t cnt8 = 0ull;
#pragma unroll
  for (int i = 0; i < EPT; i++) cnt8 += (k_first + i < p.len) ? (1ull << ((ev[i] >> 29) * 8u)) : 0ull;
  uint32_t cnt[4];
#pragma unroll
  for (int w = 0; w < 4; w++) cnt[w] = ((uint32_t)(cnt8 >> (16 * w)) & 0xFFu) | (((uint32_t)(cnt8 >> (16 * w + 8)) & 0xFFu) << 16);
  uint32_t inc_scan[4];
#pragma unroll
  for (int w = 0; w < 4; w++) {
    uint32_t v = cnt[w];
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)v, o);
      if (lane >= o) v += u;
    }
    inc_scan[w] = v;
    if (lane == 63) s_wtot[wave][w] = v;
  }
  __syncthreads();  // (also: every thread has its row out of s_stage)
  if (tid < 8) {
    // tile `tid`: its total, then (after the exchange below) the waves' bases
    uint32_t tot = 0u;
    for (int w = 0; w < NW; w++) tot += (s_wtot[w][tid >> 1] >> ((tid & 1) * 16)) & 0xFFFFu;
    uint32_t v = tot;
    for (int o = 1; o < 8; o <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)v, o);
      if (tid >= o) v += u;
    }
    s_lstart[tid + 1] = v;
    if (tid == 0) s_lstart[0] = 0u;
    uint32_t run = v - tot;  // (the tile's start in the part's final order: < 8192, and so is every place below: 16 bits hold it)
    for (int w = 0; w < NW; w++) {
      const uint32_t c = (s_wtot[w][tid >> 1] >> ((tid & 1) * 16)) & 0xFFFFu;
      // the two tiles of a word are written by two lanes: 16-bit halves through a short pointer
      reinterpret_cast<unsigned short*>(&s_wbase[w][0])[tid] = (unsigned short)run;
      run += c;
    }
  }
  __syncthreads();
  // running places: the wave's base + the earlier threads of the wave; tiles 0-3 and 4-7 in two 64-bit words of four 16-bit
  // fields (written so that nothing per instance is shared with the counting loop above: the compiler kept 64 registers of
  // common subexpressions alive between the two, i.e. half the occupancy)
  uint64_t place_lo, place_hi;
  {
    uint32_t pw[4];
#pragma unroll
    for (int w = 0; w < 4; w++) pw[w] = s_wbase[wave][w] + inc_scan[w] - cnt[w];
    place_lo = (uint64_t)pw[0] | ((uint64_t)pw[1] << 32);
    place_hi = (uint64_t)pw[2] | ((uint64_t)pw[3] << 32);
  }
#pragma unroll
  for (int i = 0; i < EPT; i++) {
    const uint32_t c = ev[i] >> 29;
    const bool hi = c >= 4u;
    const uint32_t sh = (c & 3u) * 16u;
    const uint32_t at = (uint32_t)((hi ? place_hi : place_lo) >> sh) & 0xFFFFu;
    const bool on = k_first + i < p.len;
    const uint64_t inc = on ? (1ull << sh) : 0ull;
    place_lo += hi ? 0ull : inc;
    place_hi += hi ? inc : 0ull;
    if (on) s_stage[at] = ev[i];
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  // out: the part in its final order, tile after tile: long runs
  uint32_t lstart[9];
#pragma unroll
  for (int q = 0; q < 9; q++) lstart[q] = s_lstart[q];
#pragma unroll 4
  for (uint32_t k = (uint32_t)tid; k < p.len; k += GTS_THREADS) {
    int c = 0;
#pragma unroll
    for (int q = 1; q < 8; q++) c += (k >= lstart[q]) ? 1 : 0;
    uint32_t ls = 0u;
#pragma unroll
    for (int q = 1; q < 8; q++) ls = (c == q) ? lstart[q] : ls;
    const int64_t pos = (int64_t)s_cur[c] + (k - ls);
    if (pos < cap) {
      point_list[pos] = s_stage[k] & 0x1FFFFFFFu;
      if (tile_keys) tile_keys[pos] = (uint32_t)(gy * grid_x + gx0 + c);
    }
  }
}

struct BinPlan { int g_per_block, g_per_wave, threads, n_chunks; size_t lds_scatter; };

static BinPlan bin_plan(int N, int T) {
  // LDS of the scatter kernel: T*4 (bases) + W*Tpad*2 (wave cursors) <= ~150 KB
  const int Tpad = (T + 1) & ~1;
  // 12 waves x 64 Gaussians: the ordered walk is a latency chain per wave, so short walks on many waves win;
  // beyond 12 the LDS footprint leaves one workgroup per CU
  int W = 12;
  while (W > 1 && (size_t)T * 4 + (size_t)W * Tpad * 2 > 150 * 1024) W = (W + 1) >> 1;
  BinPlan p;
  p.threads = W * 64;
  // batches of 64 Gaussians per wave: as many as keep the launch at <= ~640 chunks (one in the headline configuration:
  // 391 chunks).  A chunk costs every kernel O(T) (table row, cursor tables) and its runs in a tile's list are short — a
  // partial line per (chunk, tile) — which at 2 M Gaussians / 8160 tiles was a 170 MB table and 1.3 ms of binning with
  // 384-Gaussian chunks; measured optimum: C3 1 batch, C4 (500 k) 2-4, C5 (2 M) 8
  int batches = (int)(((int64_t)N + (int64_t)W * 64 * 640 - 1) / ((int64_t)W * 64 * 640));
  batches = batches < 1 ? 1 : (batches > 16 ? 16 : batches);
  p.g_per_wave = 64 * batches;
  p.g_per_block = W * p.g_per_wave;
  p.n_chunks = (N + p.g_per_block - 1) / p.g_per_block;
  p.lds_scatter = (size_t)T * 4 + (size_t)W * Tpad * 2;
  return p;
}

// which sort?  Beyond BIN_GROUPED_MIN_T tiles only the grouped one fits; below, it wins on many Gaussians over many tiles
// (2 M / 8160 tiles / 38.8 M instances: 0.40 ms against 0.70 ms; 500 k / 4096 tiles / 4.5 M instances: the frame 0.531
// against 0.551 ms) and loses on the small scenes (300 k / 2500 tiles: +0.04 ms per frame, 150 k / 2500: +0.02 — its five
// launches cost more than they save).  riggs_set_option("bin_grouped", 0 / 1) overrides the choice where both fit
// (measurements, tests; -1 = by size, the default).
static bool bin_grouped(int N, int T) {
  if (T > BIN_GROUPED_MIN_T) return true;
  const int forced = option(OPT_BIN_GROUPED);
  if (forced >= 0) return forced != 0 && T >= 64;
  return T >= BIN_GROUPED_AUTO_T && N >= BIN_GROUPED_AUTO_N;
}
// (the direct sort's scatter launch hosts the colour job; up to 16 coefficients per channel — degree 3.  The grouped sort's
// first-level scatter does not: where that sort runs — half a million Gaussians and more — its launches have no idle slots, and the
// colour blocks in its 1024-thread workgroups cost more than they save: C4 0.511 -> 0.518 ms, C5 1.423 -> 1.497 ms)
bool binning_hosts_color(int N, int T, int sh_coeffs) {
  return option(OPT_COLOR_SIDE_JOBS) != 0 && N > 0 && sh_coeffs >= 1 && sh_coeffs <= 16 && T <= 65535 && !bin_grouped(N, T);
}
// the colour job rides on a scatter launch of `threads` threads and `lds` bytes per workgroup: blocks of gpb Gaussians — as many
// as the workgroup has threads and its LDS holds rows for (at least 64; the LDS grows to 256 rows if it is smaller)
static void color_job_plan(const ColorJob* job_rec, int N, int T, int sh_coeffs, int threads, int& n_col, int& gpb, size_t& lds) {
  n_col = 0; gpb = 0;
  if (!job_rec || !binning_hosts_color(N, T, sh_coeffs)) return;
  const size_t row = (size_t)((sh_coeffs * 3) | 1) * 4;  // (the longest row: one (N, M, 3) tensor)
  gpb = threads < 384 ? threads : 384;
  while (gpb > 64 && gpb * row > (lds > 256 * row ? lds : 256 * row)) gpb -= 64;
  if (gpb * row > lds) lds = gpb * row;
  n_col = (N + gpb - 1) / gpb;
}
struct GBinPlan { int G, gxg, W, g_per_block, g_per_wave, n_chunks; size_t lds; };
static GBinPlan gbin_plan(int N, int T, int grid_x) {
  GBinPlan p;
  p.gxg = (grid_x + 7) >> 3;
  const int grid_y = grid_x > 0 ? (T + grid_x - 1) / grid_x : 0;
  p.G = p.gxg * grid_y;
  // waves per workgroup: sixteen, fewer where their 32-bit cursors (G x 4 bytes per wave, next to the G x 4 of the chunk's
  // starts) would not fit the LDS (3840 x 2160: 4050 groups, 8 waves)
  int W = 16;
  while (W > 1 && (size_t)p.G * 4 + (size_t)W * p.G * 4 > 150 * 1024) W >>= 1;
  p.W = W;
  // batches of 64 Gaussians per wave so that a launch has <= ~512 chunks (one round of the scatter kernel's workgroups; the
  // chunk x group table and every workgroup's O(G) set-up stay small; bin_scan keeps a column segment in registers up to
  // 16 x 32 chunks) — and so few that a chunk's run in a group stays below 65 536 instances: a Gaussian adds up to 8
  // instances to a group, and the waves' starts inside a chunk's run are handed over as 16-bit (16 waves: 7 batches,
  // 16 x 7 x 64 x 8 = 57 344)
  int batches = (int)(((int64_t)N + (int64_t)W * 64 * 512 - 1) / ((int64_t)W * 64 * 512));
  const int max_batches = 65535 / (W * 64 * 8);
  batches = batches < 1 ? 1 : (batches > max_batches ? max_batches : batches);
  p.g_per_wave = 64 * batches;
  p.g_per_block = W * p.g_per_wave;
  p.n_chunks = (N + p.g_per_block - 1) / p.g_per_block;
  p.lds = (size_t)p.G * 4 + (size_t)W * p.G * 4;
  return p;
}

// (the larger of the two sorts' tables wherever both sorts fit: which one runs — by size, or forced with
// riggs_set_option("bin_grouped") — then never changes what an arena must hold, and pruning a scene across the switch
// (499 999 Gaussians instead of 500 000 over 25 600 tiles) does not ask for a LARGER arena than before)
size_t bin_table_bytes(int N, int T, int grid_x) {
  size_t grouped = 0, direct = 0;
  if (T >= 64) {
    GBinPlan p = gbin_plan(N > 0 ? N : 1, T, grid_x);
    grouped = align_up(((size_t)p.n_chunks + 1) * p.G * 4) + align_up((size_t)(p.G + 1) * 4) + align_up((size_t)(T + 1) * 4) +
              align_up((size_t)p.n_chunks * p.W * ((p.G + 1) & ~1) * 2) + align_up((size_t)(N > 0 ? N : 1) * 8);
  }
  if (T <= BIN_GROUPED_MIN_T) {
    BinPlan p = bin_plan(N > 0 ? N : 1, T);
    direct = align_up(((size_t)p.n_chunks + 1) * T * 4) + align_up((size_t)(T + 1) * 4) + align_up((size_t)(N > 0 ? N : 1) * 8);
  }
  return grouped > direct ? grouped : direct;
}
size_t bin_scratch_bytes(int64_t cap, int T, int grid_x) {  // (grouped binning: lives in the checkpoint area)
  GBinPlan p = gbin_plan(1, T, grid_x);
  const size_t c = (size_t)(cap > 0 ? cap : 1);
  return align_up(c * 4) + align_up((c / BIN_PART + (size_t)p.G + 2) * 8 * 4) + align_up((c / BIN_PART + (size_t)p.G + 2) * 16);
}

int launch_binning(int N, int T, int grid_x, int64_t cap, const uint32_t* order, const uint32_t* tiles,
                   const ushort4* rect, void* table_mem, void* scratch, uint32_t* point_list, uint32_t* tile_keys, const BinOut& out,
                   hipStream_t s, const ColorJob* job_rec, int job_sh_coeffs) {
  (void)tiles;  // (the rectangles alone say which Gaussians are visible: preprocess_fwd leaves a culled one's empty)
  if (T > 65535) { set_error("image too large: %d tiles (at most 65 535)", T); return 2; }
  char* mem = (char*)table_mem;
  static unsigned long long attr_done = 0ull;  // (per device: once_per_device)
  if (once_per_device(attr_done)) {
    // (dynamic + static LDS <= 160 KB: the kernel also has ~1 KB of static scratch for its scans)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bin_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bin_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gbin_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gbin_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
  }
  if (bin_grouped(N, T)) {
    const GBinPlan p = gbin_plan(N, T, grid_x);
    if (p.lds > 150 * 1024) { set_error("image too large: %d tile groups", p.G); return 2; }
    uint32_t* table = (uint32_t*)mem;
    uint32_t* group_count = (uint32_t*)(mem + align_up(((size_t)p.n_chunks + 1) * p.G * 4));
    uint32_t* tile_count = (uint32_t*)((char*)group_count + align_up((size_t)(p.G + 1) * 4));
    uint32_t* wave_start = (uint32_t*)((char*)tile_count + align_up((size_t)(T + 1) * 4));
    ushort4* srect = (ushort4*)((char*)wave_start + align_up((size_t)p.n_chunks * p.W * ((p.G + 1) & ~1) * 2));
    uint32_t* inter = (uint32_t*)scratch;
    uint32_t* part_hist = (uint32_t*)((char*)scratch + align_up((size_t)(cap > 0 ? cap : 1) * 4));
    const unsigned n_parts = (unsigned)((cap > 0 ? cap : 1) / BIN_PART + p.G + 1);
    uint4* part_tab = (uint4*)((char*)part_hist + align_up(((size_t)(cap > 0 ? cap : 1) / BIN_PART + (size_t)p.G + 2) * 8 * 4));
    hipLaunchKernelGGL(gbin_count_kernel, dim3(p.n_chunks), dim3(p.W * 64), (size_t)p.W * ((p.G + 1) & ~1) * 2, s, N, T, p.G, p.gxg,
                       p.g_per_block, p.g_per_wave, order, rect, srect, table, wave_start, tile_count);
    hipLaunchKernelGGL(bin_scan_kernel, dim3((p.G + 63) / 64), dim3(1024), 0, s, p.G, p.n_chunks, table, group_count);
    // (+ 1: the extra workgroup that lists the second level's parts)
    hipLaunchKernelGGL(gbin_scatter_kernel, dim3(p.n_chunks + 1), dim3(p.W * 64), p.lds, s, N, p.G, p.gxg, cap, p.g_per_block, p.g_per_wave,
                       order, srect, table, wave_start, group_count, inter, part_tab, n_parts);
    hipLaunchKernelGGL(gbin_tcount_kernel, dim3(n_parts), dim3(512), 0, s, p.gxg, grid_x, part_tab, inter, part_hist, tile_count);
    // (+ 1: the extra workgroup that writes the ranges, the counters and the forward's work list)
    hipLaunchKernelGGL(gbin_tscatter_kernel, dim3(n_parts + 1), dim3(GTS_THREADS), 0, s, T, p.G, p.gxg, grid_x, cap, group_count, part_tab,
                       inter, part_hist, tile_count, point_list, tile_keys, out);
    return 0;
  }
  BinPlan p = bin_plan(N, T);
  uint32_t* table = (uint32_t*)mem;
  uint32_t* tile_count = (uint32_t*)(mem + align_up(((size_t)p.n_chunks + 1) * T * 4));
  ushort4* srect = (ushort4*)((char*)tile_count + align_up((size_t)(T + 1) * 4));
  if (p.lds_scatter > 150 * 1024) {  // (cannot happen: large tile grids take the grouped path)
    set_error("image too large: %d tiles", T);
    return 2;
  }
  hipLaunchKernelGGL(bin_count_kernel, dim3(p.n_chunks), dim3(p.threads), (size_t)T * 4, s, N, T, grid_x, p.g_per_block,
                     order, rect, srect, table);
  hipLaunchKernelGGL(bin_scan_kernel, dim3((T + 63) / 64), dim3(1024), 0, s, T, p.n_chunks, table, tile_count);
  // (+ 1: the extra workgroup that writes the ranges, the counters and the forward's work list)
  int n_col = 0, gpb = 0;
  size_t lds = p.lds_scatter;
  color_job_plan(job_rec, N, T, job_sh_coeffs, p.threads, n_col, gpb, lds);
  hipLaunchKernelGGL(bin_scatter_kernel, dim3(p.n_chunks + 1 + n_col), dim3(p.threads), lds, s, N, T, grid_x, cap,
                     p.g_per_block, p.g_per_wave, order, srect, table, tile_count, point_list, tile_keys, out, p.n_chunks, job_rec, gpb);
  return 0;
}


// =====================================================================================================
// Depth sort of the Gaussians: stable LSD radix sort of (u32 depth bits, u32 index) pairs on 12-bit digits —
// bits [0, 12), [12, 24) and, ONLY IF NEEDED, [24, 32) — each pass a counting sort built like the tile binning above:
//   rs_count   : per chunk of 2048 elements, a 4096-bin histogram in LDS             -> table[chunk][bin]
//   bin_scan   : per bin, exclusive scan over the chunks (same kernel as the binning)  -> bin_count[bin]
//   rs_scatter : per chunk: scan of bin_count in LDS, per-wave offsets, then each wave ranks its elements
//                64 at a time in input order (equal digits matched with 12 ballots) and scatters.
// The top byte of a positive float is its sign and the upper seven exponent bits: it is the same for every depth of a
// scene that lies inside one of the ranges [2, 8), [0.5, 2), [8, 32), ... (a D-NeRF / ZJU camera looks at its subject
// from 2 - 6 units).  preprocess_fwd records the top bytes it saw; the first kernel folds them into counters[2] =
// "third pass needed", and the third pass — inside the second pass's scatter launch, behind in-launch barriers — is a load
// of that flag when it is not: two passes (six short launches) instead of three 11-bit passes (nine).  Between the passes the elements
// travel as (key, value) PAIRS — one 8-byte scattered store per element instead of two of 4: the scatter kernels are bound by
// the number of single-dword line writes — and the last pass writes the values alone (no kernel reads sorted keys).
// rocPRIM picks a block sort + ~9 merge passes (18 launches, 0.12 ms) at N = 3e5 and Onesweep's chained
// look-back costs the same at this size.
// =====================================================================================================
#define RS_BITS 12
#define RS_BINS (1 << RS_BITS)
#define RS_CHUNK 2048   // elements per workgroup
#define RS_CHUNK_BIG 8192  // ... from RS_BIG_MIN_N keys on
#ifndef RS_BIG_MIN_N
#define RS_BIG_MIN_N 1000000
#endif
#define RS_SC_WAVES 8   // waves of a scatter workgroup (4 steps of 64 elements each)

// routing of a pass: (source, destination) by the "third pass needed" flag (counters[2])
struct RsBufs {
  const uint32_t* k_in;   // the caller's keys (never written): pass 0 reads them, the value is the index
  uint2 *a, *b;           // two scratch lists of (key, value) PAIRS: one 8-byte scattered store per element instead of two of 4
  uint32_t* v_out;        // the result: the values in key order (the LAST pass writes only them: nobody reads sorted keys)
};
// two passes:   in -> a -> out            three passes:   in -> b -> a -> out
__device__ __forceinline__ void rs_route(const RsBufs& r, int pass, bool three, const uint2*& src, uint2*& dst) {
  if (pass == 0) { src = nullptr; dst = three ? r.b : r.a; }
  else if (pass == 1) { src = three ? r.b : r.a; dst = three ? r.a : nullptr; }
  else { src = r.a; dst = nullptr; }
}

// (the last workgroup of the first pass also totals the per-workgroup tile counts of preprocess_fwd into
// counters[0] = R for the host, and every workgroup of it derives the flag from the recorded top bytes — the last one
// publishes it in counters[2] for the later launches)
template <int CHUNK>
__global__ __launch_bounds__(CHUNK / 8) void rs_count_kernel(int N, int pass, RsBufs bufs, uint32_t* __restrict__ table,
                                                             const uint32_t* __restrict__ part, int n_part,
                                                             uint32_t* __restrict__ counters, uint32_t* __restrict__ arrivals) {
  constexpr int NT = CHUNK / 8, NW = NT / 64;  // (eight elements per thread)
  __shared__ uint32_t s_hist[RS_BINS];
  __shared__ uint32_t s_w[NW], s_lo[NW], s_hi[NW];
  // (pass 0: this thread's keys are asked for before the reduction below, not behind its round trip and barrier; the later passes
  // read their source list where the flag they route by is known)
  const int first = blockIdx.x * CHUNK;
  uint32_t k0[CHUNK / NT];
#pragma unroll
  for (int k = 0; k < CHUNK / NT; k++) {
    const int i = first + k * NT + threadIdx.x;
    k0[k] = (pass == 0 && i < N) ? bufs.k_in[i] : 0u;
  }
  bool three;
  if (pass == 0) {
    // part[0 .. n_part): tile counts; part[n_part .. 2 n_part): (min top byte << 8) | max top byte of the visible keys
    uint32_t v = 0, lo = 0xFFu, hi = 0u;
    for (int i = threadIdx.x; i < n_part; i += NT) {
      v += part[i];
      const uint32_t tb = part[n_part + i];
      lo = min(lo, tb >> 8); hi = max(hi, tb & 0xFFu);
    }
    for (int o = 32; o > 0; o >>= 1) {
      v += (uint32_t)__shfl_xor((int)v, o);
      lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
    }
    if ((threadIdx.x & 63) == 0) { s_w[threadIdx.x >> 6] = v; s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    uint32_t total = 0;
    lo = 0xFFu; hi = 0u;
#pragma unroll
    for (int w = 0; w < NW; w++) { total += s_w[w]; lo = min(lo, s_lo[w]); hi = max(hi, s_hi[w]); }
    three = hi > lo;  // (no visible key at all: lo = 255 > hi = 0: two passes)
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
      counters[0] = total; counters[1] = 0u; counters[2] = three ? 1u : 0u; counters[3] = 0u;
      arrivals[0] = 0u;
    }
  } else {
    three = counters[2] != 0u;
  }
  const uint2* src; uint2* dst;
  rs_route(bufs, pass, three, src, dst);
  const int shift = pass * RS_BITS;
  for (int b = threadIdx.x; b < RS_BINS; b += NT) s_hist[b] = 0u;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < CHUNK / NT; k++) {
    const int i = first + k * NT + threadIdx.x;
    if (i < N) atomicAdd(&s_hist[((src ? src[i].x : k0[k]) >> shift) & (RS_BINS - 1)], 1u);
  }
  __syncthreads();
  uint32_t* row = table + (size_t)blockIdx.x * RS_BINS;
  for (int b = threadIdx.x; b < RS_BINS; b += NT) row[b] = s_hist[b];
}

// (the third pass's scan when that pass is skipped: bin_scan_kernel itself is shared with the tile binning, so a thin
// wrapper checks the flag)
__global__ __launch_bounds__(1024) void rs_scan_kernel(int n_chunks, uint32_t* __restrict__ table, uint32_t* __restrict__ bin_count,
                                                       const uint32_t* __restrict__ counters, int pass) {
  (void)counters; (void)pass;
  bin_scan_body(RS_BINS, n_chunks, table, bin_count);
}

// The THIRD pass (top byte: 256 bins) INSIDE the second pass's scatter launch: it runs only when the depths straddle a
// power-of-four boundary; as three launches that leave at once it cost 13.7 us of every frame, as one 4.8 us (a graph node's
// floor), here a load of the flag.  When it does run: every workgroup of the scatter launch publishes its part of the second
// pass (release fence: the other XCDs' L2s do not see plain stores before a kernel boundary) and arrives at a counter; the
// first G of them — G = half of what is resident at once, so that the others always find a CU — wait for all arrivals
// (bounded spin), invalidate their caches and run the pass: they histogram their chunks, publish the 256 counts with
// write-through stores, meet at a second in-launch barrier (counters[3]; both counters cleared by the first kernel of the
// sort) and then each reads the whole table — every other chunk's counts — to place its own elements.
#define RS3_BINS 256
#define RS3_MAX_WG 512  // upper bound of the third pass's workgroups (launch_depth_sort takes half of what is resident; each loops over its chunks)
typedef __attribute__((address_space(1))) uint32_t rs_gu32;
template <int CHUNK>
__device__ void rs_third_pass_body(int N, int n_chunks, const RsBufs& bufs, uint32_t* __restrict__ table3,
                                   uint32_t* __restrict__ counters, const int G /* participating workgroups: blockIdx.x < G */) {
  __shared__ uint32_t s_hist[RS3_BINS];
  __shared__ uint32_t s_start[RS3_BINS];
  __shared__ unsigned short s_wave[RS_SC_WAVES][RS3_BINS];
  __shared__ int s_fail;
  const uint2* src = bufs.a;
  uint32_t* vd = bufs.v_out;
  constexpr int NT = RS_SC_WAVES * 64, STEPS = CHUNK / NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- phase A: the counts of this workgroup's chunks, published with write-through stores
  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += G) {
    if (tid < RS3_BINS) s_hist[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int st = 0; st < STEPS; st++) {
      const int i = chunk * CHUNK + wave * (64 * STEPS) + st * 64 + lane;
      if (i < N) atomicAdd(&s_hist[src[i].x >> 24], 1u);
    }
    __syncthreads();
    if (tid < RS3_BINS)
      __hip_atomic_store((rs_gu32*)(table3 + (size_t)chunk * RS3_BINS + tid), s_hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) s_fail = 0;
  __syncthreads();
  // ---- the in-launch barrier
  if (tid == 0) {
    __hip_atomic_fetch_add((rs_gu32*)(counters + 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    while (__hip_atomic_load((rs_gu32*)(counters + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)G) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 22)) { s_fail = 1; break; }  // (a workgroup that never became resident; not on an idle chip)
    }
  }
  __syncthreads();
  if (s_fail) {  // flag the frame (bit 1 of counters[1]) instead of hanging: its ordering is undefined
    if (tid == 0) atomicOr(counters + 1, 2u);
    return;
  }
  // ---- phase B: per chunk, the start of every bin (total over all chunks, part of the earlier ones), then rank and scatter
  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += G) {
    for (int e = tid; e < RS_SC_WAVES * RS3_BINS / 2; e += NT) reinterpret_cast<uint32_t*>(&s_wave[0][0])[e] = 0u;
    if (tid < RS3_BINS) {
      uint32_t total = 0, before = 0;
      for (int c = 0; c < n_chunks; c += 4) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
          v[u] = (c + u < n_chunks) ? __hip_atomic_load((rs_gu32*)(table3 + (size_t)(c + u) * RS3_BINS + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) { total += v[u]; if (c + u < chunk) before += v[u]; }
      }
      s_hist[tid] = total;
      s_start[tid] = before;
    }
    __syncthreads();
    uint32_t key[STEPS], val[STEPS];
    int dig[STEPS];
#pragma unroll
    for (int st = 0; st < STEPS; st++) {
      const int i = chunk * CHUNK + wave * (64 * STEPS) + st * 64 + lane;
      key[st] = 0u; val[st] = 0u; dig[st] = -1;
      if (i < N) {
        const uint2 kv = src[i];
        key[st] = kv.x; val[st] = kv.y;
        dig[st] = (int)(key[st] >> 24);
        atomicAdd(reinterpret_cast<uint32_t*>(&s_wave[wave][0]) + (dig[st] >> 1), 1u << (16 * (dig[st] & 1)));
      }
    }
    if (wave == 0) {  // exclusive scan of the 256 totals (4 per lane) -> start of every bin
      uint32_t c4[4], t = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) { c4[u] = s_hist[lane * 4 + u]; t += c4[u]; }
      uint32_t v = t;
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u2 = (uint32_t)__shfl_up((int)v, o);
        if (lane >= o) v += u2;
      }
      uint32_t run = v - t;
#pragma unroll
      for (int u = 0; u < 4; u++) { s_start[lane * 4 + u] += run; run += c4[u]; }
    }
    __syncthreads();
    for (int b = tid; b < RS3_BINS; b += NT) {  // counts -> offsets of the waves inside the chunk's segment
      unsigned short r = 0;
#pragma unroll
      for (int w = 0; w < RS_SC_WAVES; w++) { const unsigned short cw = s_wave[w][b]; s_wave[w][b] = r; r += cw; }
    }
    __syncthreads();
    unsigned short* cur = s_wave[wave];
#pragma unroll
    for (int st = 0; st < STEPS; st++) {
      const bool on = dig[st] >= 0;
      uint64_t same = __builtin_amdgcn_ballot_w64(on);
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const uint64_t vote = __builtin_amdgcn_ballot_w64(on && ((dig[st] >> b) & 1));
        same &= ((dig[st] >> b) & 1) ? vote : ~vote;
      }
      if (on) {
        const uint64_t below = same & ((1ull << lane) - 1ull);
        const uint32_t rank = (uint32_t)__builtin_popcountll(below);
        const unsigned short base = cur[dig[st]];
        const uint32_t pos = s_start[dig[st]] + base + rank;
        vd[pos] = val[st];
        if ((same >> lane) >> 1 == 0ull) cur[dig[st]] = base + (unsigned short)__builtin_popcountll(same);
      }
    }
    __syncthreads();
  }
}

template <int CHUNK>
__global__ __launch_bounds__(RS_SC_WAVES * 64) void rs_scatter_kernel(int N, int pass, RsBufs bufs,
                                                                      const uint32_t* __restrict__ table /* exclusive over chunks */,
                                                                      const uint32_t* __restrict__ bin_count,
                                                                      uint32_t* __restrict__ counters,
                                                                      uint32_t* __restrict__ table3 /* = table, for the third pass */,
                                                                      uint32_t* __restrict__ arrivals, int g3) {
  __shared__ uint32_t s_start[RS_BINS];                      // global start of this chunk's segment in each bin
  __shared__ unsigned short s_wave[RS_SC_WAVES][RS_BINS];    // per-wave counts -> per-wave running offsets
  __shared__ uint32_t s_part[RS_SC_WAVES];
  const bool three = counters[2] != 0u;
  const uint2* src; uint2* dst;
  rs_route(bufs, pass, three, src, dst);
  const int shift = pass * RS_BITS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NT = RS_SC_WAVES * 64, PER = RS_BINS / NT;   // bins per thread in the scan
  // every global load of the workgroup up front — its elements, its row of the chunk table, the bin totals — so that the
  // scan and the two barriers below run under ONE round trip instead of in front of two more (the compiler keeps loads
  // behind a barrier where it finds them)
  constexpr int STEPS = CHUNK / NT;
  const int wfirst = blockIdx.x * CHUNK + wave * (64 * STEPS);
  uint32_t key[STEPS], val[STEPS];
#pragma unroll
  for (int st = 0; st < STEPS; st++) {
    const int i = wfirst + st * 64 + lane;
    key[st] = 0u; val[st] = 0u;
    if (i < N) {
      if (src) { const uint2 kv = src[i]; key[st] = kv.x; val[st] = kv.y; }
      else { key[st] = bufs.k_in[i]; val[st] = (uint32_t)i; }
    }
  }
  const uint32_t* row = table + (size_t)blockIdx.x * RS_BINS;
  uint32_t rw[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) rw[k] = row[tid * PER + k];
  // exclusive scan of bin_count
  uint32_t c[PER], tsum = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) { c[k] = bin_count[tid * PER + k]; tsum += c[k]; }
  uint32_t v = tsum;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = (uint32_t)__shfl_up((int)v, o);
    if (lane >= o) v += u;
  }
  if (lane == 63) s_part[wave] = v;
  for (int e = tid; e < RS_SC_WAVES * RS_BINS / 2; e += NT) reinterpret_cast<uint32_t*>(&s_wave[0][0])[e] = 0u;
  __syncthreads();
  uint32_t run = v - tsum;
  for (int w = 0; w < wave; w++) run += s_part[w];
#pragma unroll
  for (int k = 0; k < PER; k++) { s_start[tid * PER + k] = run + rw[k]; run += c[k]; }
  // this wave's elements: STEPS steps of 64 consecutive elements
  int dig[STEPS];
#pragma unroll
  for (int st = 0; st < STEPS; st++) {
    const int i = wfirst + st * 64 + lane;
    dig[st] = -1;
    if (i < N) {
      dig[st] = (int)((key[st] >> shift) & (RS_BINS - 1));
      // 16-bit counters packed in pairs (a wave adds at most 64 x STEPS <= 1024 per bin, a chunk at most 8192)
      atomicAdd(reinterpret_cast<uint32_t*>(&s_wave[wave][0]) + (dig[st] >> 1), 1u << (16 * (dig[st] & 1)));
    }
  }
  __syncthreads();
  // counts -> offsets of the waves inside the chunk's segment
  for (int b = tid; b < RS_BINS; b += NT) {
    unsigned short r = 0;
#pragma unroll
    for (int w = 0; w < RS_SC_WAVES; w++) { const unsigned short cw = s_wave[w][b]; s_wave[w][b] = r; r += cw; }
  }
  __syncthreads();
  unsigned short* cur = s_wave[wave];
#pragma unroll
  for (int st = 0; st < STEPS; st++) {
    const bool on = dig[st] >= 0;
    // lanes with the same digit (match-any over the digit's bits)
    uint64_t same = __builtin_amdgcn_ballot_w64(on);
#pragma unroll
    for (int b = 0; b < RS_BITS; b++) {
      const uint64_t vote = __builtin_amdgcn_ballot_w64(on && ((dig[st] >> b) & 1));
      same &= ((dig[st] >> b) & 1) ? vote : ~vote;
    }
    if (on) {
      const uint64_t below = same & ((1ull << lane) - 1ull);
      const uint32_t rank = (uint32_t)__builtin_popcountll(below);
      const unsigned short base = cur[dig[st]];
      const uint32_t pos = s_start[dig[st]] + base + rank;
      if (dst) dst[pos] = make_uint2(key[st], val[st]); else bufs.v_out[pos] = val[st];
      if ((same >> lane) >> 1 == 0ull) cur[dig[st]] = base + (unsigned short)__builtin_popcountll(same);  // highest lane of the group
    }
  }
  if (pass != 1 || !three) return;  // two passes sufficed (the usual case)
  // ---- the third pass, inside this launch (see rs_third_pass_body)
  __shared__ int s_late;
  __atomic_thread_fence(__ATOMIC_RELEASE);  // (agent scope: this workgroup's part of the second pass reaches memory)
  __syncthreads();
  if (tid == 0) {
    s_late = 0;
    __hip_atomic_fetch_add((rs_gu32*)arrivals, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)blockIdx.x < g3) {
      uint32_t spins = 0;
      while (__hip_atomic_load((rs_gu32*)arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) { s_late = 1; break; }
      }
    }
  }
  __syncthreads();
  if ((int)blockIdx.x >= g3) return;
  if (s_late) {  // flag the frame (bit 1 of counters[1]) instead of hanging: its ordering is undefined
    if (tid == 0) atomicOr(counters + 1, 2u);
    return;
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (agent scope: drop what this XCD's caches hold of the second pass's output)
  rs_third_pass_body<CHUNK>(N, (int)gridDim.x, bufs, table3, counters, g3);
}

size_t depth_sort_table_bytes(int N) {
  const size_t n = (size_t)(N > 0 ? N : 1), chunks = (n + RS_CHUNK - 1) / RS_CHUNK;
  return align_up((chunks + 1) * RS_BINS * 4) + align_up(RS_BINS * 4) + 2 * align_up(n * 8) + 256;  // table, bin counts, two lists of pairs, arrival counter
}

// the result ends in (keys_out, vals_out); keys_in is left intact; values are the element indices
template <int CHUNK>
static int launch_depth_sort_t(int N, const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, void* table_mem,
                               const uint32_t* block_info, uint32_t* counters, hipStream_t s) {
  const int chunks = (N + CHUNK - 1) / CHUNK;
  char* mem = (char*)table_mem;
  uint32_t* table = (uint32_t*)mem;
  // (the layout of depth_sort_table_bytes: sized for the small chunks, whichever are used)
  uint32_t* bin_count = (uint32_t*)(mem + align_up(((size_t)((N + RS_CHUNK - 1) / RS_CHUNK) + 1) * RS_BINS * 4));
  RsBufs b;
  (void)keys_out;  // (the sorted keys are not produced: no kernel reads them — the order is the result)
  b.k_in = keys_in;
  b.a = (uint2*)((char*)bin_count + align_up(RS_BINS * 4));
  b.b = (uint2*)((char*)b.a + align_up((size_t)(N > 0 ? N : 1) * 8));
  b.v_out = vals_out;
  // The third pass (inside the second scatter launch) has in-launch barriers: its G workgroups must be RESIDENT at once — half
  // of what the device holds of the scatter kernel (occupancy x compute units), so that the launch's other workgroups, and
  // kernels of other streams (the RCCL collectives of the overlapped exchanges), find compute units.  Each of the G loops over
  // its chunks, so any G is correct; a workgroup that still does not become resident ends the bounded spin and flags the frame.
  static int rs3_grid = 0;
  if (rs3_grid == 0) {
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(rs_scatter_kernel<CHUNK>), RS_SC_WAVES * 64, 0) != hipSuccess ||
        hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
      per_cu = 1; cus = 64; (void)hipGetLastError();
    }
    const int resident = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
    rs3_grid = resident / 2 > 0 ? resident / 2 : 1;
    if (rs3_grid > RS3_MAX_WG) rs3_grid = RS3_MAX_WG;
  }
  uint32_t* arrivals = (uint32_t*)((char*)b.b + align_up((size_t)(N > 0 ? N : 1) * 8));
  for (int pass = 0; pass < 2; pass++) {
    hipLaunchKernelGGL(rs_count_kernel<CHUNK>, dim3(chunks), dim3(CHUNK / 8), 0, s, N, pass, b, table, block_info, (N + 255) / 256, counters,
                       arrivals);
    hipLaunchKernelGGL(rs_scan_kernel, dim3((RS_BINS + 63) / 64), dim3(1024), 0, s, chunks, table, bin_count, counters, pass);
    // (the table of the 12-bit passes is free again when the third pass starts: chunks x 256 counts fit into it)
    hipLaunchKernelGGL(rs_scatter_kernel<CHUNK>, dim3(chunks), dim3(RS_SC_WAVES * 64), 0, s, N, pass, b, table, bin_count, counters,
                       table, arrivals, chunks < rs3_grid ? chunks : rs3_grid);
  }
  return 0;
}

// the result ends in (keys_out, vals_out); keys_in is left intact; values are the element indices
int launch_depth_sort(int N, const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, void* table_mem,
                      const uint32_t* block_info, uint32_t* counters, hipStream_t s) {
  // chunks of 2048 keys fill the chip at the bench size (98 workgroups at 200 k keys; a chunk costs its workgroup an O(4096)
  // set-up); from a million keys on the chunk x bin table of those (16 MB at 2 M) is what the count and scan kernels spend
  // their time on: 8192 keys per chunk there (2 M keys, per pass: count 16 -> 10, scan 21 -> 5.5, scatter 61 -> 56 us;
  // 4096 keys per chunk: 12 / 7 / 60).  The scatter kernel stays at ~55 us whatever the chunk: 4 M scattered 4-byte stores
  // (key and value, up to 64 different bins per wave step) at the rate the L2 takes single-dword line writes — the same
  // bound as the tile sort's walk; runs only get longer with fewer bins (8-bit digits: four passes), not with larger chunks.
  if (N >= RS_BIG_MIN_N) return launch_depth_sort_t<RS_CHUNK_BIG>(N, keys_in, keys_out, vals_out, table_mem, block_info, counters, s);
  return launch_depth_sort_t<RS_CHUNK>(N, keys_in, keys_out, vals_out, table_mem, block_info, counters, s);
}

}  // namespace riggs
