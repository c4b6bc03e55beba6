// Tile binning as a STABLE COUNTING SORT, replacing emit + radix sort + ranges.
//
// The Gaussians are already in depth order (stable sort of the depth bits), so the per-tile lists of
// upstream's 64-bit (tile | depth) key sort are exactly a stable partition of the emitted instances
// by tile id.  With T <= a few thousand tiles that is a counting sort:
//   bin_count   : per chunk of depth-ordered Gaussians, a tile histogram in LDS        -> table[chunk][tile]
//   bin_scan    : per tile, exclusive scan over the chunks (table rewritten in place)   -> tile_count[tile]
//   bin_offsets : exclusive scan over the tiles -> ranges[tile], slot_base[tile], R, overflow flag
//   bin_scatter : per chunk: LDS cursors = tile start + chunk offset + offset of the preceding waves,
//                 then every wave walks ITS Gaussians in depth order (one Gaussian per step, lanes = the
//                 tiles of its rectangle) and writes point_list[cursor++] — order-preserving by construction.
// HBM traffic: N*(4+8+4) read per pass, R*4(+4) written once, 2*chunks*T*4 for the table — against
// ~R*32 B for two radix passes over (key, value) pairs plus the emit pass.
#include "raster_internal.h"

namespace riggs {

#define BIN_G_PER_WAVE 128  // Gaussians walked by one wave

__device__ __forceinline__ int rect_tile(const ushort4 rc, int l, int grid_x) {
  const int w = rc.z - rc.x;
  // l / w for small non-negative ints without the integer-division sequence (exact: |error| << 0.5/w)
  const int ry = (int)(((float)l + 0.5f) * __builtin_amdgcn_rcpf((float)w));
  const int rx = l - ry * w;
  return (rc.y + ry) * grid_x + rc.x + rx;
}

__global__ __launch_bounds__(512) void bin_count_kernel(int N, int T, int grid_x, int g_per_block,
                                                        const uint32_t* __restrict__ order,
                                                        const uint32_t* __restrict__ tiles,
                                                        const ushort4* __restrict__ rect,
                                                        uint32_t* __restrict__ table) {
  extern __shared__ uint32_t s_hist[];  // [T]
  for (int t = threadIdx.x; t < T; t += blockDim.x) s_hist[t] = 0u;
  __syncthreads();
  const int first = blockIdx.x * g_per_block;
  const int end = min(N, first + g_per_block);
  for (int s = first + threadIdx.x; s < end; s += blockDim.x) {
    const uint32_t g = order[s];
    const int n = (int)tiles[g];
    if (n == 0) continue;
    const ushort4 rc = rect[g];
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
__global__ __launch_bounds__(1024) void bin_scan_kernel(int T, int n_chunks, uint32_t* __restrict__ table,
                                                        uint32_t* __restrict__ tile_count) {
  __shared__ uint32_t s_seg[SCAN_SEG][64];
  const int tl = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tl;
  const int per = (n_chunks + SCAN_SEG - 1) / SCAN_SEG;
  const int b0 = seg * per, b1 = min(n_chunks, b0 + per);
  uint32_t sum = 0;
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

// exclusive scan over the tiles (one workgroup): ranges, checkpoint slot bases, R and the overflow flag
__global__ __launch_bounds__(1024) void bin_offsets_kernel(int T, int64_t cap, const uint32_t* __restrict__ tile_count,
                                                           uint32_t* __restrict__ tile_start, uint2* __restrict__ ranges,
                                                           uint32_t* __restrict__ slot_base,
                                                           uint32_t* __restrict__ counters) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0u;
  __syncthreads();
  for (int base = 0; base < T; base += 1024) {
    const int t = base + tid;
    const uint32_t c = (t < T) ? tile_count[t] : 0u;
    // inclusive scan inside the wave
    uint32_t v = c;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)v, o);
      if (lane >= o) v += u;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wave; w++) wave_off += s_wave[w];
    const uint32_t carry = s_carry;
    const uint32_t start = carry + wave_off + v - c;
    if (t < T) {
      tile_start[t] = start;
      // clamp to the arena: on overflow (flagged below) the frame is invalid but every access stays in bounds
      const uint32_t lo = (uint32_t)min((int64_t)start, cap), hi = (uint32_t)min((int64_t)start + c, cap);
      // empty tiles keep (0, 0) like upstream's identifyTileRanges (ranges is zeroed before)
      if (hi > lo) ranges[t] = make_uint2(lo, hi);
      slot_base[t] = (lo >> 6) + (uint32_t)t;
    }
    __syncthreads();
    if (tid == 1023) s_carry = carry + wave_off + v;
    __syncthreads();
  }
  if (tid == 0) {
    const uint32_t R = s_carry;
    tile_start[T] = R;
    slot_base[T] = ((uint32_t)min((int64_t)R, cap) >> 6) + (uint32_t)T;
    counters[0] = R;
    counters[1] = ((int64_t)R > cap) ? 1u : 0u;
  }
}

// Scatter.  LDS: s_base[T] (u32 absolute start of this block's segment in each tile) and
// s_rel[W][T] (u16 offsets of each wave's sub-segment, then used as that wave's running cursor).
__global__ __launch_bounds__(512) void bin_scatter_kernel(int N, int T, int grid_x, int64_t cap, int g_per_block,
                                                          const uint32_t* __restrict__ order,
                                                          const uint32_t* __restrict__ tiles,
                                                          const ushort4* __restrict__ rect,
                                                          const uint32_t* __restrict__ table,
                                                          const uint32_t* __restrict__ tile_start,
                                                          uint32_t* __restrict__ point_list,
                                                          uint32_t* __restrict__ tile_keys) {
  extern __shared__ uint32_t s_mem[];
  const int W = blockDim.x >> 6;
  uint32_t* s_base = s_mem;                                            // [T]
  const int Tpad = (T + 1) & ~1;
  unsigned short* s_rel = reinterpret_cast<unsigned short*>(s_mem + T);  // [W][Tpad]
  uint32_t* s_rel32 = s_mem + T;                                        // same storage as packed pairs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < W * (Tpad >> 1); e += blockDim.x) s_rel32[e] = 0u;
  __syncthreads();
  const int first = blockIdx.x * g_per_block + wave * BIN_G_PER_WAVE;
  const int end = min(N, min(first + BIN_G_PER_WAVE, (blockIdx.x + 1) * g_per_block));
  // (i) per-wave tile histogram (16-bit counters packed in pairs; a wave adds at most 128 per tile)
  uint32_t my_g[BIN_G_PER_WAVE / 64], my_n[BIN_G_PER_WAVE / 64];
  ushort4 my_rc[BIN_G_PER_WAVE / 64];
#pragma unroll
  for (int k = 0; k < BIN_G_PER_WAVE / 64; k++) {
    const int s = first + k * 64 + lane;
    my_g[k] = 0u; my_n[k] = 0u; my_rc[k] = make_ushort4(0, 0, 0, 0);
    if (s < end) {
      my_g[k] = order[s];
      my_n[k] = tiles[my_g[k]];
      if (my_n[k]) my_rc[k] = rect[my_g[k]];
    }
    if (my_n[k]) {
      uint32_t* hist = s_rel32 + (size_t)wave * (Tpad >> 1);
      for (int y = my_rc[k].y; y < my_rc[k].w; y++)
        for (int x = my_rc[k].x; x < my_rc[k].z; x++) {
          const int t = y * grid_x + x;
          atomicAdd(&hist[t >> 1], 1u << (16 * (t & 1)));
        }
    }
  }
  __syncthreads();
  // (ii) counts -> offsets of the waves inside the block's segment; absolute base of the segment
  const uint32_t* row = table + (size_t)blockIdx.x * T;
  for (int t = tid; t < T; t += blockDim.x) {
    s_base[t] = tile_start[t] + row[t];
    uint32_t run = 0;
    for (int w = 0; w < W; w++) {
      const unsigned short c = s_rel[(size_t)w * Tpad + t];
      s_rel[(size_t)w * Tpad + t] = (unsigned short)run;
      run += c;
    }
  }
  __syncthreads();
  // (iii) ordered walk: one Gaussian per step, lanes = tiles of its rectangle
  unsigned short* cur = s_rel + (size_t)wave * Tpad;
#pragma unroll
  for (int k = 0; k < BIN_G_PER_WAVE / 64; k++) {
    const uint64_t live = __builtin_amdgcn_ballot_w64(my_n[k] != 0u);
    uint64_t m = live;
    while (m) {
      const int src = __builtin_ctzll(m);
      m &= m - 1;
      const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)my_g[k], src);
      const int n = __builtin_amdgcn_readlane((int)my_n[k], src);
      ushort4 rc;
      const int r0 = __builtin_amdgcn_readlane((int)my_rc[k].x | ((int)my_rc[k].y << 16), src);
      const int r1 = __builtin_amdgcn_readlane((int)my_rc[k].z | ((int)my_rc[k].w << 16), src);
      rc.x = (unsigned short)(r0 & 0xFFFF); rc.y = (unsigned short)((uint32_t)r0 >> 16);
      rc.z = (unsigned short)(r1 & 0xFFFF); rc.w = (unsigned short)((uint32_t)r1 >> 16);
      for (int l = lane; l < n; l += 64) {
        const int t = rect_tile(rc, l, grid_x);
        const unsigned short rel = cur[t];
        cur[t] = rel + 1;
        const int64_t pos = (int64_t)s_base[t] + rel;
        if (pos < cap) { point_list[pos] = g; tile_keys[pos] = (uint32_t)t; }
      }
    }
  }
}

struct BinPlan { int g_per_block, threads, n_chunks; size_t lds_scatter; };

static BinPlan bin_plan(int N, int T) {
  // LDS of the scatter kernel: T*4 (bases) + W*Tpad*2 (wave cursors) <= ~150 KB
  const int Tpad = (T + 1) & ~1;
  int W = 8;
  while (W > 1 && (size_t)T * 4 + (size_t)W * Tpad * 2 > 150 * 1024) W >>= 1;
  BinPlan p;
  p.threads = W * 64;
  p.g_per_block = W * BIN_G_PER_WAVE;
  p.n_chunks = (N + p.g_per_block - 1) / p.g_per_block;
  p.lds_scatter = (size_t)T * 4 + (size_t)W * Tpad * 2;
  return p;
}

size_t bin_table_bytes(int N, int T) {
  BinPlan p = bin_plan(N > 0 ? N : 1, T);
  return align_up(((size_t)p.n_chunks + 1) * T * 4) + align_up((size_t)(T + 1) * 4) * 2;
}

int launch_binning(int N, int T, int grid_x, int64_t cap, const uint32_t* order, const uint32_t* tiles,
                   const ushort4* rect, void* table_mem, uint32_t* point_list, uint32_t* tile_keys, uint2* ranges,
                   uint32_t* slot_base, uint32_t* counters, hipStream_t s) {
  BinPlan p = bin_plan(N, T);
  char* mem = (char*)table_mem;
  uint32_t* table = (uint32_t*)mem;
  uint32_t* tile_count = (uint32_t*)(mem + align_up(((size_t)p.n_chunks + 1) * T * 4));
  uint32_t* tile_start = (uint32_t*)((char*)tile_count + align_up((size_t)(T + 1) * 4));
  if ((size_t)T * 4 > 150 * 1024) { set_error("too many tiles for the LDS histogram (%d)", T); return 2; }
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bin_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bin_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(bin_count_kernel, dim3(p.n_chunks), dim3(p.threads), (size_t)T * 4, s, N, T, grid_x, p.g_per_block,
                     order, tiles, rect, table);
  hipLaunchKernelGGL(bin_scan_kernel, dim3((T + 63) / 64), dim3(1024), 0, s, T, p.n_chunks, table, tile_count);
  hipLaunchKernelGGL(bin_offsets_kernel, dim3(1), dim3(1024), 0, s, T, cap, tile_count, tile_start, ranges, slot_base,
                     counters);
  hipLaunchKernelGGL(bin_scatter_kernel, dim3(p.n_chunks), dim3(p.threads), p.lds_scatter, s, N, T, grid_x, cap,
                     p.g_per_block, order, tiles, rect, table, tile_start, point_list, tile_keys);
  return 0;
}

}  // namespace riggs
