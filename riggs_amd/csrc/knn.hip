// simple_knn._C.distCUDA2 replacement (call site /root/reference/scene/gaussian_model.py:170):
// mean squared distance of every point to its 3 nearest neighbours.  Exact search.
//
// Two paths: below 2048 points (P = J skeleton joints, small clouds) the brute-force kernel — the simplest design that keeps the
// chip busy — and from there on a uniform grid (further down).  Brute force: one query point per lane, candidate points streamed through
// LDS in 1024-point tiles (16 KiB as float4, read back with wave-uniform broadcast addresses),
// best-3 kept in registers.  O(P^2) distance evaluations at ~6 VALU ops each.
#include "common.h"

namespace riggs {

#define KNN_TILE 1024

__global__ __launch_bounds__(256) void dist2_knn3_kernel(int P, const float* __restrict__ pts, float* __restrict__ out) {
  __shared__ float4 tile[KNN_TILE];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (i < P) { qx = pts[3 * i]; qy = pts[3 * i + 1]; qz = pts[3 * i + 2]; }
  float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
  for (int base = 0; base < P; base += KNN_TILE) {
    __syncthreads();
    for (int t = threadIdx.x; t < KNN_TILE; t += 256) {
      const int j = base + t;
      tile[t] = (j < P) ? make_float4(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], 0.f)
                        : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
    }
    __syncthreads();
    const int nb = min(KNN_TILE, P - base);
    for (int t = 0; t < nb; t++) {
      const float4 c = tile[t];
      const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
      float d = dx * dx + dy * dy + dz * dz;
      if (base + t == i) d = INFINITY;  // exclude self
      // insert into the sorted triple
      const float m0 = fminf(b0, d), x0 = fmaxf(b0, d);
      const float m1 = fminf(b1, x0), x1 = fmaxf(b1, x0);
      b0 = m0; b1 = m1; b2 = fminf(b2, x1);
    }
  }
  if (i < P) {
    float s = 0.f;
    if (b0 < INFINITY) s += b0;
    if (b1 < INFINITY) s += b1;
    if (b2 < INFINITY) s += b2;
    out[i] = s / 3.0f;
  }
}


// ---- the same answer through a uniform grid (P >= KNN_GRID_MIN_P) -------------------------------------------------------
// Upstream's simple-knn sorts along a Morton curve and prunes boxes; any exact 3-NN gives the same numbers.  Here: a uniform
// grid over the bounding box with ~4 points per cell (at most KNN_MAX_CELLS cells), the points counting-sorted by cell, and
// per point a search of the 3 x 3 x 3 block of cells around its own, widened ring by ring until the third-nearest distance
// found is no larger than the distance to the nearest face of the searched block (then nothing outside can be nearer: exact).
// O(P) for clouds without extreme outliers; an isolated point widens its block up to the whole grid.  The brute-force kernel
// above stays as the small-P path and as this one's test oracle (riggs_dist2_knn3_bruteforce).
#define KNN_GRID_MIN_P 2048
#define KNN_MAX_CELLS (1 << 21)

struct KnnGrid { float ox, oy, oz, inv_h, h; int nx, ny, nz; };

__global__ __launch_bounds__(1024) void knn_grid_setup_kernel(int P, const float* __restrict__ pts, KnnGrid* __restrict__ grid, int max_cells,
                                                             uint32_t* __restrict__ cell_count) {
  __shared__ float s_lo[3][16], s_hi[3][16];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < P; i += 1024)
    for (int a = 0; a < 3; a++) { const float v = pts[3 * (size_t)i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
  for (int a = 0; a < 3; a++) {
    for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
    if ((threadIdx.x & 63) == 0) { s_lo[a][threadIdx.x >> 6] = lo[a]; s_hi[a][threadIdx.x >> 6] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float e[3];
    KnnGrid g;
    for (int a = 0; a < 3; a++) {
      float l = INFINITY, h = -INFINITY;
      for (int w = 0; w < 16; w++) { l = fminf(l, s_lo[a][w]); h = fmaxf(h, s_hi[a][w]); }
      e[a] = fmaxf(h - l, 0.0f);
      (a == 0 ? g.ox : a == 1 ? g.oy : g.oz) = l;
    }
    const float emax = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], 1e-30f));
    // cell edge: ~4 points per cell over the axes that have an extent (a plane or a line of points gets a 2-D / 1-D grid)
    const float target = fmaxf((float)P * 0.25f, 1.0f);
    int dims = 0; float vol = 1.0f;
    for (int a = 0; a < 3; a++) if (e[a] > 1e-6f * emax) { dims++; vol *= e[a]; }
    float h = dims == 0 ? emax : powf(vol / target, 1.0f / (float)dims);
    h = fmaxf(h, emax * (1.0f / 1000.0f));  // (at most ~1000 cells along an axis)
    int nx, ny, nz;
    for (;;) {
      nx = (int)(e[0] / h) + 1; ny = (int)(e[1] / h) + 1; nz = (int)(e[2] / h) + 1;
      if ((long long)nx * ny * nz <= (long long)max_cells) break;
      h *= 1.26f;
    }
    g.h = h; g.inv_h = 1.0f / h; g.nx = nx; g.ny = ny; g.nz = nz;
    *grid = g;
  }
  for (int c = threadIdx.x; c <= max_cells; c += 1024) cell_count[c] = 0u;
}
__device__ __forceinline__ void knn_cell_of(const KnnGrid& g, float x, float y, float z, int& ix, int& iy, int& iz) {
  ix = min(g.nx - 1, max(0, (int)((x - g.ox) * g.inv_h)));
  iy = min(g.ny - 1, max(0, (int)((y - g.oy) * g.inv_h)));
  iz = min(g.nz - 1, max(0, (int)((z - g.oz) * g.inv_h)));
}
__global__ __launch_bounds__(256) void knn_grid_count_kernel(int P, const float* __restrict__ pts, const KnnGrid* __restrict__ grid,
                                                             uint32_t* __restrict__ cell_count, uint32_t* __restrict__ cell_of) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const KnnGrid g = *grid;
  int ix, iy, iz;
  knn_cell_of(g, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], ix, iy, iz);
  const uint32_t c = (uint32_t)((iz * g.ny + iy) * g.nx + ix);
  cell_of[i] = c;
  atomicAdd(&cell_count[c], 1u);
}
// one workgroup: exclusive scan over the cells in place (cell_start), a second copy as the fill cursors
__global__ __launch_bounds__(1024) void knn_grid_scan_kernel(const KnnGrid* __restrict__ grid, uint32_t* __restrict__ cell_start,
                                                            uint32_t* __restrict__ cursor) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  const KnnGrid g = *grid;
  const int n = g.nx * g.ny * g.nz;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0u;
  __syncthreads();
  for (int base = 0; base < n + 1; base += 1024) {
    const int i = base + tid;
    const uint32_t c = i < n ? cell_start[i] : 0u;
    uint32_t v = c;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)v, o); if (lane >= o) v += u; }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    uint32_t off = s_carry;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    if (i <= n) { cell_start[i] = off + v - c; cursor[i] = off + v - c; }
    __syncthreads();
    if (tid == 1023) s_carry = off + v;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void knn_grid_scatter_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ cell_of,
                                                               uint32_t* __restrict__ cursor, float4* __restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const uint32_t pos = atomicAdd(&cursor[cell_of[i]], 1u);  // (the order inside a cell is arbitrary: a best-3 is a set)
  sorted[pos] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __int_as_float(i));
}
__global__ __launch_bounds__(256) void knn_grid_query_kernel(int P, const KnnGrid* __restrict__ grid, const uint32_t* __restrict__ cell_start,
                                                             const float4* __restrict__ sorted, float* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= P) return;
  const KnnGrid g = *grid;
  const float4 q = sorted[t];
  const int self = __float_as_int(q.w);
  int cx, cy, cz;
  knn_cell_of(g, q.x, q.y, q.z, cx, cy, cz);
  float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
  for (int r = 1;; r++) {
    const int x0 = max(0, cx - r), x1 = min(g.nx - 1, cx + r), y0 = max(0, cy - r), y1 = min(g.ny - 1, cy + r);
    const int z0 = max(0, cz - r), z1 = min(g.nz - 1, cz + r);
    for (int z = z0; z <= z1; z++)
      for (int y = y0; y <= y1; y++) {
        // (ring r > 1: the inner block was searched before, only its shell is new — a row of cells is contiguous in the list)
        const bool inner_row = r > 1 && z > cz - r && z < cz + r && y > cy - r && y < cy + r;
        for (int seg = 0; seg < (inner_row ? 2 : 1); seg++) {
          int xa = x0, xb = x1;
          if (inner_row) { if (seg == 0) { xa = cx - r; xb = cx - r; } else { xa = cx + r; xb = cx + r; } if (xa < 0 || xa > g.nx - 1) continue; }
          const uint32_t first = cell_start[(z * g.ny + y) * g.nx + xa], last = cell_start[(z * g.ny + y) * g.nx + xb + 1];
          for (uint32_t k = first; k < last; k++) {
            const float4 c = sorted[k];
            const float dx = c.x - q.x, dy = c.y - q.y, dz = c.z - q.z;
            float d = dx * dx + dy * dy + dz * dz;
            if (__float_as_int(c.w) == self) d = INFINITY;
            const float m0 = fminf(b0, d), e0 = fmaxf(b0, d);
            const float m1 = fminf(b1, e0), e1 = fmaxf(b1, e0);
            b0 = m0; b1 = m1; b2 = fminf(b2, e1);
          }
        }
      }
    // nothing outside the searched block can be nearer than the distance to its nearest face that is not the grid's border
    float face = INFINITY;
    if (x0 > 0) face = fminf(face, q.x - (g.ox + (float)x0 * g.h));
    if (x1 < g.nx - 1) face = fminf(face, (g.ox + (float)(x1 + 1) * g.h) - q.x);
    if (y0 > 0) face = fminf(face, q.y - (g.oy + (float)y0 * g.h));
    if (y1 < g.ny - 1) face = fminf(face, (g.oy + (float)(y1 + 1) * g.h) - q.y);
    if (z0 > 0) face = fminf(face, q.z - (g.oz + (float)z0 * g.h));
    if (z1 < g.nz - 1) face = fminf(face, (g.oz + (float)(z1 + 1) * g.h) - q.z);
    if (face == INFINITY) break;                       // the block is the whole grid
    face = fmaxf(face, 0.0f) * 0.999f;                 // (a hair of slack for the rounding of the cell arithmetic)
    if (b2 <= face * face) break;
  }
  float s = 0.f;
  if (b0 < INFINITY) s += b0;
  if (b1 < INFINITY) s += b1;
  if (b2 < INFINITY) s += b2;
  out[self] = s / 3.0f;
}

}  // namespace riggs

extern "C" {
// [grid parameters 256 B | cell table (max_cells + 1) | cursors (max_cells + 1) | cell of every point P | sorted points P x 16 B]
static int knn_max_cells(int P) {
  int c = 1024;
  while (c < P / 2 && c < KNN_MAX_CELLS) c <<= 1;
  return c;
}
size_t riggs_knn_workspace_bytes(int32_t P) {
  if (P < KNN_GRID_MIN_P) return 256;
  const size_t mc = (size_t)knn_max_cells(P);
  return 256 + 2 * riggs::align_up((mc + 1) * 4) + riggs::align_up((size_t)P * 4) + riggs::align_up((size_t)P * 16);
}

int riggs_dist2_knn3_bruteforce(int32_t P, const float* points, float* out, riggs_stream stream) {
  RIGGS_REQUIRE(P >= 0, "num_points < 0");
  if (P == 0) return 0;
  hipLaunchKernelGGL(riggs::dist2_knn3_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, points, out);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_dist2_knn3(int32_t P, const float* points, float* out, void* workspace, riggs_stream stream) {
  RIGGS_REQUIRE(P >= 0, "num_points < 0");
  if (P < KNN_GRID_MIN_P) return riggs_dist2_knn3_bruteforce(P, points, out, stream);
  RIGGS_REQUIRE(workspace != nullptr, "riggs_dist2_knn3: workspace required");
  using namespace riggs;
  hipStream_t s = (hipStream_t)stream;
  const int mc = knn_max_cells(P);
  char* w = (char*)workspace;
  KnnGrid* grid = (KnnGrid*)w;
  uint32_t* cell_start = (uint32_t*)(w + 256);
  uint32_t* cursor = (uint32_t*)((char*)cell_start + align_up(((size_t)mc + 1) * 4));
  uint32_t* cell_of = (uint32_t*)((char*)cursor + align_up(((size_t)mc + 1) * 4));
  float4* sorted = (float4*)((char*)cell_of + align_up((size_t)P * 4));
  hipLaunchKernelGGL(knn_grid_setup_kernel, dim3(1), dim3(1024), 0, s, P, points, grid, mc, cell_start);
  hipLaunchKernelGGL(knn_grid_count_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, grid, cell_start, cell_of);
  hipLaunchKernelGGL(knn_grid_scan_kernel, dim3(1), dim3(1024), 0, s, grid, cell_start, cursor);
  hipLaunchKernelGGL(knn_grid_scatter_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, cell_of, cursor, sorted);
  hipLaunchKernelGGL(knn_grid_query_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, grid, cell_start, sorted, out);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}
}
