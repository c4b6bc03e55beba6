// simple_knn._C.distCUDA2 replacement (call site /root/reference/scene/gaussian_model.py:170):
// mean squared distance of every point to its 3 nearest neighbours.  Exact search.
//
// Init-time only (P <= 150k points, or P = J skeleton joints), so the design is the simplest
// one that keeps the chip busy: one query point per lane, candidate points streamed through
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

}  // namespace riggs

extern "C" {
size_t riggs_knn_workspace_bytes(int32_t) { return 256; }

int riggs_dist2_knn3(int32_t P, const float* points, float* out, void* workspace, riggs_stream stream) {
  (void)workspace;
  RIGGS_REQUIRE(P >= 0, "num_points < 0");
  if (P == 0) return 0;
  hipLaunchKernelGGL(riggs::dist2_knn3_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, points, out);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}
}
