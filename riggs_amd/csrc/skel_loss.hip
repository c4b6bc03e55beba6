// Skeleton projection loss of the trainer (second half):
//   TrainRig.cal_skeleton_loss        /root/reference/train_rig.py:309-314   (weight 1e-3 by default, arguments/__init__.py:184)
//   TrainRig.sampling_skeleton_points /root/reference/train_rig.py:264-276
//   project_nodes_to_2d_elements      /root/reference/utils/other_utils.py:101-127
//   pytorch3d.loss.chamfer_distance(x, y, norm=1)   — third-party, not vendored: restated from its published definition
// S equally spaced points on each of the J-1 bones of the POSED skeleton are projected with the camera's pinhole and
// compared with the M thinned-silhouette pixels of the frame by a two-sided L1 chamfer distance (mean of nearest-neighbour
// distances in either direction).  The reference builds ~20 small tensors and a KNN extension call and lets autograd replay
// them; here the forward is a memset and two launches (all-pairs nearest neighbours of both directions in one grid; a
// fixed-order reduction) and the backward two (per-bone gradient incl. the pull of the pixels on their points, per-joint
// gather), all deterministic: the only atomics are 64-bit max and integer adds.  Sizes are tiny (P = S (J-1) ≈ 0.5-3 k
// points, M ≈ 0.1-5 k pixels): the work is launch-bound, so it is kept to four kernel nodes of a captured iteration.
#include "common.h"

namespace riggs {

struct SkelProjArgs {
  int J, S, M, P;
  const int* parents;
  const float* nodes;   // (J, 3) posed joints
  const float* t;       // (S) line parameters
  const float* view;    // (4, 4) world_view_transform as the reference stores it (row-vector convention)
  float fx, fy, cx, cy;
  const float* thinned; // (M, 2) (row, col)
  const int* m_dev;     // optional device-side pixel count (<= M = the buffer's capacity): one captured graph, any frame
  // state, zeroed by one memset node at the head of the forward:
  unsigned long long* near_x;  // (P) ~(distance bits << 32 | index of the nearest pixel): atomicMax = nearest, lowest index
  unsigned long long* near_y;  // (M) the same for every pixel over the sample points
  float* bone_grad;            // (J-1, 6) child part, parent part
};

__device__ __forceinline__ int skel_pixels(const SkelProjArgs& a) { return a.m_dev ? min(max(a.m_dev[0], 1), a.M) : a.M; }

// point p = s (J-1) + (k-1) lies on bone k (child k, parent parents[k]) at parameter t[s]
__device__ __forceinline__ void skel_point(const SkelProjArgs& a, int p, float& tx, float& ty, float& tz, float& tt, int& k) {
  const int nb = a.J - 1;
  const int s = p / nb;
  k = p - s * nb + 1;
  const int par = a.parents[k];
  tt = a.t[s];
  const float u = 1.0f - tt;
  const float px = tt * a.nodes[3 * k + 0] + u * a.nodes[3 * par + 0];
  const float py = tt * a.nodes[3 * k + 1] + u * a.nodes[3 * par + 1];
  const float pz = tt * a.nodes[3 * k + 2] + u * a.nodes[3 * par + 2];
  const float* V = a.view;
  tx = px * V[0] + py * V[4] + pz * V[8] + V[12];
  ty = px * V[1] + py * V[5] + pz * V[9] + V[13];
  tz = px * V[2] + py * V[6] + pz * V[10] + V[14];
}

__device__ __forceinline__ float2 skel_project(const SkelProjArgs& a, int p) {
  float tx, ty, tz, tt;
  int k;
  skel_point(a, p, tx, ty, tz, tt, k);
  return make_float2(a.fy * ty / tz + a.cy, a.fx * tx / tz + a.cx);
}

// All-pairs nearest neighbours, both directions in one grid of single-wave workgroups: a workgroup takes 64 queries and a
// slice of 256 candidates (staged in LDS, read as broadcasts), and merges its minimum into the query's slot with one 64-bit
// atomicMax of ~(distance, index) — order-independent, so the result is deterministic, and ties keep the lowest index.
// Projections are recomputed wherever they are needed (40 flops) instead of a separate launch and a round trip through HBM.
#define SKEL_Q 64
#define SKEL_C 256
__global__ void __launch_bounds__(SKEL_Q) skel_nearest_kernel(SkelProjArgs a, int n_xblocks, int n_yslices_of_x) {
  __shared__ float2 s_c[SKEL_C];
  const bool xdir = (int)blockIdx.x < n_xblocks;  // queries = sample points, candidates = pixels
  int qb, cs;
  if (xdir) { qb = blockIdx.x / n_yslices_of_x; cs = blockIdx.x - qb * n_yslices_of_x; }
  else {
    const int r = blockIdx.x - n_xblocks, n_slices = (a.P + SKEL_C - 1) / SKEL_C;
    qb = r / n_slices; cs = r - qb * n_slices;
  }
  const int M = skel_pixels(a);
  const int nq = xdir ? a.P : M, nc = xdir ? M : a.P;
  const int i = qb * SKEL_Q + threadIdx.x;
  const int base = cs * SKEL_C, n = min(SKEL_C, nc - base);
  if (n <= 0 || qb * SKEL_Q >= nq) return;  // beyond the frame's pixel count (the grid is sized for the capacity)
  const float2* pix = reinterpret_cast<const float2*>(a.thinned);
  for (int j = threadIdx.x; j < n; j += SKEL_Q) s_c[j] = xdir ? pix[base + j] : skel_project(a, base + j);
  float2 me = make_float2(0.f, 0.f);
  if (i < nq) me = xdir ? skel_project(a, i) : pix[i];
  __syncthreads();
  float best = 3.0e38f;
  int bj = 0;
#pragma unroll 8
  for (int j = 0; j < n; ++j) {
    const float2 o = s_c[j];
    const float d = fabsf(me.x - o.x) + fabsf(me.y - o.y);
    if (d < best) { best = d; bj = j; }
  }
  if (i < nq) {
    const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(base + bj);
    atomicMax((xdir ? a.near_x : a.near_y) + i, ~key);
  }
}

__device__ __forceinline__ float sgn(float v) { return (float)(v > 0.f) - (float)(v < 0.f); }

// One workgroup: loss = mean of the nearest distances of either set, summed in a fixed order.
__global__ void __launch_bounds__(1024) skel_finish_kernel(SkelProjArgs a, const float* weight, float* loss) {
  __shared__ float s_a[1024], s_b[1024];
  float sx = 0.f, sy = 0.f;
  for (int i = threadIdx.x; i < a.P; i += 1024) sx += __uint_as_float((unsigned)(~a.near_x[i] >> 32));
  const int M = skel_pixels(a);
  for (int m = threadIdx.x; m < M; m += 1024) sy += __uint_as_float((unsigned)(~a.near_y[m] >> 32));
  s_a[threadIdx.x] = sx;
  s_b[threadIdx.x] = sy;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      s_a[threadIdx.x] += s_a[threadIdx.x + w];
      s_b[threadIdx.x] += s_b[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float l = s_a[0] / (float)a.P + s_b[0] / (float)M;
    loss[0] = l;
    loss[1] = weight ? weight[0] * l : l;  // the trainer's weighted term (train_rig.py:467-470)
  }
}

// One workgroup per bone.  (i) every pixel that chose a sample point of this bone adds its pull sign(proj - pixel) to the
// point — integer counters in LDS, exact under any order; (ii) gradient of every sample point w.r.t. its projection (its own
// nearest pixel plus the pulls), back through the pinhole and the view transform, split between the bone's two joints;
// (iii) fixed-order reduction over the workgroup.
#define SKEL_MAX_S 2048
__global__ void __launch_bounds__(256) skel_bone_grad_kernel(SkelProjArgs a, const float* g_loss, const float* g_weighted,
                                                             const float* weight) {
  __shared__ int s_pull[SKEL_MAX_S][2];
  __shared__ float s_r[4][6];
  const int nb = a.J - 1;
  const int kb = blockIdx.x;  // bone kb+1
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s = threadIdx.x; s < a.S; s += 256) { s_pull[s][0] = 0; s_pull[s][1] = 0; }
  __syncthreads();
  const int M = skel_pixels(a);
  for (int m = threadIdx.x; m < M; m += 256) {
    const int p = (int)(unsigned)(~a.near_y[m]);
    const int s = p / nb;
    if (p - s * nb != kb) continue;
    const float2 pr = skel_project(a, p);
    const int sy = (int)sgn(pr.x - a.thinned[2 * m + 0]), sx = (int)sgn(pr.y - a.thinned[2 * m + 1]);
    if (sy) atomicAdd(&s_pull[s][0], sy);
    if (sx) atomicAdd(&s_pull[s][1], sx);
  }
  __syncthreads();
  const float g = (g_loss ? g_loss[0] : 0.f) + (g_weighted ? g_weighted[0] * (weight ? weight[0] : 1.f) : 0.f);
  const float wx = g / (float)a.P, wy = g / (float)M;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = threadIdx.x; s < a.S; s += 256) {
    const int p = s * nb + kb;
    float tx, ty, tz, tt;
    int k;
    skel_point(a, p, tx, ty, tz, tt, k);
    const float py = a.fy * ty / tz + a.cy, px = a.fx * tx / tz + a.cx;
    const int ix = (int)(unsigned)(~a.near_x[p]);
    const float gy = wx * sgn(py - a.thinned[2 * ix + 0]) + wy * (float)s_pull[s][0];
    const float gx = wx * sgn(px - a.thinned[2 * ix + 1]) + wy * (float)s_pull[s][1];
    const float iz = 1.0f / tz;
    const float gtx = gx * a.fx * iz, gty = gy * a.fy * iz;
    const float gtz = -(gy * a.fy * ty + gx * a.fx * tx) * iz * iz;
    const float* V = a.view;
    const float g0 = gtx * V[0] + gty * V[1] + gtz * V[2];
    const float g1 = gtx * V[4] + gty * V[5] + gtz * V[6];
    const float g2 = gtx * V[8] + gty * V[9] + gtz * V[10];
    const float u = 1.0f - tt;
    acc[0] += tt * g0; acc[1] += tt * g1; acc[2] += tt * g2;
    acc[3] += u * g0;  acc[4] += u * g1;  acc[5] += u * g2;
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const float v = wave_sum_bcast(acc[c]);
    if (lane == 0) s_r[wave][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6) a.bone_grad[6 * kb + threadIdx.x] = (s_r[0][threadIdx.x] + s_r[1][threadIdx.x]) + (s_r[2][threadIdx.x] + s_r[3][threadIdx.x]);
}

// joint j: its own bone's child part plus the parent parts of its child bones, in ascending bone order
__global__ void skel_joint_gather_kernel(SkelProjArgs a, float* grad_nodes) {
  for (int j = threadIdx.x; j < a.J; j += blockDim.x) {
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (j >= 1) { g0 = a.bone_grad[6 * (j - 1) + 0]; g1 = a.bone_grad[6 * (j - 1) + 1]; g2 = a.bone_grad[6 * (j - 1) + 2]; }
    for (int k = 1; k < a.J; ++k) {
      if (a.parents[k] == j) { g0 += a.bone_grad[6 * (k - 1) + 3]; g1 += a.bone_grad[6 * (k - 1) + 4]; g2 += a.bone_grad[6 * (k - 1) + 5]; }
    }
    grad_nodes[3 * j + 0] = g0; grad_nodes[3 * j + 1] = g1; grad_nodes[3 * j + 2] = g2;
  }
}

static int skel_fill(SkelProjArgs& a, int J, int S, int M, const int32_t* parents, const float* d_nodes, const float* t,
                     const float* view, float fx, float fy, float cx, float cy, const float* thinned, const int32_t* pixel_count,
                     float* state) {
  RIGGS_REQUIRE(J >= 2 && J <= 4096, "need 2..4096 joints");
  RIGGS_REQUIRE(S >= 1 && M >= 1, "empty point set: the reference's mean over it is undefined");
  RIGGS_REQUIRE(S <= SKEL_MAX_S, "more than 2048 samples per bone");
  RIGGS_REQUIRE((size_t)S * (J - 1) < (1u << 30) && M < (1 << 30), "point set too large");
  RIGGS_REQUIRE(parents && d_nodes && t && view && thinned && state, "NULL buffer");
  memset(&a, 0, sizeof(a));
  a.J = J; a.S = S; a.M = M; a.P = S * (J - 1);
  a.parents = parents; a.nodes = d_nodes; a.t = t; a.view = view;
  a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.thinned = thinned; a.m_dev = pixel_count;
  RIGGS_REQUIRE(((uintptr_t)state & 7) == 0, "state must be 8-byte aligned");
  float* f = state;
  a.near_x = (unsigned long long*)f;  f += 2 * (size_t)a.P;
  a.near_y = (unsigned long long*)f;  f += 2 * (size_t)M;
  a.bone_grad = f;
  return 0;
}

}  // namespace riggs

using namespace riggs;

extern "C" {

size_t riggs_skeleton_projection_state_floats(int32_t J, int32_t S, int32_t M) {
  if (J < 2 || S < 0 || M < 0) return 0;
  return 2 * (size_t)S * (J - 1) + 2 * (size_t)M + 6 * (size_t)(J - 1);
}

int riggs_skeleton_projection_forward(int32_t J, int32_t S, int32_t M, const int32_t* parents, const float* d_nodes,
                                      const float* t, const float* world_view_transform, float fx, float fy, float cx,
                                      float cy, const float* thinned, const int32_t* pixel_count, const float* weight, float* state,
                                      float* loss2, riggs_stream stream) {
  SkelProjArgs a;
  if (int rc = skel_fill(a, J, S, M, parents, d_nodes, t, world_view_transform, fx, fy, cx, cy, thinned, pixel_count, state)) return rc;
  RIGGS_REQUIRE(loss2, "NULL buffer");
  hipStream_t s = (hipStream_t)stream;
  // zero = "no neighbour yet" for the complemented keys
  RIGGS_HIP_CHECK(hipMemsetAsync(state, 0, sizeof(float) * (2 * (size_t)a.P + 2 * (size_t)M), s));
  const int ysl = (M + SKEL_C - 1) / SKEL_C, xsl = (a.P + SKEL_C - 1) / SKEL_C;
  const int n_xblocks = ((a.P + SKEL_Q - 1) / SKEL_Q) * ysl, n_yblocks = ((M + SKEL_Q - 1) / SKEL_Q) * xsl;
  hipLaunchKernelGGL(skel_nearest_kernel, dim3(n_xblocks + n_yblocks), dim3(SKEL_Q), 0, s, a, n_xblocks, ysl);
  hipLaunchKernelGGL(skel_finish_kernel, dim3(1), dim3(1024), 0, s, a, weight, loss2);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_skeleton_projection_backward(int32_t J, int32_t S, int32_t M, const int32_t* parents, const float* d_nodes,
                                       const float* t, const float* world_view_transform, float fx, float fy, float cx,
                                       float cy, const float* thinned, const int32_t* pixel_count, const float* weight,
                                       float* state, const float* g_loss, const float* g_weighted, float* grad_nodes,
                                       riggs_stream stream) {
  SkelProjArgs a;
  if (int rc = skel_fill(a, J, S, M, parents, d_nodes, t, world_view_transform, fx, fy, cx, cy, thinned, pixel_count, state)) return rc;
  RIGGS_REQUIRE((g_loss || g_weighted) && grad_nodes, "NULL buffer");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(skel_bone_grad_kernel, dim3(J - 1), dim3(256), 0, s, a, g_loss, g_weighted, weight);
  hipLaunchKernelGGL(skel_joint_gather_kernel, dim3(1), dim3(256), 0, s, a, grad_nodes);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
