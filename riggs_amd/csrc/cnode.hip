// Stage-1 control-node deformation, the per-Gaussian part (second half):
//   ControlNodeWarp.cal_nn_weight  /root/reference/utils/time_utils.py:934-964   (KNN over the control nodes + Gaussian kernel)
//   ControlNodeWarp.forward        /root/reference/utils/time_utils.py:1133-1191 (blend of the node deformations)
//   quaternion_to_matrix           /root/reference/utils/time_utils.py:115-132
//   pytorch3d.ops.knn_points       third-party, not vendored: restated as "K smallest squared distances, ascending, ties to
//                                  the lowest index"
// The reference runs a KNN extension call, ~10 gathers of (N, K, ·) tensors, an einsum and a dozen elementwise passes, and
// autograd replays them.  Here: ONE forward launch — the M control nodes (3 + hyper_dim coordinates) sit in LDS, a thread scans
// them for its Gaussian with a register-resident sorted K-list, then blends its K nodes' attributes — and a backward built on
// per-node inverse lists: a counting sort of the N K neighbour entries by node, a per-Gaussian pass that writes each entry's
// 21 + hyper_dim contributions (translation, rotation, scale, dL/dR of the local frame, radius, node weight, hyper coordinates)
// as a 128-byte row at the entry's sorted position, and one workgroup per node that sums its contiguous rows and applies the
// node-level chain rules (quaternion -> matrix, exp, sigmoid).  The first backward (LDS float atomics + per-workgroup partial
// tables) is kept behind riggs_set_option("cnode_bwd_atomics", 1): the LDS executes float atomics lane by lane.  Forward is issue-bound by the
// scan (N M (3 + hyper) FMAs); the tables it reads are L2-resident.
#include "common.h"

namespace riggs {

#define CN_KMAX 8
#define CN_LOCAL_FRAME 1
#define CN_ROT_AS_RES 2

struct CNodeArgs {
  int N, M, K, hyper, feat_stride, node_stride, flags;
  const float* x;         // (N, 3)
  const float* feature;   // (N, feat_stride) or NULL; the first `hyper` columns are the hyper coordinates
  const float* mask;      // (N) or NULL (= 1)
  const float* nodes;     // (M, node_stride): xyz then hyper coordinates
  const float* radius_log;    // (M)  _node_radius
  const float* weight_logit;  // (M)  _node_weight or NULL
  const float* trans;     // (M, 3)
  const float* rot;       // (M, 4)
  const float* scale;     // (M, 3)
  const float* local_rot; // (M, 4) raw (the kernel adds (1, 0, 0, 0)), or NULL
  float* d_xyz; float* d_rot; float* d_scale;   // forward outputs
  int* nn_idx; float* nn_weight; float* nn_dist;  // (N, K)
  // backward
  const float* g_xyz; const float* g_rot; const float* g_scale;  // upstream, each may be NULL
  float* g_feature;   // (N, feat_stride) or NULL
  float* g_mask;      // (N) or NULL
  float* partial;     // (blocks, M, nacc)
  int nacc;
  // inverse-list backward
  float4* rows;       // (N K, 8) float4: the 21 + hyper contributions of every neighbour entry, grouped by node
  int* sorted;        // (N K) position of every entry in the node-grouped order
  int* hist;          // (list blocks, M) entry counts -> exclusive offsets inside the node's segment
  int* node_off;      // (M + 1)
  int list_blocks;
};

__device__ __forceinline__ void cn_quat_to_mat(const float q[4], float R[9]) {
  const float r = q[0], i = q[1], j = q[2], k = q[3];
  const float s = 2.0f / (r * r + i * i + j * j + k * k);
  R[0] = 1.f - s * (j * j + k * k); R[1] = s * (i * j - k * r); R[2] = s * (i * k + j * r);
  R[3] = s * (i * j + k * r); R[4] = 1.f - s * (i * i + k * k); R[5] = s * (j * k - i * r);
  R[6] = s * (i * k - j * r); R[7] = s * (j * k + i * r); R[8] = 1.f - s * (i * i + j * j);
}

// Sorted K-list insert; strict comparisons keep the lower node index on ties.
template <int K>
__device__ __forceinline__ void cn_insert(float (&bd)[K], int (&bi)[K], float d, int j) {
#pragma unroll
  for (int k = K - 1; k >= 0; --k) {
    const bool up = (k > 0) && (d < bd[k > 0 ? k - 1 : 0]);
    if (up) { bd[k] = bd[k - 1 > 0 ? k - 1 : 0]; bi[k] = bi[k - 1 > 0 ? k - 1 : 0]; }
    else if (d < bd[k]) { bd[k] = d; bi[k] = j; }
  }
}

// per-neighbour kernel weight: u = exp(-d / (2 r^2)) * sigmoid(weight logit)
__device__ __forceinline__ void cn_kernel_weight(const CNodeArgs& a, int node, float d, float& e, float& nw, float& r2) {
  const float r = expf(a.radius_log[node]);
  r2 = r * r;
  e = expf(-d / (2.0f * r2));
  nw = a.weight_logit ? 1.0f / (1.0f + expf(-a.weight_logit[node])) : 1.0f;
}

template <int K, int DP4>
__global__ void __launch_bounds__(256) cnode_forward_kernel(CNodeArgs a) {
  extern __shared__ float4 s_nodes[];  // (M, DP4) float4: coordinates padded with zeros
  const int D = 3 + a.hyper;
  for (int e = threadIdx.x; e < a.M * DP4; e += 256) {
    const int n = e / DP4, c = e - n * DP4;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int d = 4 * c + q; v[q] = d < D ? a.nodes[(size_t)n * a.node_stride + d] : 0.f; }
    s_nodes[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.N) return;
  float4 me[DP4];
  {
    float v[4 * DP4];
#pragma unroll
    for (int d = 0; d < 4 * DP4; ++d) v[d] = 0.f;
    v[0] = a.x[3 * (size_t)i + 0]; v[1] = a.x[3 * (size_t)i + 1]; v[2] = a.x[3 * (size_t)i + 2];
#pragma unroll
    for (int d = 3; d < 4 * DP4; ++d) if (d < D) v[d] = a.feature[(size_t)i * a.feat_stride + (d - 3)];
#pragma unroll
    for (int c = 0; c < DP4; ++c) me[c] = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
  }
  float bd[K];
  int bi[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { bd[k] = 3.0e38f; bi[k] = 0; }
  for (int j = 0; j < a.M; ++j) {
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < DP4; ++c) {
      const float4 o = s_nodes[j * DP4 + c];
      const float dx = me[c].x - o.x, dy = me[c].y - o.y, dz = me[c].z - o.z, dw = me[c].w - o.w;
      d = fmaf(dx, dx, d); d = fmaf(dy, dy, d); d = fmaf(dz, dz, d); d = fmaf(dw, dw, d);
    }
    if (d < bd[K - 1]) cn_insert<K>(bd, bi, d, j);
  }
  // weights
  float w[K], vsum = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float e, nw, r2;
    cn_kernel_weight(a, bi[k], bd[k], e, nw, r2);
    w[k] = e * nw + 1e-7f;
    vsum += w[k];
  }
  const float m = a.mask ? a.mask[i] : 1.0f;
  const float x0 = me[0].x, x1 = me[0].y, x2 = me[0].z;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    w[k] = w[k] / vsum;
    const int n = bi[k];
    a.nn_idx[(size_t)i * K + k] = n;
    a.nn_weight[(size_t)i * K + k] = w[k];
    a.nn_dist[(size_t)i * K + k] = bd[k];
    float y0 = a.trans[3 * n + 0], y1 = a.trans[3 * n + 1], y2 = a.trans[3 * n + 2];
    if (a.flags & CN_LOCAL_FRAME) {
      const float q[4] = {a.local_rot[4 * n + 0] + 1.0f, a.local_rot[4 * n + 1], a.local_rot[4 * n + 2], a.local_rot[4 * n + 3]};
      float R[9];
      cn_quat_to_mat(q, R);
      const float4 nd = s_nodes[n * DP4];
      const float r0 = x0 - nd.x, r1 = x1 - nd.y, r2 = x2 - nd.z;
      y0 += R[0] * r0 + R[1] * r1 + R[2] * r2 + nd.x;
      y1 += R[3] * r0 + R[4] * r1 + R[5] * r2 + nd.y;
      y2 += R[6] * r0 + R[7] * r1 + R[8] * r2 + nd.z;
    }
    t0 += w[k] * y0; t1 += w[k] * y1; t2 += w[k] * y2;
    const float bias = (a.flags & CN_ROT_AS_RES) ? 0.f : 1.f;
    q0 += w[k] * (a.rot[4 * n + 0] + bias); q1 += w[k] * a.rot[4 * n + 1]; q2 += w[k] * a.rot[4 * n + 2]; q3 += w[k] * a.rot[4 * n + 3];
    s0 += w[k] * a.scale[3 * n + 0]; s1 += w[k] * a.scale[3 * n + 1]; s2 += w[k] * a.scale[3 * n + 2];
  }
  if (a.flags & CN_LOCAL_FRAME) { t0 -= x0; t1 -= x1; t2 -= x2; }
  a.d_xyz[3 * (size_t)i + 0] = t0 * m; a.d_xyz[3 * (size_t)i + 1] = t1 * m; a.d_xyz[3 * (size_t)i + 2] = t2 * m;
  if (a.flags & CN_ROT_AS_RES) {
    a.d_rot[4 * (size_t)i + 0] = q0 * m;
  } else {
    a.d_rot[4 * (size_t)i + 0] = (q0 - 1.0f) * m + 1.0f;
  }
  a.d_rot[4 * (size_t)i + 1] = q1 * m; a.d_rot[4 * (size_t)i + 2] = q2 * m; a.d_rot[4 * (size_t)i + 3] = q3 * m;
  a.d_scale[3 * (size_t)i + 0] = s0 * m; a.d_scale[3 * (size_t)i + 1] = s1 * m; a.d_scale[3 * (size_t)i + 2] = s2 * m;
}

// accumulator columns of a node: [0:3] translation, [3:7] rotation, [7:10] scale, [10:19] dL/dR (row-major), [19] radius,
// [20] node weight (before the sigmoid's derivative), [21:21+hyper] hyper coordinates
#define CN_ACC_FIXED 21
#define CN_BWD_THREADS 1024

// LISTS = false: per-node gradients through LDS float atomics + one partial table per workgroup (the first version; the LDS
// executes float atomics lane by lane).  LISTS = true: only the per-Gaussian outputs and four scalars per neighbour entry are
// written; the per-node sums are made by cnode_node_reduce_kernel over inverse lists.
template <int K, bool LISTS>
__global__ void __launch_bounds__(CN_BWD_THREADS) cnode_backward_kernel(CNodeArgs a) {
  extern __shared__ float s_acc[];  // (M, nacc)
  const int nacc = a.nacc;
  if (!LISTS) {
    for (int e = threadIdx.x; e < a.M * nacc; e += CN_BWD_THREADS) s_acc[e] = 0.f;
    __syncthreads();
  }
  const bool local = a.flags & CN_LOCAL_FRAME;
  const float bias = (a.flags & CN_ROT_AS_RES) ? 0.f : 1.f;
  for (int i = blockIdx.x * CN_BWD_THREADS + threadIdx.x; i < a.N; i += gridDim.x * CN_BWD_THREADS) {
    const float m = a.mask ? a.mask[i] : 1.0f;
    float g[3] = {0.f, 0.f, 0.f}, h[4] = {0.f, 0.f, 0.f, 0.f}, s[3] = {0.f, 0.f, 0.f};
    if (a.g_xyz) { g[0] = a.g_xyz[3 * (size_t)i]; g[1] = a.g_xyz[3 * (size_t)i + 1]; g[2] = a.g_xyz[3 * (size_t)i + 2]; }
    if (a.g_rot) { h[0] = a.g_rot[4 * (size_t)i]; h[1] = a.g_rot[4 * (size_t)i + 1]; h[2] = a.g_rot[4 * (size_t)i + 2]; h[3] = a.g_rot[4 * (size_t)i + 3]; }
    if (a.g_scale) { s[0] = a.g_scale[3 * (size_t)i]; s[1] = a.g_scale[3 * (size_t)i + 1]; s[2] = a.g_scale[3 * (size_t)i + 2]; }
    const float x0 = a.x[3 * (size_t)i], x1 = a.x[3 * (size_t)i + 1], x2 = a.x[3 * (size_t)i + 2];
    // pass 1: weights and dL/dw
    float w[K], u[K], e_[K], r2_[K], dw[K], rel[K][3];
    float vsum = 0.f;
    float ts[3] = {0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f}, ss[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < K; ++k) {
      {
        const int n = a.nn_idx[(size_t)i * K + k];
        float nw;
        cn_kernel_weight(a, n, a.nn_dist[(size_t)i * K + k], e_[k], nw, r2_[k]);
        u[k] = e_[k] * nw;
        vsum += u[k] + 1e-7f;
      }
    }
    float wdw = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      {
        const int n = a.nn_idx[(size_t)i * K + k];
        w[k] = (u[k] + 1e-7f) / vsum;
        float y0 = a.trans[3 * n], y1 = a.trans[3 * n + 1], y2 = a.trans[3 * n + 2];
        if (local) {
          const float q[4] = {a.local_rot[4 * n] + 1.0f, a.local_rot[4 * n + 1], a.local_rot[4 * n + 2], a.local_rot[4 * n + 3]};
          float R[9];
          cn_quat_to_mat(q, R);
          const float n0 = a.nodes[(size_t)n * a.node_stride], n1 = a.nodes[(size_t)n * a.node_stride + 1], n2 = a.nodes[(size_t)n * a.node_stride + 2];
          const float r0 = x0 - n0, r1 = x1 - n1, r2 = x2 - n2;
          rel[k][0] = r0; rel[k][1] = r1; rel[k][2] = r2;
          y0 += R[0] * r0 + R[1] * r1 + R[2] * r2 + n0;
          y1 += R[3] * r0 + R[4] * r1 + R[5] * r2 + n1;
          y2 += R[6] * r0 + R[7] * r1 + R[8] * r2 + n2;
          // dL/dR += w (g m) (x - n)^T
          if (!LISTS) {
          float* acc = s_acc + (size_t)n * nacc + 10;
          const float wg0 = w[k] * g[0] * m, wg1 = w[k] * g[1] * m, wg2 = w[k] * g[2] * m;
          atomicAdd(acc + 0, wg0 * r0); atomicAdd(acc + 1, wg0 * r1); atomicAdd(acc + 2, wg0 * r2);
          atomicAdd(acc + 3, wg1 * r0); atomicAdd(acc + 4, wg1 * r1); atomicAdd(acc + 5, wg1 * r2);
          atomicAdd(acc + 6, wg2 * r0); atomicAdd(acc + 7, wg2 * r1); atomicAdd(acc + 8, wg2 * r2);
          }
        }
        const float ro0 = a.rot[4 * n] + bias, ro1 = a.rot[4 * n + 1], ro2 = a.rot[4 * n + 2], ro3 = a.rot[4 * n + 3];
        const float c0 = a.scale[3 * n], c1 = a.scale[3 * n + 1], c2 = a.scale[3 * n + 2];
        dw[k] = m * (g[0] * y0 + g[1] * y1 + g[2] * y2 + h[0] * ro0 + h[1] * ro1 + h[2] * ro2 + h[3] * ro3 + s[0] * c0 + s[1] * c1 + s[2] * c2);
        wdw += w[k] * dw[k];
        ts[0] += w[k] * y0; ts[1] += w[k] * y1; ts[2] += w[k] * y2;
        rs[0] += w[k] * ro0; rs[1] += w[k] * ro1; rs[2] += w[k] * ro2; rs[3] += w[k] * ro3;
        ss[0] += w[k] * c0; ss[1] += w[k] * c1; ss[2] += w[k] * c2;
        if (!LISTS) {
        float* acc = s_acc + (size_t)n * nacc;
        const float wm = w[k] * m;
        atomicAdd(acc + 0, wm * g[0]); atomicAdd(acc + 1, wm * g[1]); atomicAdd(acc + 2, wm * g[2]);
        atomicAdd(acc + 3, wm * h[0]); atomicAdd(acc + 4, wm * h[1]); atomicAdd(acc + 5, wm * h[2]); atomicAdd(acc + 6, wm * h[3]);
        atomicAdd(acc + 7, wm * s[0]); atomicAdd(acc + 8, wm * s[1]); atomicAdd(acc + 9, wm * s[2]);
        }
      }
    }
    if (a.g_mask) {
      if (local) { ts[0] -= x0; ts[1] -= x1; ts[2] -= x2; }
      rs[0] -= bias;
      a.g_mask[i] = g[0] * ts[0] + g[1] * ts[1] + g[2] * ts[2] + h[0] * rs[0] + h[1] * rs[1] + h[2] * rs[2] + h[3] * rs[3] +
                    s[0] * ss[0] + s[1] * ss[1] + s[2] * ss[2];
    }
    // pass 2: through the normalisation and the kernel
    float gf[13];
#pragma unroll
    for (int d = 0; d < 13; ++d) gf[d] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      {
        const int n = a.nn_idx[(size_t)i * K + k];
        const float dist = a.nn_dist[(size_t)i * K + k];
        const float dv = (dw[k] - wdw) / vsum;
        float* acc = s_acc + (size_t)n * nacc;
        const float dd = dv * u[k] * (-1.0f / (2.0f * r2_[k]));
        float row[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) row[c] = 0.f;
        if (!LISTS) atomicAdd(acc + 19, dv * u[k] * dist / r2_[k]);
        if (!LISTS && a.weight_logit) atomicAdd(acc + 20, dv * e_[k]);
        if (a.hyper > 0) {
#pragma unroll
          for (int d = 0; d < 13; ++d) {
            if (d < a.hyper) {
              const float diff = a.feature[(size_t)i * a.feat_stride + d] - a.nodes[(size_t)n * a.node_stride + 3 + d];
              const float v = dd * 2.0f * diff;
              gf[d] += v;
              if (!LISTS) atomicAdd(acc + CN_ACC_FIXED + d, -v);
              if (LISTS) row[CN_ACC_FIXED + d] = -v;
            }
          }
        }
        if (LISTS) {
          const float c1 = w[k] * m;
          row[0] = c1 * g[0]; row[1] = c1 * g[1]; row[2] = c1 * g[2];
          row[3] = c1 * h[0]; row[4] = c1 * h[1]; row[5] = c1 * h[2]; row[6] = c1 * h[3];
          row[7] = c1 * s[0]; row[8] = c1 * s[1]; row[9] = c1 * s[2];
          if (local) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
              row[10 + 3 * r] = c1 * g[r] * rel[k][0]; row[11 + 3 * r] = c1 * g[r] * rel[k][1]; row[12 + 3 * r] = c1 * g[r] * rel[k][2];
            }
          }
          row[19] = dv * u[k] * dist / r2_[k];
          row[20] = dv * e_[k];
          float4* dst = a.rows + 8 * (size_t)a.sorted[(size_t)i * K + k];
#pragma unroll
          for (int q = 0; q < 8; ++q) dst[q] = make_float4(row[4 * q], row[4 * q + 1], row[4 * q + 2], row[4 * q + 3]);
        }
      }
    }
    if (a.g_feature) {
      for (int d = 0; d < a.feat_stride; ++d) a.g_feature[(size_t)i * a.feat_stride + d] = 0.f;
#pragma unroll
      for (int d = 0; d < 13; ++d) if (d < a.hyper) a.g_feature[(size_t)i * a.feat_stride + d] = gf[d];
    }
  }
  if (!LISTS) {
    __syncthreads();
    float* out = a.partial + (size_t)blockIdx.x * a.M * nacc;
    for (int e = threadIdx.x; e < a.M * nacc; e += CN_BWD_THREADS) out[e] = s_acc[e];
  }
}

// ---- inverse lists: a counting sort of the N K neighbour entries by node (M <= 2730 bins live in LDS) ---------------------
#define CN_LIST_CHUNK 4096  // entries per workgroup of the histogram / scatter passes

__global__ void __launch_bounds__(256) cnode_list_hist_kernel(CNodeArgs a) {
  extern __shared__ int s_bins[];
  for (int n = threadIdx.x; n < a.M; n += 256) s_bins[n] = 0;
  __syncthreads();
  const int total = a.N * a.K, base = blockIdx.x * CN_LIST_CHUNK;
  for (int e = base + threadIdx.x; e < min(base + CN_LIST_CHUNK, total); e += 256) atomicAdd(&s_bins[a.nn_idx[e]], 1);
  __syncthreads();
  for (int n = threadIdx.x; n < a.M; n += 256) a.hist[(size_t)blockIdx.x * a.M + n] = s_bins[n];
}

// one workgroup: per node an exclusive scan over the list blocks (eight loads in flight), then over the nodes
__global__ void __launch_bounds__(1024) cnode_list_scan_kernel(CNodeArgs a) {
  __shared__ int s_tot[4096];
  for (int n = threadIdx.x; n < a.M; n += 1024) {
    int run = 0, b = 0;
    int* col = a.hist + n;
    for (; b + 8 <= a.list_blocks; b += 8) {
      int t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = col[(size_t)(b + q) * a.M];
#pragma unroll
      for (int q = 0; q < 8; ++q) { col[(size_t)(b + q) * a.M] = run; run += t[q]; }
    }
    for (; b < a.list_blocks; ++b) { const int t = col[(size_t)b * a.M]; col[(size_t)b * a.M] = run; run += t; }
    s_tot[n] = run;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int n = 0; n < a.M; ++n) { a.node_off[n] = run; run += s_tot[n]; }
    a.node_off[a.M] = run;
  }
}

__global__ void __launch_bounds__(256) cnode_list_scatter_kernel(CNodeArgs a) {
  extern __shared__ int s_bins[];
  for (int n = threadIdx.x; n < a.M; n += 256) s_bins[n] = a.node_off[n] + a.hist[(size_t)blockIdx.x * a.M + n];
  __syncthreads();
  const int total = a.N * a.K, base = blockIdx.x * CN_LIST_CHUNK;
  for (int e = base + threadIdx.x; e < min(base + CN_LIST_CHUNK, total); e += 256) {
    a.sorted[e] = atomicAdd(&s_bins[a.nn_idx[e]], 1);
  }
}


struct CNodeGrads {
  float* g_trans; float* g_rot; float* g_scale; float* g_local_rot; float* g_radius_log; float* g_weight_logit;
  float* g_nodes_hyper;  // (M, hyper)
};

// node-level chain rules on the summed accumulator row S of node n; lane c of 32 writes its share
__device__ __forceinline__ void cn_node_chain(const CNodeArgs& a, const CNodeGrads& o, int n, const float* S, int c) {
  if (c < 3) { o.g_trans[3 * n + c] = S[c]; o.g_scale[3 * n + c] = S[7 + c]; }
  if (c < 4) o.g_rot[4 * n + c] = S[3 + c];
  if (c == 4) o.g_radius_log[n] = S[19];
  if (c == 5 && o.g_weight_logit) {
    const float sg = 1.0f / (1.0f + expf(-a.weight_logit[n]));
    o.g_weight_logit[n] = S[20] * sg * (1.0f - sg);
  }
  if (c >= 8 && c < 8 + a.hyper && o.g_nodes_hyper) o.g_nodes_hyper[(size_t)n * a.hyper + (c - 8)] = S[CN_ACC_FIXED + (c - 8)];
  if (c == 6 && o.g_local_rot) {
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.flags & CN_LOCAL_FRAME) {
      // R = I + s A(q), s = 2 / |q|^2
      const float r = a.local_rot[4 * n] + 1.0f, i = a.local_rot[4 * n + 1], j = a.local_rot[4 * n + 2], k = a.local_rot[4 * n + 3];
      const float n2 = r * r + i * i + j * j + k * k, sc = 2.0f / n2;
      const float* G = S + 10;
      const float A[9] = {-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k), j * k - i * r,
                          i * k - j * r, j * k + i * r, -(i * i + j * j)};
      float ds = 0.f, dA[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) { ds += G[e] * A[e]; dA[e] = sc * G[e]; }
      dq[0] = -k * dA[1] + j * dA[2] + k * dA[3] - i * dA[5] - j * dA[6] + i * dA[7];
      dq[1] = j * dA[1] + k * dA[2] + j * dA[3] - 2.f * i * dA[4] - r * dA[5] + k * dA[6] + r * dA[7] - 2.f * i * dA[8];
      dq[2] = -2.f * j * dA[0] + i * dA[1] + r * dA[2] + i * dA[3] + k * dA[5] - r * dA[6] + k * dA[7] - 2.f * j * dA[8];
      dq[3] = -2.f * k * dA[0] - r * dA[1] + i * dA[2] + r * dA[3] - 2.f * k * dA[4] + j * dA[5] + i * dA[6] + j * dA[7];
      const float f = ds * (-2.0f / (n2 * n2)) * 2.0f;
      dq[0] += f * r; dq[1] += f * i; dq[2] += f * j; dq[3] += f * k;
    }
    o.g_local_rot[4 * n] = dq[0]; o.g_local_rot[4 * n + 1] = dq[1]; o.g_local_rot[4 * n + 2] = dq[2]; o.g_local_rot[4 * n + 3] = dq[3];
  }
}

// One workgroup per node: its entries' rows are contiguous (the per-Gaussian pass wrote them in node-grouped order), 32 row
// lanes x 32 columns sum them with four loads in flight each — no atomics, no gathers; then the node-level chain rules.
#define CN_NODE_THREADS 1024
__global__ void __launch_bounds__(CN_NODE_THREADS) cnode_node_reduce_kernel(CNodeArgs a, CNodeGrads o) {
  __shared__ float s_w[32][33];
  __shared__ float s_row[32];
  const int n = blockIdx.x, c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int lo = a.node_off[n], hi = a.node_off[n + 1];
  const float* base = reinterpret_cast<const float*>(a.rows) + c;
  float v4[4] = {0.f, 0.f, 0.f, 0.f};
  int p = lo + r;
  for (; p + 96 < hi; p += 128) {
    float t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = base[32 * (size_t)(p + 32 * q)];
#pragma unroll
    for (int q = 0; q < 4; ++q) v4[q] += t[q];
  }
  for (; p < hi; p += 32) v4[0] += base[32 * (size_t)p];
  s_w[r][c] = (v4[0] + v4[1]) + (v4[2] + v4[3]);
  __syncthreads();
  if (threadIdx.x < 32) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 32; ++w) v += s_w[w][threadIdx.x];
    s_row[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x < 32) cn_node_chain(a, o, n, s_row, threadIdx.x);
}

// 8 nodes per workgroup, 32 lanes per node: lane c sums column c of the workgroups' partial tables in a fixed order, then the
// node-level chain rules.
__global__ void __launch_bounds__(256) cnode_finish_kernel(CNodeArgs a, CNodeGrads o, int blocks) {
  __shared__ float s_v[8][32];
  const int nl = threadIdx.x >> 5, c = threadIdx.x & 31;
  const int n = blockIdx.x * 8 + nl;
  float v = 0.f;
  if (n < a.M && c < a.nacc) {
    // four interleaved accumulators, eight loads in flight: a fixed order, but not one load latency per term
    float v4[4] = {0.f, 0.f, 0.f, 0.f};
    const float* src = a.partial + (size_t)n * a.nacc + c;
    const size_t step = (size_t)a.M * a.nacc;
    int b = 0;
    for (; b + 8 <= blocks; b += 8) {
      float t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = src[(size_t)(b + q) * step];
#pragma unroll
      for (int q = 0; q < 8; ++q) v4[q & 3] += t[q];
    }
    for (; b < blocks; ++b) v4[b & 3] += src[(size_t)b * step];
    v = (v4[0] + v4[1]) + (v4[2] + v4[3]);
  }
  s_v[nl][c] = v;
  __syncthreads();
  if (n >= a.M) return;
  const float* S = s_v[nl];
  cn_node_chain(a, o, n, S, c);
}

static int cn_check(int N, int M, int K, int hyper, int feat_stride, int node_stride, const void* feature) {
  RIGGS_REQUIRE(N >= 0 && M >= 1, "bad sizes");
  RIGGS_REQUIRE(K >= 1 && K <= CN_KMAX && K <= M, "K must be 1..8 and at most the number of nodes");
  RIGGS_REQUIRE(hyper >= 0 && hyper <= 11, "hyper_dim must be 0..11");
  RIGGS_REQUIRE(node_stride >= 3 + hyper, "nodes need 3 + hyper_dim columns");
  RIGGS_REQUIRE(hyper == 0 || (feature && feat_stride >= hyper), "feature needs hyper_dim columns");
  return 0;
}

template <int K>
static void cn_launch_forward(const CNodeArgs& a, int dp4, hipStream_t s) {
  const dim3 grid((a.N + 255) / 256), block(256);
  const size_t lds = (size_t)a.M * dp4 * sizeof(float4);
  switch (dp4) {
    case 1: hipLaunchKernelGGL((cnode_forward_kernel<K, 1>), grid, block, lds, s, a); break;
    case 2: hipLaunchKernelGGL((cnode_forward_kernel<K, 2>), grid, block, lds, s, a); break;
    case 3: hipLaunchKernelGGL((cnode_forward_kernel<K, 3>), grid, block, lds, s, a); break;
    default: hipLaunchKernelGGL((cnode_forward_kernel<K, 4>), grid, block, lds, s, a); break;
  }
}

}  // namespace riggs

using namespace riggs;

extern "C" {

// workgroups of the persistent backward grid: one per CU, two where two gradient tables fit a CU's LDS
int riggs_cnode_backward_blocks(int32_t N, int32_t M, int32_t hyper) {
  const int want = (N + CN_BWD_THREADS - 1) / CN_BWD_THREADS;
  const int cap = (size_t)M * (CN_ACC_FIXED + hyper) * sizeof(float) <= 78 * 1024 ? 512 : 256;
  return want < 1 ? 1 : (want > cap ? cap : want);
}

static int cn_list_blocks(int N, int K) {
  const long long total = (long long)N * K;
  const int b = (int)((total + CN_LIST_CHUNK - 1) / CN_LIST_CHUNK);
  return b < 1 ? 1 : b;
}

static bool cn_use_lists() { return option(OPT_CNODE_BWD_ATOMICS) == 0; }

size_t riggs_cnode_backward_workspace_floats(int32_t N, int32_t M, int32_t K, int32_t hyper) {
  const size_t atomics = (size_t)riggs_cnode_backward_blocks(N, M, hyper) * M * (CN_ACC_FIXED + hyper);
  const size_t lists = 33 * (size_t)N * K + (size_t)cn_list_blocks(N, K) * M + (size_t)M + 1 + 8;
  return atomics > lists ? atomics : lists;
}

int riggs_cnode_forward(int32_t N, int32_t M, int32_t K, int32_t hyper, int32_t feat_stride, int32_t node_stride, int32_t flags,
                        const float* x, const float* feature, const float* motion_mask, const float* nodes,
                        const float* node_radius_log, const float* node_weight_logit, const float* node_trans,
                        const float* node_rot, const float* node_scale, const float* local_rot, float* d_xyz, float* d_rot,
                        float* d_scale, int32_t* nn_idx, float* nn_weight, float* nn_dist, riggs_stream stream) {
  if (int rc = cn_check(N, M, K, hyper, feat_stride, node_stride, feature)) return rc;
  RIGGS_REQUIRE(!(flags & CN_LOCAL_FRAME) || local_rot, "local_frame needs local_rotation");
  if (N == 0) return 0;
  RIGGS_REQUIRE(x && nodes && node_radius_log && node_trans && node_rot && node_scale && d_xyz && d_rot && d_scale && nn_idx &&
                nn_weight && nn_dist, "NULL buffer");
  const int dp4 = (3 + hyper + 3) / 4;
  RIGGS_REQUIRE((size_t)M * dp4 * 16 <= 128 * 1024, "control nodes do not fit the 128 KB LDS table");
  CNodeArgs a;
  memset(&a, 0, sizeof(a));
  a.N = N; a.M = M; a.K = K; a.hyper = hyper; a.feat_stride = feat_stride; a.node_stride = node_stride; a.flags = flags;
  a.x = x; a.feature = feature; a.mask = motion_mask; a.nodes = nodes; a.radius_log = node_radius_log;
  a.weight_logit = node_weight_logit; a.trans = node_trans; a.rot = node_rot; a.scale = node_scale; a.local_rot = local_rot;
  a.d_xyz = d_xyz; a.d_rot = d_rot; a.d_scale = d_scale; a.nn_idx = nn_idx; a.nn_weight = nn_weight; a.nn_dist = nn_dist;
  hipStream_t s = (hipStream_t)stream;
  static unsigned long long attr_set = 0ull;
  if (once_per_device(attr_set)) {
#define CN_ATTR(KK, DD) RIGGS_HIP_CHECK(hipFuncSetAttribute((const void*)cnode_forward_kernel<KK, DD>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
#define CN_ATTR4(KK) CN_ATTR(KK, 1) CN_ATTR(KK, 2) CN_ATTR(KK, 3) CN_ATTR(KK, 4)
    CN_ATTR4(1) CN_ATTR4(2) CN_ATTR4(3) CN_ATTR4(4) CN_ATTR4(5) CN_ATTR4(6) CN_ATTR4(7) CN_ATTR4(8)
  }
  switch (K) {
    case 1: cn_launch_forward<1>(a, dp4, s); break;
    case 2: cn_launch_forward<2>(a, dp4, s); break;
    case 3: cn_launch_forward<3>(a, dp4, s); break;
    case 4: cn_launch_forward<4>(a, dp4, s); break;
    case 5: cn_launch_forward<5>(a, dp4, s); break;
    case 6: cn_launch_forward<6>(a, dp4, s); break;
    case 7: cn_launch_forward<7>(a, dp4, s); break;
    default: cn_launch_forward<8>(a, dp4, s); break;
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_cnode_backward(int32_t N, int32_t M, int32_t K, int32_t hyper, int32_t feat_stride, int32_t node_stride, int32_t flags,
                         const float* x, const float* feature, const float* motion_mask, const float* nodes,
                         const float* node_radius_log, const float* node_weight_logit, const float* node_trans,
                         const float* node_rot, const float* node_scale, const float* local_rot, const int32_t* nn_idx,
                         const float* nn_dist, const float* g_xyz, const float* g_rot, const float* g_scale,
                         float* g_feature, float* g_motion_mask, float* g_node_trans, float* g_node_rot, float* g_node_scale,
                         float* g_local_rot, float* g_node_radius_log, float* g_node_weight_logit, float* g_nodes_hyper,
                         float* workspace, riggs_stream stream) {
  if (int rc = cn_check(N, M, K, hyper, feat_stride, node_stride, feature)) return rc;
  RIGGS_REQUIRE(x && nodes && node_radius_log && node_trans && node_rot && node_scale && (N == 0 || (nn_idx && nn_dist)) &&
                g_node_trans && g_node_rot && g_node_scale && g_node_radius_log && workspace, "NULL buffer");
  RIGGS_REQUIRE(!(flags & CN_LOCAL_FRAME) || local_rot, "local_frame needs local_rotation");
  RIGGS_REQUIRE(hyper == 0 || g_nodes_hyper, "NULL buffer");
  CNodeArgs a;
  memset(&a, 0, sizeof(a));
  a.N = N; a.M = M; a.K = K; a.hyper = hyper; a.feat_stride = feat_stride; a.node_stride = node_stride; a.flags = flags;
  a.x = x; a.feature = feature; a.mask = motion_mask; a.nodes = nodes; a.radius_log = node_radius_log;
  a.weight_logit = node_weight_logit; a.trans = node_trans; a.rot = node_rot; a.scale = node_scale; a.local_rot = local_rot;
  a.nn_idx = const_cast<int32_t*>(nn_idx); a.nn_dist = const_cast<float*>(nn_dist);
  a.g_xyz = g_xyz; a.g_rot = g_rot; a.g_scale = g_scale; a.g_feature = g_feature; a.g_mask = g_motion_mask;
  a.partial = workspace; a.nacc = CN_ACC_FIXED + hyper;
  RIGGS_REQUIRE(a.nacc <= 32, "accumulator row too wide");
  hipStream_t s = (hipStream_t)stream;
  CNodeGrads o = {g_node_trans, g_node_rot, g_node_scale, g_local_rot, g_node_radius_log,
                  node_weight_logit ? g_node_weight_logit : nullptr, g_nodes_hyper};
#define CN_BLAUNCH(LISTS, grid, block, lds)                                                                          \
  switch (K) {                                                                                                       \
    case 1: hipLaunchKernelGGL((cnode_backward_kernel<1, LISTS>), grid, block, lds, s, a); break;                     \
    case 2: hipLaunchKernelGGL((cnode_backward_kernel<2, LISTS>), grid, block, lds, s, a); break;                     \
    case 3: hipLaunchKernelGGL((cnode_backward_kernel<3, LISTS>), grid, block, lds, s, a); break;                     \
    case 4: hipLaunchKernelGGL((cnode_backward_kernel<4, LISTS>), grid, block, lds, s, a); break;                     \
    case 5: hipLaunchKernelGGL((cnode_backward_kernel<5, LISTS>), grid, block, lds, s, a); break;                     \
    case 6: hipLaunchKernelGGL((cnode_backward_kernel<6, LISTS>), grid, block, lds, s, a); break;                     \
    case 7: hipLaunchKernelGGL((cnode_backward_kernel<7, LISTS>), grid, block, lds, s, a); break;                     \
    default: hipLaunchKernelGGL((cnode_backward_kernel<8, LISTS>), grid, block, lds, s, a); break;                    \
  }
  if (cn_use_lists()) {
    // (i) counting sort of the N K neighbour entries by node -> every entry's position; (ii) per-Gaussian pass: outputs of the
    // Gaussians + each entry's 21 + hyper contributions as a 128-byte row at its position; (iii) one workgroup per node sums its
    // (contiguous) rows
    RIGGS_REQUIRE(M <= 4096, "more than 4096 control nodes");
    RIGGS_REQUIRE(((uintptr_t)workspace & 15) == 0, "workspace must be 16-byte aligned");
    a.list_blocks = cn_list_blocks(N, K);
    a.rows = (float4*)workspace;
    a.sorted = (int*)(workspace + 32 * (size_t)N * K);
    a.hist = a.sorted + (size_t)N * K;
    a.node_off = a.hist + (size_t)a.list_blocks * M;
    hipLaunchKernelGGL(cnode_list_hist_kernel, dim3(a.list_blocks), dim3(256), (size_t)M * sizeof(int), s, a);
    hipLaunchKernelGGL(cnode_list_scan_kernel, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(cnode_list_scatter_kernel, dim3(a.list_blocks), dim3(256), (size_t)M * sizeof(int), s, a);
    const int gb = (N + CN_BWD_THREADS - 1) / CN_BWD_THREADS;
    const dim3 grid(gb < 1 ? 1 : gb), block(CN_BWD_THREADS);
    CN_BLAUNCH(true, grid, block, 0)
    hipLaunchKernelGGL(cnode_node_reduce_kernel, dim3(M), dim3(CN_NODE_THREADS), 0, s, a, o);
  } else {
    const size_t lds = (size_t)M * a.nacc * sizeof(float);
    RIGGS_REQUIRE(lds <= 160 * 1024 - 1024, "per-node gradient table does not fit LDS (M (21 + hyper_dim) floats <= 159 KB)");
    static unsigned long long attr_set = 0ull;
    if (once_per_device(attr_set)) {
#define CN_BATTR(KK) RIGGS_HIP_CHECK(hipFuncSetAttribute((const void*)cnode_backward_kernel<KK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      CN_BATTR(1) CN_BATTR(2) CN_BATTR(3) CN_BATTR(4) CN_BATTR(5) CN_BATTR(6) CN_BATTR(7) CN_BATTR(8)
    }
    const int blocks = riggs_cnode_backward_blocks(N, M, hyper);
    const dim3 grid(blocks), block(CN_BWD_THREADS);
    CN_BLAUNCH(false, grid, block, lds)
    hipLaunchKernelGGL(cnode_finish_kernel, dim3((M + 7) / 8), dim3(256), 0, s, a, o, blocks);
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
