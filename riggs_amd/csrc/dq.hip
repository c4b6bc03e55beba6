// Dual-quaternion blending of rigid transforms on gfx950 — utils/dual_quaternion.py of the reference:
//   QT2DQ :135-143 (normalise q, dual part = standardised (0, t) * q / 2), DQ2QT :146-165, DQBlending :168-179 (weighted sum of
//   the nodes' dual quaternions, then back to rotation + translation), interpolate :182-187, transformation_blending :190-197.
// Two data patterns, one launch per direction each:
//   SHARED  — one set of K <= 1024 nodes for all N rows (skinning: q (K, 4), t (K, 3), weights (N, K)).  The nodes' dual
//             quaternions are built once per workgroup in LDS; a wave takes 64 rows, whose weights are ONE contiguous block of
//             64 x K floats: it is loaded coalesced into an LDS tile (odd row pitch: conflict-free) and read back row-per-lane.
//             Backward: lane <-> node for the (K x N) . (N x 8) contraction dL/ddq_k = sum_n w_nk dL/db_n (register
//             accumulators over the workgroup's tiles, no atomics; per-workgroup partials, a finishing workgroup applies the
//             QT2DQ chain rule), lane <-> row for dL/dw_nk = dL/db_n . dq_k, written back through the same tile.
//   ROWS    — every row has its own K <= 8 transforms (q (N, K, 4), ...; e.g. K nearest control nodes, or the two ends of an
//             interpolation): a thread per row, everything in registers.
// Two properties of the reference are reproduced on purpose (results parity, see oracle/dq_ref.py): a 3-D q is normalised over
// the NODE axis (torch.nn.functional.normalize's default dim=1), and the dual part's sign is standardised independently of
// the real part's.
#include "common.h"

namespace riggs {

#define DQ_MAX_K_SHARED 1024
#define DQ_MAX_K_ROWS 8

struct DqArgs {
  int N, K, norm_nodes, out_mode;  // out_mode 0: R (N, 9); 1: q (N, 4) through matrix_to_quaternion; 2: (N, 16) = [R | t; 0 0 0 1]
  const float *q, *t, *w;
  float *out_rot, *out_t;
  const float *g_rot, *g_t;        // backward: cotangents in the layout of the outputs
  float *gq, *gt, *gw;
  float* partial;                  // SHARED backward: [workgroups][K][8]
  int n_wg;
};

// ---- QT2DQ for one node: qn = normalised quaternion ------------------------------------------------------------------
__device__ __forceinline__ float dq_node(const float qn[4], const float t[3], float dq[8]) {
#pragma clang fp contract(off)
  // quaternion_raw_multiply((0, t), qn) in the reference's operation order (:97-104), then standardize_quaternion (:93-94)
  // (its first operand has a zero real part: the 0 * q terms are dropped, the remaining sums keep their order)
  const float pw = ((-(t[0] * qn[1])) - t[1] * qn[2]) - t[2] * qn[3];
  const float px = (t[0] * qn[0] + t[1] * qn[3]) - t[2] * qn[2];
  const float py = ((-(t[0] * qn[3])) + t[1] * qn[0]) + t[2] * qn[1];
  const float pz = (t[0] * qn[2] - t[1] * qn[1]) + t[2] * qn[0];
  const float s = pw < 0.0f ? -1.0f : 1.0f;
  dq[0] = qn[0]; dq[1] = qn[1]; dq[2] = qn[2]; dq[3] = qn[3];
  dq[4] = s * pw * 0.5f; dq[5] = s * px * 0.5f; dq[6] = s * py * 0.5f; dq[7] = s * pz * 0.5f;
  return s;
}
// its transpose: (dL/dqn, dL/dt) from dL/ddq
__device__ __forceinline__ void dq_node_bwd(const float qn[4], const float t[3], float s, const float g[8], float gqn[4], float gt[3]) {
  const float gw = g[4] * s * 0.5f, gx = g[5] * s * 0.5f, gy = g[6] * s * 0.5f, gz = g[7] * s * 0.5f;
  gqn[0] = g[0] + (t[0] * gx + t[1] * gy + t[2] * gz);
  gqn[1] = g[1] + (-t[0] * gw + t[2] * gy - t[1] * gz);
  gqn[2] = g[2] + (-t[1] * gw - t[2] * gx + t[0] * gz);
  gqn[3] = g[3] + (-t[2] * gw + t[1] * gx - t[0] * gy);
  gt[0] = -qn[1] * gw + qn[0] * gx - qn[3] * gy + qn[2] * gz;
  gt[1] = -qn[2] * gw + qn[3] * gx + qn[0] * gy - qn[1] * gz;
  gt[2] = -qn[3] * gw - qn[2] * gx + qn[1] * gy + qn[0] * gz;
}

// ---- DQ2QT ----------------------------------------------------------------------------------------------------------------
struct DqOut {
  float r[4], d[4], rn, R[9], t[3];
  float q[4], num[4], qa;  // matrix_to_quaternion (out_mode 1)
  int best;
};
__device__ __forceinline__ void dq2qt(const float b[8], DqOut& o, bool want_q) {
  o.rn = fmaxf(sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3]), 1e-8f);
  const float inv = 1.0f / o.rn;
#pragma unroll
  for (int e = 0; e < 4; e++) { o.r[e] = b[e] * inv; o.d[e] = b[4 + e] * inv; }
  const float w0 = o.r[0], x0 = o.r[1], y0 = o.r[2], z0 = o.r[3], w1 = o.d[0], x1 = o.d[1], y1 = o.d[2], z1 = o.d[3];
  o.t[0] = 2.0f * (-w1 * x0 + x1 * w0 - y1 * z0 + z1 * y0);
  o.t[1] = 2.0f * (-w1 * y0 + x1 * z0 + y1 * w0 - z1 * x0);
  o.t[2] = 2.0f * (-w1 * z0 - x1 * y0 + y1 * x0 + z1 * w0);
  o.R[0] = 1.0f - 2.0f * y0 * y0 - 2.0f * z0 * z0; o.R[1] = 2.0f * x0 * y0 - 2.0f * w0 * z0; o.R[2] = 2.0f * x0 * z0 + 2.0f * w0 * y0;
  o.R[3] = 2.0f * x0 * y0 + 2.0f * w0 * z0; o.R[4] = 1.0f - 2.0f * x0 * x0 - 2.0f * z0 * z0; o.R[5] = 2.0f * y0 * z0 - 2.0f * w0 * x0;
  o.R[6] = 2.0f * x0 * z0 - 2.0f * w0 * y0; o.R[7] = 2.0f * y0 * z0 + 2.0f * w0 * x0; o.R[8] = 1.0f - 2.0f * x0 * x0 - 2.0f * y0 * y0;
  if (want_q) {
    // matrix_to_quaternion (:15-74): four candidates, the one with the largest q_abs (the first of equal maxima), / (2 max(q_abs, 0.1))
    const float* m = o.R;
    const float s[4] = {1.0f + m[0] + m[4] + m[8], 1.0f + m[0] - m[4] - m[8], 1.0f - m[0] + m[4] - m[8], 1.0f - m[0] - m[4] + m[8]};
    float qa[4];
#pragma unroll
    for (int e = 0; e < 4; e++) qa[e] = s[e] > 0.0f ? sqrtf(s[e]) : 0.0f;
    int best = 0;
#pragma unroll
    for (int e = 1; e < 4; e++) if (qa[e] > qa[best]) best = e;
    const float qq[4] = {qa[0] * qa[0], qa[1] * qa[1], qa[2] * qa[2], qa[3] * qa[3]};
    float n0, n1, n2, n3;
    if (best == 0) { n0 = qq[0]; n1 = m[7] - m[5]; n2 = m[2] - m[6]; n3 = m[3] - m[1]; }
    else if (best == 1) { n0 = m[7] - m[5]; n1 = qq[1]; n2 = m[3] + m[1]; n3 = m[2] + m[6]; }
    else if (best == 2) { n0 = m[2] - m[6]; n1 = m[3] + m[1]; n2 = qq[2]; n3 = m[5] + m[7]; }
    else { n0 = m[3] - m[1]; n1 = m[6] + m[2]; n2 = m[7] + m[5]; n3 = qq[3]; }
    o.best = best; o.qa = qa[best];
    o.num[0] = n0; o.num[1] = n1; o.num[2] = n2; o.num[3] = n3;
    const float den = 2.0f * fmaxf(o.qa, 0.1f);
#pragma unroll
    for (int e = 0; e < 4; e++) o.q[e] = o.num[e] / den;
  }
}
// dL/d(blended dual quaternion) from the cotangents of the outputs
__device__ __forceinline__ void dq2qt_bwd(const DqOut& o, const float* g_rot, const float g_t[3], bool as_q, float gb[8]) {
  float gR[9];
  if (as_q) {
    const float den = 2.0f * fmaxf(o.qa, 0.1f);
    const float gn[4] = {g_rot[0] / den, g_rot[1] / den, g_rot[2] / den, g_rot[3] / den};
    const float dot = g_rot[0] * o.num[0] + g_rot[1] * o.num[1] + g_rot[2] * o.num[2] + g_rot[3] * o.num[3];
    const float g_qa = o.qa > 0.1f ? -2.0f * dot / (den * den) : 0.0f;
    const float gs = gn[o.best] + (o.qa > 0.0f ? g_qa / (2.0f * o.qa) : 0.0f);
#pragma unroll
    for (int e = 0; e < 9; e++) gR[e] = 0.0f;
    if (o.best == 0) {
      gR[0] = gs; gR[4] = gs; gR[8] = gs;
      gR[7] += gn[1]; gR[5] -= gn[1]; gR[2] += gn[2]; gR[6] -= gn[2]; gR[3] += gn[3]; gR[1] -= gn[3];
    } else if (o.best == 1) {
      gR[0] = gs; gR[4] = -gs; gR[8] = -gs;
      gR[7] += gn[0]; gR[5] -= gn[0]; gR[3] += gn[2]; gR[1] += gn[2]; gR[2] += gn[3]; gR[6] += gn[3];
    } else if (o.best == 2) {
      gR[0] = -gs; gR[4] = gs; gR[8] = -gs;
      gR[2] += gn[0]; gR[6] -= gn[0]; gR[3] += gn[1]; gR[1] += gn[1]; gR[5] += gn[3]; gR[7] += gn[3];
    } else {
      gR[0] = -gs; gR[4] = -gs; gR[8] = gs;
      gR[3] += gn[0]; gR[1] -= gn[0]; gR[6] += gn[1]; gR[2] += gn[1]; gR[7] += gn[2]; gR[5] += gn[2];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 9; e++) gR[e] = g_rot[e];
  }
  const float w0 = o.r[0], x0 = o.r[1], y0 = o.r[2], z0 = o.r[3], w1 = o.d[0], x1 = o.d[1], y1 = o.d[2], z1 = o.d[3];
  const float a = 2.0f * g_t[0], b = 2.0f * g_t[1], c = 2.0f * g_t[2];
  float gr[4], gd[4];
  gr[0] = 2.0f * (-z0 * gR[1] + y0 * gR[2] + z0 * gR[3] - x0 * gR[5] - y0 * gR[6] + x0 * gR[7]) + (x1 * a + y1 * b + z1 * c);
  gr[1] = 2.0f * (y0 * gR[1] + z0 * gR[2] + y0 * gR[3] - 2.0f * x0 * gR[4] - w0 * gR[5] + z0 * gR[6] + w0 * gR[7] - 2.0f * x0 * gR[8]) +
          (-w1 * a - z1 * b + y1 * c);
  gr[2] = 2.0f * (-2.0f * y0 * gR[0] + x0 * gR[1] + w0 * gR[2] + x0 * gR[3] + z0 * gR[5] - w0 * gR[6] + z0 * gR[7] - 2.0f * y0 * gR[8]) +
          (z1 * a - w1 * b - x1 * c);
  gr[3] = 2.0f * (-2.0f * z0 * gR[0] - w0 * gR[1] + x0 * gR[2] + w0 * gR[3] - 2.0f * z0 * gR[4] + y0 * gR[5] + x0 * gR[6] + y0 * gR[7]) +
          (-y1 * a + x1 * b - w1 * c);
  gd[0] = -x0 * a - y0 * b - z0 * c;
  gd[1] = w0 * a + z0 * b - y0 * c;
  gd[2] = -z0 * a + w0 * b + x0 * c;
  gd[3] = y0 * a - x0 * b + w0 * c;
  // r = real / rn, d = imag / rn, rn = |real|
  const float inv = 1.0f / o.rn;
  const float rg = o.r[0] * gr[0] + o.r[1] * gr[1] + o.r[2] * gr[2] + o.r[3] * gr[3];
  const float dg = o.d[0] * gd[0] + o.d[1] * gd[1] + o.d[2] * gd[2] + o.d[3] * gd[3];
#pragma unroll
  for (int e = 0; e < 4; e++) { gb[e] = (gr[e] - o.r[e] * (rg + dg)) * inv; gb[4 + e] = gd[e] * inv; }
}

__device__ __forceinline__ void dq_store(const DqArgs& a, int n, const DqOut& o) {
  if (a.out_mode == 1) {
    reinterpret_cast<float4*>(a.out_rot)[n] = make_float4(o.q[0], o.q[1], o.q[2], o.q[3]);
  } else if (a.out_mode == 0) {
#pragma unroll
    for (int e = 0; e < 9; e++) a.out_rot[(size_t)n * 9 + e] = o.R[e];
  } else {
    float4* T = reinterpret_cast<float4*>(a.out_rot) + (size_t)n * 4;
    T[0] = make_float4(o.R[0], o.R[1], o.R[2], o.t[0]); T[1] = make_float4(o.R[3], o.R[4], o.R[5], o.t[1]);
    T[2] = make_float4(o.R[6], o.R[7], o.R[8], o.t[2]); T[3] = make_float4(0.f, 0.f, 0.f, 1.f);
    return;
  }
  a.out_t[3 * (size_t)n] = o.t[0]; a.out_t[3 * (size_t)n + 1] = o.t[1]; a.out_t[3 * (size_t)n + 2] = o.t[2];
}
// cotangents of row n in the layout of the outputs -> (g_rot[9] | g_rot[4], g_t[3])
__device__ __forceinline__ void dq_load_cotangents(const DqArgs& a, int n, float g_rot[9], float g_t[3]) {
  if (a.out_mode == 1) {
    const float4 v = reinterpret_cast<const float4*>(a.g_rot)[n];
    g_rot[0] = v.x; g_rot[1] = v.y; g_rot[2] = v.z; g_rot[3] = v.w;
  } else if (a.out_mode == 0) {
#pragma unroll
    for (int e = 0; e < 9; e++) g_rot[e] = a.g_rot[(size_t)n * 9 + e];
  } else {
    const float4* T = reinterpret_cast<const float4*>(a.g_rot) + (size_t)n * 4;
    const float4 r0 = T[0], r1 = T[1], r2 = T[2];
    g_rot[0] = r0.x; g_rot[1] = r0.y; g_rot[2] = r0.z; g_rot[3] = r1.x; g_rot[4] = r1.y; g_rot[5] = r1.z;
    g_rot[6] = r2.x; g_rot[7] = r2.y; g_rot[8] = r2.z;
    g_t[0] = r0.w; g_t[1] = r1.w; g_t[2] = r2.w;
    return;
  }
  g_t[0] = a.g_t ? a.g_t[3 * (size_t)n] : 0.f; g_t[1] = a.g_t ? a.g_t[3 * (size_t)n + 1] : 0.f; g_t[2] = a.g_t ? a.g_t[3 * (size_t)n + 2] : 0.f;
}

// ---- SHARED: the nodes' dual quaternions in LDS ---------------------------------------------------------------------------
// s_dq[K][8], s_sgn[K], s_nrm: per-component norms over the nodes (norm_nodes) — per-quaternion norms are recomputed where needed
__device__ void dq_stage_nodes(const DqArgs& a, float* s_dq, float* s_sgn, float* s_nrm /*[4]*/, float* s_red /*[64][4]*/) {
  const int tid = threadIdx.x;
  if (a.norm_nodes) {
    const int c = tid & 3, slice = tid >> 2;  // 64 slices of the node axis per component
    float p = 0.f;
    for (int k = slice; k < a.K; k += 64) { const float v = a.q[4 * k + c]; p += v * v; }
    s_red[slice * 4 + c] = p;
    __syncthreads();
    if (tid < 4) {
      float tot = 0.f;
      for (int s = 0; s < 64; s++) tot += s_red[s * 4 + tid];  // fixed order
      s_nrm[tid] = fmaxf(sqrtf(tot), 1e-12f);
    }
    __syncthreads();
  }
  for (int k = tid; k < a.K; k += blockDim.x) {
    float q[4], qn[4], t[3], dq[8];
#pragma unroll
    for (int e = 0; e < 4; e++) q[e] = a.q[4 * k + e];
#pragma unroll
    for (int e = 0; e < 3; e++) t[e] = a.t[3 * k + e];
    if (a.norm_nodes) {
#pragma unroll
      for (int e = 0; e < 4; e++) qn[e] = q[e] / s_nrm[e];
    } else {
      const float nr = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
#pragma unroll
      for (int e = 0; e < 4; e++) qn[e] = q[e] / nr;
    }
    s_sgn[k] = dq_node(qn, t, dq);
#pragma unroll
    for (int e = 0; e < 8; e++) s_dq[8 * k + e] = dq[e];
  }
  __syncthreads();
}

// a wave's tile of weights: rows [n0, n0 + rows) x columns [k0, k0 + KC) -> tile[r * KP + c]  (KP odd)
__device__ __forceinline__ void dq_load_tile(const float* __restrict__ w, int K, int n0, int rows, int k0, int KC, int KP, float* tile,
                                             int lane) {
  if (K <= 64) {  // the 64 rows are one contiguous block: coalesced, every lane busy
    const int total = rows * K;
    const float* base = w + (size_t)n0 * K;
    int r = lane / K, c = lane - r * K;
    const int dr = 64 / K, dc = 64 - dr * K;
    for (int idx = lane; idx < total; idx += 64) {
      tile[r * KP + c] = base[idx];
      r += dr; c += dc;
      if (c >= K) { c -= K; r++; }
    }
  } else {  // a 256-byte piece of one row per instruction
    for (int r = 0; r < rows; r++)
      if (lane < KC) tile[r * KP + lane] = w[(size_t)(n0 + r) * K + k0 + lane];
  }
}
__device__ __forceinline__ void dq_store_tile(float* __restrict__ gw, int K, int n0, int rows, int k0, int KC, int KP, const float* tile,
                                              int lane) {
  if (K <= 64) {
    const int total = rows * K;
    float* base = gw + (size_t)n0 * K;
    int r = lane / K, c = lane - r * K;
    const int dr = 64 / K, dc = 64 - dr * K;
    for (int idx = lane; idx < total; idx += 64) {
      base[idx] = tile[r * KP + c];
      r += dr; c += dc;
      if (c >= K) { c -= K; r++; }
    }
  } else {
    for (int r = 0; r < rows; r++)
      if (lane < KC) gw[(size_t)(n0 + r) * K + k0 + lane] = tile[r * KP + lane];
  }
}

// dynamic LDS: [K * 8 dq | K sgn | 4 nrm | 256 red | 4 waves x 64 x KP tile (| 4 waves x 64 x 8 row gradients: backward)]
__global__ __launch_bounds__(256) void dqb_shared_fwd_kernel(DqArgs a) {
  extern __shared__ float lds[];
  float* s_dq = lds;
  float* s_sgn = s_dq + 8 * a.K;
  float* s_nrm = s_sgn + a.K;
  float* s_red = s_nrm + 4;
  const int KP = (a.K < 64 ? a.K : 64) | 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* tile = s_red + 256 + wave * 64 * KP;
  dq_stage_nodes(a, s_dq, s_sgn, s_nrm, s_red);
  for (int n0 = (blockIdx.x * 4 + wave) * 64; n0 < a.N; n0 += gridDim.x * 256) {
    const int rows = min(64, a.N - n0);
    float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < a.K; k0 += 64) {
      const int KC = min(64, a.K - k0);
      dq_load_tile(a.w, a.K, n0, rows, k0, KC, KP, tile, lane);
      __builtin_amdgcn_wave_barrier();
      if (lane < rows) {
        for (int c = 0; c < KC; c++) {
          const float wv = tile[lane * KP + c];
          const float4 lo = reinterpret_cast<const float4*>(s_dq)[2 * (k0 + c)], hi = reinterpret_cast<const float4*>(s_dq)[2 * (k0 + c) + 1];
          b[0] += wv * lo.x; b[1] += wv * lo.y; b[2] += wv * lo.z; b[3] += wv * lo.w;
          b[4] += wv * hi.x; b[5] += wv * hi.y; b[6] += wv * hi.z; b[7] += wv * hi.w;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane < rows) {
      DqOut o;
      dq2qt(b, o, a.out_mode == 1);
      dq_store(a, n0 + lane, o);
    }
  }
}

// NCH = chunks of 64 nodes a lane accumulates (K <= 64 * NCH)
template <int NCH>
__global__ __launch_bounds__(256) void dqb_shared_bwd_kernel(DqArgs a) {
  extern __shared__ float lds[];
  float* s_dq = lds;
  float* s_sgn = s_dq + 8 * a.K;
  float* s_nrm = s_sgn + a.K;
  float* s_red = s_nrm + 4;
  const int KP = (a.K < 64 ? a.K : 64) | 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* tile = s_red + 256 + wave * 64 * KP;
  float* s_gb = s_red + 256 + 4 * 64 * KP + wave * 64 * 8;  // this wave's rows' dL/db
  dq_stage_nodes(a, s_dq, s_sgn, s_nrm, s_red);
  float acc[NCH][8];
#pragma unroll
  for (int ch = 0; ch < NCH; ch++)
#pragma unroll
    for (int e = 0; e < 8; e++) acc[ch][e] = 0.f;
  for (int n0 = (blockIdx.x * 4 + wave) * 64; n0 < a.N; n0 += gridDim.x * 256) {
    const int rows = min(64, a.N - n0);
    // pass 1: the blended dual quaternion of the lane's row
    float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < a.K; k0 += 64) {
      const int KC = min(64, a.K - k0);
      if (NCH > 1 || k0 == 0) dq_load_tile(a.w, a.K, n0, rows, k0, KC, KP, tile, lane);
      __builtin_amdgcn_wave_barrier();
      if (lane < rows) {
        for (int c = 0; c < KC; c++) {
          const float wv = tile[lane * KP + c];
          const float4 lo = reinterpret_cast<const float4*>(s_dq)[2 * (k0 + c)], hi = reinterpret_cast<const float4*>(s_dq)[2 * (k0 + c) + 1];
          b[0] += wv * lo.x; b[1] += wv * lo.y; b[2] += wv * lo.z; b[3] += wv * lo.w;
          b[4] += wv * hi.x; b[5] += wv * hi.y; b[6] += wv * hi.z; b[7] += wv * hi.w;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    float gb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (lane < rows) {
      DqOut o;
      dq2qt(b, o, a.out_mode == 1);
      float g_rot[9], g_t[3];
      dq_load_cotangents(a, n0 + lane, g_rot, g_t);
      dq2qt_bwd(o, g_rot, g_t, a.out_mode == 1, gb);
    }
    reinterpret_cast<float4*>(s_gb)[2 * lane] = make_float4(gb[0], gb[1], gb[2], gb[3]);
    reinterpret_cast<float4*>(s_gb)[2 * lane + 1] = make_float4(gb[4], gb[5], gb[6], gb[7]);
    __builtin_amdgcn_wave_barrier();
    // pass 2, per chunk of nodes: lane <-> node: dL/ddq_k += sum_rows w[row][k] dL/db[row]; then lane <-> row: dL/dw[row][k]
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      const int k0 = ch * 64;
      if (k0 < a.K) {
        const int KC = min(64, a.K - k0);
        if (NCH > 1) {  // (one chunk: the tile of pass 1 is still there)
          dq_load_tile(a.w, a.K, n0, rows, k0, KC, KP, tile, lane);
          __builtin_amdgcn_wave_barrier();
        }
        if (lane < KC) {
          for (int r = 0; r < rows; r++) {
            const float wv = tile[r * KP + lane];
            const float4 lo = reinterpret_cast<const float4*>(s_gb)[2 * r], hi = reinterpret_cast<const float4*>(s_gb)[2 * r + 1];
            acc[ch][0] += wv * lo.x; acc[ch][1] += wv * lo.y; acc[ch][2] += wv * lo.z; acc[ch][3] += wv * lo.w;
            acc[ch][4] += wv * hi.x; acc[ch][5] += wv * hi.y; acc[ch][6] += wv * hi.z; acc[ch][7] += wv * hi.w;
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (a.gw) {
          if (lane < rows) {
            for (int c = 0; c < KC; c++) {
              const float4 lo = reinterpret_cast<const float4*>(s_dq)[2 * (k0 + c)], hi = reinterpret_cast<const float4*>(s_dq)[2 * (k0 + c) + 1];
              tile[lane * KP + c] = gb[0] * lo.x + gb[1] * lo.y + gb[2] * lo.z + gb[3] * lo.w + gb[4] * hi.x + gb[5] * hi.y + gb[6] * hi.z + gb[7] * hi.w;
            }
          }
          __builtin_amdgcn_wave_barrier();
          dq_store_tile(a.gw, a.K, n0, rows, k0, KC, KP, tile, lane);
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }
  // the four waves' accumulators -> one partial per workgroup, folded in wave order into ONE [K][8] table (the tiles are free now)
  float* s_acc = s_red + 256;
  for (int wv = 0; wv < 4; wv++) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) {
        const int k = ch * 64 + lane;
        if (k < a.K) {
#pragma unroll
          for (int e = 0; e < 8; e++) s_acc[k * 8 + e] = (wv == 0 ? 0.f : s_acc[k * 8 + e]) + acc[ch][e];
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.K * 8; i += 256) a.partial[(size_t)blockIdx.x * a.K * 8 + i] = s_acc[i];
}

// one workgroup: sums the partials and applies QT2DQ's chain rule per node (and the normalisation's over the nodes)
__global__ __launch_bounds__(256) void dqb_shared_finish_kernel(DqArgs a) {
  extern __shared__ float lds[];
  float* s_g = lds;                 // [K][8] dL/ddq
  float* s_gqn = s_g + 8 * a.K;     // [K][4] dL/dqn
  float* s_nrm = s_gqn + 4 * a.K;   // [4]
  float* s_dot = s_nrm + 4;         // [4]  sum_k qn_kc dL/dqn_kc
  float* s_red = s_dot + 4;         // [64][4]
  const int tid = threadIdx.x;
  for (int i = tid; i < a.K * 8; i += 256) {
    float tot = 0.f;
    for (int g = 0; g < a.n_wg; g++) tot += a.partial[(size_t)g * a.K * 8 + i];
    s_g[i] = tot;
  }
  if (a.norm_nodes) {
    const int c = tid & 3, slice = tid >> 2;
    float p = 0.f;
    for (int k = slice; k < a.K; k += 64) { const float v = a.q[4 * k + c]; p += v * v; }
    s_red[slice * 4 + c] = p;
    __syncthreads();
    if (tid < 4) {
      float tot = 0.f;
      for (int s = 0; s < 64; s++) tot += s_red[s * 4 + tid];
      s_nrm[tid] = fmaxf(sqrtf(tot), 1e-12f);
    }
  }
  __syncthreads();
  for (int k = tid; k < a.K; k += 256) {
    float q[4], qn[4], t[3], dq[8], gqn[4], gt[3];
#pragma unroll
    for (int e = 0; e < 4; e++) q[e] = a.q[4 * k + e];
#pragma unroll
    for (int e = 0; e < 3; e++) t[e] = a.t[3 * k + e];
    float nr = 1.f;
    if (a.norm_nodes) {
#pragma unroll
      for (int e = 0; e < 4; e++) qn[e] = q[e] / s_nrm[e];
    } else {
      nr = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
#pragma unroll
      for (int e = 0; e < 4; e++) qn[e] = q[e] / nr;
    }
    const float s = dq_node(qn, t, dq);
    dq_node_bwd(qn, t, s, s_g + 8 * k, gqn, gt);
#pragma unroll
    for (int e = 0; e < 3; e++) a.gt[3 * k + e] = gt[e];
    if (a.norm_nodes) {
#pragma unroll
      for (int e = 0; e < 4; e++) s_gqn[4 * k + e] = gqn[e];
    } else {
      const float dot = qn[0] * gqn[0] + qn[1] * gqn[1] + qn[2] * gqn[2] + qn[3] * gqn[3];
#pragma unroll
      for (int e = 0; e < 4; e++) a.gq[4 * k + e] = (gqn[e] - qn[e] * dot) / nr;
    }
  }
  if (a.norm_nodes) {
    __syncthreads();
    const int c = tid & 3, slice = tid >> 2;
    float p = 0.f;
    for (int k = slice; k < a.K; k += 64) p += (a.q[4 * k + c] / s_nrm[c]) * s_gqn[4 * k + c];
    s_red[slice * 4 + c] = p;
    __syncthreads();
    if (tid < 4) {
      float tot = 0.f;
      for (int s2 = 0; s2 < 64; s2++) tot += s_red[s2 * 4 + tid];
      s_dot[tid] = tot;
    }
    __syncthreads();
    for (int i = tid; i < a.K * 4; i += 256) {
      const int c2 = i & 3;
      const float qn = a.q[i] / s_nrm[c2];
      a.gq[i] = (s_gqn[i] - qn * s_dot[c2]) / s_nrm[c2];
    }
  }
}

// ---- ROWS: every row its own K <= 8 transforms, a thread per row ---------------------------------------------------------
template <int KK, bool BWD>
__global__ __launch_bounds__(256) void dqb_rows_kernel(DqArgs a) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= a.N) return;
  float q[KK][4], t[KK][3], w[KK], qn[KK][4], dq[KK][8], sg[KK], nrm[KK > 4 ? KK : 4];
#pragma unroll
  for (int k = 0; k < KK; k++) {
    if (k < a.K) {
#pragma unroll
      for (int e = 0; e < 4; e++) q[k][e] = a.q[((size_t)n * a.K + k) * 4 + e];
#pragma unroll
      for (int e = 0; e < 3; e++) t[k][e] = a.t[((size_t)n * a.K + k) * 3 + e];
      w[k] = a.w[(size_t)n * a.K + k];
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++) q[k][e] = 0.f;
#pragma unroll
      for (int e = 0; e < 3; e++) t[k][e] = 0.f;
      w[k] = 0.f;
    }
  }
  if (a.norm_nodes) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < KK; k++) s += q[k][e] * q[k][e];
      nrm[e] = fmaxf(sqrtf(s), 1e-12f);
    }
  } else {
#pragma unroll
    for (int k = 0; k < KK; k++) nrm[k] = fmaxf(sqrtf(q[k][0] * q[k][0] + q[k][1] * q[k][1] + q[k][2] * q[k][2] + q[k][3] * q[k][3]), 1e-12f);
  }
  float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < KK; k++) {
#pragma unroll
    for (int e = 0; e < 4; e++) qn[k][e] = q[k][e] / (a.norm_nodes ? nrm[e] : nrm[k]);
    sg[k] = dq_node(qn[k], t[k], dq[k]);
#pragma unroll
    for (int e = 0; e < 8; e++) b[e] += w[k] * dq[k][e];  // (a padded node has weight 0 and a zero dual quaternion)
  }
  DqOut o;
  dq2qt(b, o, a.out_mode == 1);
  if (!BWD) { dq_store(a, n, o); return; }
  float g_rot[9], g_t[3], gb[8];
  dq_load_cotangents(a, n, g_rot, g_t);
  dq2qt_bwd(o, g_rot, g_t, a.out_mode == 1, gb);
  float gqn[KK][4], dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < KK; k++) {
    if (k < a.K) {
      float gd[8], gtk[3];
      float gwk = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) { gwk += gb[e] * dq[k][e]; gd[e] = gb[e] * w[k]; }
      if (a.gw) a.gw[(size_t)n * a.K + k] = gwk;
      dq_node_bwd(qn[k], t[k], sg[k], gd, gqn[k], gtk);
#pragma unroll
      for (int e = 0; e < 3; e++) a.gt[((size_t)n * a.K + k) * 3 + e] = gtk[e];
      if (a.norm_nodes) {
#pragma unroll
        for (int e = 0; e < 4; e++) dot[e] += qn[k][e] * gqn[k][e];
      } else {
        const float d1 = qn[k][0] * gqn[k][0] + qn[k][1] * gqn[k][1] + qn[k][2] * gqn[k][2] + qn[k][3] * gqn[k][3];
#pragma unroll
        for (int e = 0; e < 4; e++) a.gq[((size_t)n * a.K + k) * 4 + e] = (gqn[k][e] - qn[k][e] * d1) / nrm[k];
      }
    }
  }
  if (a.norm_nodes) {
#pragma unroll
    for (int k = 0; k < KK; k++)
      if (k < a.K) {
#pragma unroll
        for (int e = 0; e < 4; e++) a.gq[((size_t)n * a.K + k) * 4 + e] = (gqn[k][e] - qn[k][e] * dot[e]) / nrm[e];
      }
  }
}

static size_t dq_shared_lds(int K, bool bwd) {
  const int KP = (K < 64 ? K : 64) | 1;
  // (the backward's fold of the four waves' sums, K x 8 floats, reuses the tiles: 4 x 64 x KP >= 8 K)
  const size_t fl = (size_t)8 * K + K + 4 + 256 + (size_t)4 * 64 * KP + (bwd ? 4 * 64 * 8 : 0);
  return fl * 4;
}
static int dq_shared_grid(int N) {
  const int tiles = (N + 255) / 256;
  return tiles < 1 ? 1 : (tiles > 1024 ? 1024 : tiles);
}

}  // namespace riggs

using namespace riggs;

extern "C" {

size_t riggs_dqb_backward_workspace_floats(int32_t N, int32_t K, int32_t shared) {
  return shared ? (size_t)dq_shared_grid(N) * (size_t)(K > 0 ? K : 1) * 8 : 1;
}

static int dq_check(int32_t N, int32_t K, int32_t shared, int32_t out_mode) {
  RIGGS_REQUIRE(N >= 0 && K >= 1, "riggs_dqb: bad sizes");
  RIGGS_REQUIRE(out_mode >= 0 && out_mode <= 2, "riggs_dqb: out_mode is 0 (matrix), 1 (quaternion) or 2 (4x4 transform)");
  if (shared) RIGGS_REQUIRE(K <= DQ_MAX_K_SHARED, "riggs_dqb: at most 1024 shared nodes");
  else RIGGS_REQUIRE(K <= DQ_MAX_K_ROWS, "riggs_dqb: at most 8 transforms per row (use the shared form for a common node set)");
  return 0;
}

int riggs_dqb_forward(int32_t N, int32_t K, int32_t shared, int32_t norm_over_nodes, int32_t out_mode, const float* q, const float* t,
                      const float* weights, float* out_rot, float* out_t, riggs_stream stream) {
  if (int rc = dq_check(N, K, shared, out_mode)) return rc;
  if (N == 0) return 0;
  RIGGS_REQUIRE(q && t && weights && out_rot && (out_t || out_mode == 2), "riggs_dqb_forward: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  DqArgs a{};
  a.N = N; a.K = K; a.norm_nodes = norm_over_nodes ? 1 : 0; a.out_mode = out_mode;
  a.q = q; a.t = t; a.w = weights; a.out_rot = out_rot; a.out_t = out_t;
  if (shared) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dqb_shared_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; }
    hipLaunchKernelGGL(dqb_shared_fwd_kernel, dim3(dq_shared_grid(N)), dim3(256), dq_shared_lds(K, false), s, a);
  } else {
    const dim3 g((N + 255) / 256);
    if (K <= 2) hipLaunchKernelGGL((dqb_rows_kernel<2, false>), g, dim3(256), 0, s, a);
    else if (K <= 4) hipLaunchKernelGGL((dqb_rows_kernel<4, false>), g, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dqb_rows_kernel<8, false>), g, dim3(256), 0, s, a);
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_dqb_backward(int32_t N, int32_t K, int32_t shared, int32_t norm_over_nodes, int32_t out_mode, const float* q, const float* t,
                       const float* weights, const float* g_rot, const float* g_t, float* dL_dq, float* dL_dt, float* dL_dweights,
                       float* workspace, riggs_stream stream) {
  if (int rc = dq_check(N, K, shared, out_mode)) return rc;
  RIGGS_REQUIRE(q && t && dL_dq && dL_dt && workspace, "riggs_dqb_backward: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  DqArgs a{};
  a.N = N; a.K = K; a.norm_nodes = norm_over_nodes ? 1 : 0; a.out_mode = out_mode;
  a.q = q; a.t = t; a.w = weights; a.g_rot = g_rot; a.g_t = g_t; a.gq = dL_dq; a.gt = dL_dt; a.gw = dL_dweights;
  a.partial = workspace;
  if (!shared) {
    if (N == 0) return 0;
    RIGGS_REQUIRE(weights && g_rot, "riggs_dqb_backward: NULL argument");
    const dim3 g((N + 255) / 256);
    if (K <= 2) hipLaunchKernelGGL((dqb_rows_kernel<2, true>), g, dim3(256), 0, s, a);
    else if (K <= 4) hipLaunchKernelGGL((dqb_rows_kernel<4, true>), g, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dqb_rows_kernel<8, true>), g, dim3(256), 0, s, a);
    RIGGS_HIP_CHECK(hipGetLastError());
    return 0;
  }
  a.n_wg = N > 0 ? dq_shared_grid(N) : 0;
  if (N > 0) {
    RIGGS_REQUIRE(weights && g_rot, "riggs_dqb_backward: NULL argument");
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dqb_shared_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dqb_shared_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dqb_shared_bwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      attr = true;
    }
    const size_t lds = dq_shared_lds(K, true);
    if (K <= 64) hipLaunchKernelGGL(dqb_shared_bwd_kernel<1>, dim3(a.n_wg), dim3(256), lds, s, a);
    else if (K <= 256) hipLaunchKernelGGL(dqb_shared_bwd_kernel<4>, dim3(a.n_wg), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(dqb_shared_bwd_kernel<16>, dim3(a.n_wg), dim3(256), lds, s, a);
  }
  hipLaunchKernelGGL(dqb_shared_finish_kernel, dim3(1), dim3(256), (size_t)(12 * K + 8 + 256) * 4, s, a);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
