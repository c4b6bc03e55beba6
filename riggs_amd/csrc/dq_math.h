// Dual-quaternion arithmetic shared by the kernels of dq.hip: QT2DQ for one node, DQ2QT, matrix_to_quaternion and their
// transposes (utils/dual_quaternion.py:15-74, 93-113, 135-165 of the reference).  Plain C++: hipcc compiles it for the device,
// and tools/scratch/dq_math_host.cpp compiles the same text for the host to check the hand-derived backward against the oracle.
#pragma once
#include <math.h>
#ifdef __HIPCC__
#define DQ_FN __device__ __forceinline__
#else
#define DQ_FN static inline
#endif

namespace riggs {

// ---- QT2DQ for one node: qn = normalised quaternion ------------------------------------------------------------------
DQ_FN float dq_node(const float qn[4], const float t[3], float dq[8]) {
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
DQ_FN void dq_node_bwd(const float qn[4], const float t[3], float s, const float g[8], float gqn[4], float gt[3]) {
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
  float q[4];  // matrix_to_quaternion (out_mode 1)
};
DQ_FN void dq2qt(const float b[8], DqOut& o, bool want_q) {
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
    const float den = 2.0f * fmaxf(qa[best], 0.1f);
    o.q[0] = n0 / den; o.q[1] = n1 / den; o.q[2] = n2 / den; o.q[3] = n3 / den;
  }
}
// dL/d(blended dual quaternion) from the cotangents of the outputs
DQ_FN void dq2qt_bwd(const DqOut& o, const float* g_rot, const float g_t[3], bool as_q, float gb[8]) {
  // Rotation cotangent.  as_q: q = matrix_to_quaternion(R(r)) is +-r for the unit r of DQ2QT, whichever of the four candidates
  // is picked; the candidates' Jacobians differ only along r itself (where R(r), which is not re-normalised, leaves the
  // rotations), and that component is projected out by r = real / |real| below.  So the cotangent of r is sign(q . r) g_q —
  // exactly what autograd's walk through the selected candidate gives after the projection (oracle/dq_ref.py walks the
  // candidate explicitly and is pinned by the reference's autograd: tests/test_oracle_dq.py; the two agree to rounding).
  float gR[9], gqr[4] = {0.f, 0.f, 0.f, 0.f};
  if (as_q) {
    const float sg = (o.q[0] * o.r[0] + o.q[1] * o.r[1] + o.q[2] * o.r[2] + o.q[3] * o.r[3]) < 0.0f ? -1.0f : 1.0f;
#pragma unroll
    for (int e = 0; e < 4; e++) gqr[e] = sg * g_rot[e];
#pragma unroll
    for (int e = 0; e < 9; e++) gR[e] = 0.0f;
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
#pragma unroll
  for (int e = 0; e < 4; e++) gr[e] += gqr[e];
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

// 16-byte accesses (device: one instruction; host: four scalars)
#ifdef __HIPCC__
#define DQ_STORE4(p, a, b, c, d) (*reinterpret_cast<float4*>(p) = make_float4(a, b, c, d))
#define DQ_LOAD4(p, a, b, c, d) do { const float4 _v = *reinterpret_cast<const float4*>(p); a = _v.x; b = _v.y; c = _v.z; d = _v.w; } while (0)
#else
#include <stddef.h>
#define DQ_STORE4(p, a, b, c, d) do { (p)[0] = a; (p)[1] = b; (p)[2] = c; (p)[3] = d; } while (0)
#define DQ_LOAD4(p, a, b, c, d) do { a = (p)[0]; b = (p)[1]; c = (p)[2]; d = (p)[3]; } while (0)
#endif

struct DqArgs {
  int N, K, norm_nodes, out_mode;  // out_mode 0: R (N, 9); 1: q (N, 4) through matrix_to_quaternion; 2: (N, 16) = [R | t; 0 0 0 1]
  const float *q, *t, *w;
  float *out_rot, *out_t;
  const float *g_rot, *g_t;        // backward: cotangents in the layout of the outputs
  float *gq, *gt, *gw;
  float* partial;                  // SHARED backward: [workgroups][K][8]
  int n_wg;
};

DQ_FN void dq_store(const DqArgs& a, int n, const DqOut& o) {
  if (a.out_mode == 1) {
    DQ_STORE4(a.out_rot + 4 * (size_t)n, o.q[0], o.q[1], o.q[2], o.q[3]);
  } else if (a.out_mode == 0) {
#pragma unroll
    for (int e = 0; e < 9; e++) a.out_rot[(size_t)n * 9 + e] = o.R[e];
  } else {
    float* T = a.out_rot + (size_t)n * 16;
    DQ_STORE4(T, o.R[0], o.R[1], o.R[2], o.t[0]); DQ_STORE4(T + 4, o.R[3], o.R[4], o.R[5], o.t[1]);
    DQ_STORE4(T + 8, o.R[6], o.R[7], o.R[8], o.t[2]); DQ_STORE4(T + 12, 0.f, 0.f, 0.f, 1.f);
    return;
  }
  a.out_t[3 * (size_t)n] = o.t[0]; a.out_t[3 * (size_t)n + 1] = o.t[1]; a.out_t[3 * (size_t)n + 2] = o.t[2];
}
// cotangents of row n in the layout of the outputs -> (g_rot[9] | g_rot[4], g_t[3])
DQ_FN void dq_load_cotangents(const DqArgs& a, int n, float g_rot[9], float g_t[3]) {
  if (a.out_mode == 1) {
    DQ_LOAD4(a.g_rot + 4 * (size_t)n, g_rot[0], g_rot[1], g_rot[2], g_rot[3]);
  } else if (a.out_mode == 0) {
#pragma unroll
    for (int e = 0; e < 9; e++) g_rot[e] = a.g_rot[(size_t)n * 9 + e];
  } else {
    const float* T = a.g_rot + (size_t)n * 16;
    DQ_LOAD4(T, g_rot[0], g_rot[1], g_rot[2], g_t[0]);
    DQ_LOAD4(T + 4, g_rot[3], g_rot[4], g_rot[5], g_t[1]);
    DQ_LOAD4(T + 8, g_rot[6], g_rot[7], g_rot[8], g_t[2]);
    return;
  }
  g_t[0] = a.g_t ? a.g_t[3 * (size_t)n] : 0.f; g_t[1] = a.g_t ? a.g_t[3 * (size_t)n + 1] : 0.f; g_t[2] = a.g_t ? a.g_t[3 * (size_t)n + 2] : 0.f;
}

// ---- one row of the ROWS form (a thread per row on the device): forward, or forward + backward ------------------------
template <int KK, bool BWD>
DQ_FN void dq_row(const DqArgs& a, int n) {
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


}  // namespace riggs
