// Forward kinematics over <= 64 joints on one wave64, device side: shared by the FK kernels (deform.hip), the skinning forward that runs the
// chain itself, and the PoseMLP backward that runs the reverse sweep in front of its own chain (pose_mlp.hip).
//   quaternion_to_matrix      utils/time_utils.py:115-132
//   chain_product_transform   skeleton_utils/skeleton_warp.py:242-273
//   matrix_to_quaternion      utils/time_utils.py:146-205
#pragma once
#include "common.h"

namespace riggs {

#define MAX_J 64
// quaternion_to_matrix with two_s = 2/|q|^2 (utils/time_utils.py:115-132)
__device__ __forceinline__ void quat_to_R_unnorm(const float* q, float* R) {
  const float r = q[0], i = q[1], j = q[2], k = q[3];
  const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
  R[0] = 1.f - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
  R[3] = two_s * (i * j + k * r); R[4] = 1.f - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
  R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1.f - two_s * (i * i + j * j);
}

// matrix_to_quaternion (utils/time_utils.py:146-205): best-conditioned candidate, no sign fix
__device__ __forceinline__ void R_to_quat(const float* m, float* q) {
  const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[4], m11 = m[5], m12 = m[6], m20 = m[8], m21 = m[9], m22 = m[10];
  float a[4] = {1.0f + m00 + m11 + m22, 1.0f + m00 - m11 - m22, 1.0f - m00 + m11 - m22, 1.0f - m00 - m11 + m22};
  float qa[4];
  int pick = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) qa[c] = a[c] > 0.f ? sqrtf(a[c]) : 0.f;
#pragma unroll
  for (int c = 1; c < 4; c++) if (qa[c] > qa[pick]) pick = c;
  float cand[4];
  if (pick == 0) { cand[0] = qa[0] * qa[0]; cand[1] = m21 - m12; cand[2] = m02 - m20; cand[3] = m10 - m01; }
  else if (pick == 1) { cand[0] = m21 - m12; cand[1] = qa[1] * qa[1]; cand[2] = m10 + m01; cand[3] = m02 + m20; }
  else if (pick == 2) { cand[0] = m02 - m20; cand[1] = m10 + m01; cand[2] = qa[2] * qa[2]; cand[3] = m12 + m21; }
  else { cand[0] = m10 - m01; cand[1] = m20 + m02; cand[2] = m21 + m12; cand[3] = qa[3] * qa[3]; }
  const float den = 2.0f * fmaxf(qa[pick], 0.1f);
#pragma unroll
  for (int c = 0; c < 4; c++) q[c] = cand[c] / den;
}

// The chain on ONE wave64, one joint per lane, everything in registers.  The chain is a TREE, and a joint's transform only
// waits for its ancestors: the sweeps go level by level (tree depth steps — 6 to 10 for a body — instead of J - 1), a joint
// reads its parent's / its children's values with lane shuffles, and a parent adds up its children in descending index order
// like the sequential sweep i = J - 1 .. 1 did: the same arithmetic per joint, the same order per sum.
// (The first version kept T and G in LDS and walked the joints in index order with a barrier per joint: 2 x 23 + 23
// dependent LDS round trips — 6.5 us forward and 10 us backward, launch or no launch.)
// Call with all 64 lanes of the wave active; lanes >= J carry harmless values.
struct FkLane {
  float T[12], G[12];  // local and global transform of the lane's joint, 3x4 row-major [R|t]
  int par, lev, maxlev;
  unsigned long long kids;  // the joint's children (bit i: joint i)
};
__device__ __forceinline__ float fk_lane_get(float v, int src) { return __shfl(v, src); }

// what a lane reads from memory for its joint (a caller with other loads to issue fetches these first: fk_load)
struct FkIn {
  float q[4];      // local rotation (un-normalised wxyz)
  float c[3];      // the PARENT joint's rest position
  float x[3];      // the joint's own rest position
  float dG[12];    // (backward) dL/dtransforms of the joint
  float gn[3];     // (backward) dL/dd_nodes of the joint, or zeros
  float G[12];     // (backward, optional) the joint's global transform as the forward left it: the chain is then not re-run
  int par;
};
__device__ __forceinline__ void fk_load(int J, const float* __restrict__ local_rot, const float* __restrict__ joints,
                                        const int32_t* __restrict__ parents, const float* __restrict__ dL_dG_in,
                                        const float* __restrict__ dL_dnodes, FkIn& in, const float* __restrict__ transforms = nullptr) {
  const int j = threadIdx.x & 63;
  in.par = 0;
#pragma unroll
  for (int e = 0; e < 4; e++) in.q[e] = (e == 0) ? 1.f : 0.f;
#pragma unroll
  for (int e = 0; e < 3; e++) { in.c[e] = 0.f; in.x[e] = 0.f; in.gn[e] = 0.f; }
#pragma unroll
  for (int e = 0; e < 12; e++) { in.dG[e] = 0.f; in.G[e] = 0.f; }
  if (j < J) {
    if (transforms) {
#pragma unroll
      for (int e = 0; e < 12; e++) in.G[e] = transforms[12 * j + e];
    }
    const int vp = (j == 0) ? 0 : parents[j];  // skeleton_warp.py:246-247
    in.par = vp;
#pragma unroll
    for (int e = 0; e < 4; e++) in.q[e] = local_rot[4 * j + e];
#pragma unroll
    for (int e = 0; e < 3; e++) { in.x[e] = joints[3 * j + e]; in.c[e] = joints[3 * vp + e]; }
    if (dL_dG_in) {
#pragma unroll
      for (int e = 0; e < 12; e++) in.dG[e] = dL_dG_in[12 * j + e];
    }
    if (dL_dnodes) {
#pragma unroll
      for (int e = 0; e < 3; e++) in.gn[e] = dL_dnodes[3 * j + e];
    }
  }
}

__device__ __forceinline__ void fk_wave_forward(int J, const FkIn& in, FkLane& f, bool have_G = false) {
  const int j = threadIdx.x & 63;
  const bool on = j < J;
  f.par = 0;
#pragma unroll
  for (int e = 0; e < 12; e++) { f.T[e] = 0.f; f.G[e] = 0.f; }
  if (on) {
    float q[4] = {in.q[0], in.q[1], in.q[2], in.q[3]};
    float R[9];
    quat_to_R_unnorm(q, R);
    const int vp = in.par;
    f.par = vp;
    const float cx = in.c[0], cy = in.c[1], cz = in.c[2];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      f.T[4 * r] = R[3 * r]; f.T[4 * r + 1] = R[3 * r + 1]; f.T[4 * r + 2] = R[3 * r + 2];
      const float c = (r == 0) ? cx : (r == 1 ? cy : cz);
      f.T[4 * r + 3] = c - (R[3 * r] * cx + R[3 * r + 1] * cy + R[3 * r + 2] * cz);  // rotate about the PARENT joint
    }
  }
  // depth of every joint by pointer doubling (parents[i] < i: joint 0 is the root), the deepest level, the children
  int d = (on && j >= 1) ? 1 : 0, anc = f.par;
#pragma unroll
  for (int s = 0; s < 6; s++) {
    const int d2 = __shfl(d, anc), a2 = __shfl(anc, anc);
    d += d2; anc = a2;
  }
  f.lev = d;
  int m = d;
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  f.maxlev = m;
  f.kids = 0ull;
  for (int p = 0; p < J; p++) {
    const unsigned long long c = __builtin_amdgcn_ballot_w64(on && j >= 1 && f.par == p);
    if (j == p) f.kids = c;
  }
  // G_i = G_parent(i) * T_i: after step k the joints of depth <= k are final (have_G: the forward's result is given)
#pragma unroll
  for (int e = 0; e < 12; e++) f.G[e] = have_G ? in.G[e] : f.T[e];
  for (int l = 1; !have_G && l <= f.maxlev; l++) {
    float Gp[12];
#pragma unroll
    for (int e = 0; e < 12; e++) Gp[e] = fk_lane_get(f.G[e], f.par);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        float v = Gp[4 * r] * f.T[c] + Gp[4 * r + 1] * f.T[4 + c] + Gp[4 * r + 2] * f.T[8 + c];
        if (c == 3) v += Gp[4 * r + 3];
        if (j >= 1) f.G[4 * r + c] = v;
      }
  }
}

// The reverse sweep: (dL/dtransforms (J,12), dL/dd_nodes (J,3) or NULL) -> dq[0..3] = dL/dlocal_rot of the lane's joint.
__device__ __forceinline__ void fk_wave_backward(int J, const FkIn& in, const FkLane& f, float (&dq)[4]) {
  const int j = threadIdx.x & 63;
  const bool on = j < J;
  float dG[12];
#pragma unroll
  for (int e = 0; e < 12; e++) dG[e] = 0.f;
  if (on) {
    const float x = in.x[0], y = in.x[1], z = in.x[2];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const float e = in.gn[r];  // posed_j = G_j [joint_j; 1]
      dG[4 * r] = in.dG[4 * r] + e * x;
      dG[4 * r + 1] = in.dG[4 * r + 1] + e * y;
      dG[4 * r + 2] = in.dG[4 * r + 2] + e * z;
      dG[4 * r + 3] = in.dG[4 * r + 3] + e;
    }
  }
  // deepest level first: the joints of level l + 1 (their dG is final) hand  [dR_G R_T^T + dt_G t_T^T | dt_G]  to their parents
  for (int l = f.maxlev - 1; l >= 0; l--) {
    float add[12];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        if (c < 3) add[4 * r + c] = dG[4 * r] * f.T[4 * c] + dG[4 * r + 1] * f.T[4 * c + 1] + dG[4 * r + 2] * f.T[4 * c + 2] + dG[4 * r + 3] * f.T[4 * c + 3];
        else add[4 * r + c] = dG[4 * r + 3];
      }
    unsigned long long km = (on && f.lev == l) ? f.kids : 0ull;
    while (__builtin_amdgcn_ballot_w64(km != 0ull) != 0ull) {
      const bool has = km != 0ull;
      const int i = has ? 63 - __builtin_clzll(km) : j;  // descending child index
#pragma unroll
      for (int e = 0; e < 12; e++) {
        const float v = fk_lane_get(add[e], i);
        if (has) dG[e] += v;
      }
      if (has) km &= ~(1ull << i);
    }
  }
  // dT_j = Rp^T dG_j (both the rotation block and the translation column); the root's is its dG
  float dT[12];
  {
    float Gp[12];
#pragma unroll
    for (int e = 0; e < 12; e++) Gp[e] = fk_lane_get(f.G[e], f.par);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 4; c++)
        dT[4 * r + c] = (j == 0) ? dG[4 * r + c] : Gp[r] * dG[c] + Gp[4 + r] * dG[4 + c] + Gp[8 + r] * dG[8 + c];
  }
  dq[0] = 0.f; dq[1] = 0.f; dq[2] = 0.f; dq[3] = 0.f;
  if (on) {
    const float cc[3] = {in.c[0], in.c[1], in.c[2]};
    float dR[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) dR[3 * a + b] = dT[4 * a + b] - dT[4 * a + 3] * cc[b];  // t = c - R c
    const float qr = in.q[0], qi = in.q[1], qj = in.q[2], qk = in.q[3];
    const float n = qr * qr + qi * qi + qj * qj + qk * qk;
    const float s = 2.0f / n;
    // A = (R - I)/s
    const float A[9] = {-(qj * qj + qk * qk), qi * qj - qk * qr, qi * qk + qj * qr,
                        qi * qj + qk * qr, -(qi * qi + qk * qk), qj * qk - qi * qr,
                        qi * qk - qj * qr, qj * qk + qi * qr, -(qi * qi + qj * qj)};
    float dotA = 0.f;
#pragma unroll
    for (int e = 0; e < 9; e++) dotA += dR[e] * A[e];
    const float gr = -qk * dR[1] + qj * dR[2] + qk * dR[3] - qi * dR[5] - qj * dR[6] + qi * dR[7];
    const float gi = qj * dR[1] + qk * dR[2] + qj * dR[3] - 2.f * qi * dR[4] - qr * dR[5] + qk * dR[6] + qr * dR[7] - 2.f * qi * dR[8];
    const float gj = -2.f * qj * dR[0] + qi * dR[1] + qr * dR[2] + qi * dR[3] + qk * dR[5] - qr * dR[6] + qk * dR[7] - 2.f * qj * dR[8];
    const float gk = -2.f * qk * dR[0] - qr * dR[1] + qi * dR[2] + qr * dR[3] - 2.f * qk * dR[4] + qj * dR[5] + qi * dR[6] + qj * dR[7];
    const float s2 = s * s;
    dq[0] = s * gr - s2 * qr * dotA;
    dq[1] = s * gi - s2 * qi * dotA;
    dq[2] = s * gj - s2 * qj * dotA;
    dq[3] = s * gk - s2 * qk * dotA;
  }
}

}  // namespace riggs
