// Per-Gaussian projection math shared by preprocess forward and backward (device only).
//
// The GEOMETRY chain (view transform, Sigma3D, cov2D, radius, tile rectangle, pixel centre)
// is written as plain left-to-right float expressions and compiled with FP contraction OFF
// (see the pragma below and the -ffp-contract=off build flag of this translation unit) so
// that depth bits, radii and tile rectangles are BIT-IDENTICAL to the CPU oracle
// (oracle/raster_ref.c, gcc -ffp-contract=off): this is what makes the tile/depth ordering
// contract testable bit-for-bit.  IEEE division and sqrt are guaranteed by
// -fhip-fp32-correctly-rounded-divide-sqrt.
#pragma once
#include "common.h"

#pragma clang fp contract(off)

namespace riggs {

#define RIGGS_NEAR_Z 0.2f

struct Cov2D {
  float tx, ty, tz;
  bool clamp_x, clamp_y;
  float M2[6];
  float a, b, c;
};

__device__ __forceinline__ void quat_to_R(const float q[4], float R[9]) {
  float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
  R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
  R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], float c6[6]) {
  float R[9], M[9];
  quat_to_R(q, R);
  float sx = mod * s[0], sy = mod * s[1], sz = mod * s[2];
#pragma unroll
  for (int i = 0; i < 3; i++) { M[3 * i] = R[3 * i] * sx; M[3 * i + 1] = R[3 * i + 1] * sy; M[3 * i + 2] = R[3 * i + 2] * sz; }
  c6[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
  c6[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
  c6[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
  c6[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
  c6[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
  c6[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

__device__ __forceinline__ void cov2d_eval(const float p[3], const float c6[6], const float* __restrict__ V, float fx,
                                           float fy, float tanx, float tany, Cov2D& o) {
  float tx = V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12];
  float ty = V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13];
  float tz = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
  float limx = 1.3f * tanx, limy = 1.3f * tany;
  float txtz = tx / tz, tytz = ty / tz;
  o.clamp_x = (txtz < -limx || txtz > limx);
  o.clamp_y = (tytz < -limy || tytz > limy);
  tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
  ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
  o.tx = tx; o.ty = ty; o.tz = tz;
  float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
  float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
  float* M2 = o.M2;
  M2[0] = J00 * V[0] + J02 * V[2];
  M2[1] = J00 * V[4] + J02 * V[6];
  M2[2] = J00 * V[8] + J02 * V[10];
  M2[3] = J11 * V[1] + J12 * V[2];
  M2[4] = J11 * V[5] + J12 * V[6];
  M2[5] = J11 * V[9] + J12 * V[10];
  float s00 = c6[0] * M2[0] + c6[1] * M2[1] + c6[2] * M2[2];
  float s10 = c6[1] * M2[0] + c6[3] * M2[1] + c6[4] * M2[2];
  float s20 = c6[2] * M2[0] + c6[4] * M2[1] + c6[5] * M2[2];
  float s01 = c6[0] * M2[3] + c6[1] * M2[4] + c6[2] * M2[5];
  float s11 = c6[1] * M2[3] + c6[3] * M2[4] + c6[4] * M2[5];
  float s21 = c6[2] * M2[3] + c6[4] * M2[4] + c6[5] * M2[5];
  o.a = (M2[0] * s00 + M2[1] * s10 + M2[2] * s20) + 0.3f;
  o.b = M2[0] * s01 + M2[1] * s11 + M2[2] * s21;
  o.c = (M2[3] * s01 + M2[4] * s11 + M2[5] * s21) + 0.3f;
}

// SH constants: /root/reference/utils/sh_utils.py:26-43
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
#define SH_C2_0 1.0925484305920792f
#define SH_C2_1 -1.0925484305920792f
#define SH_C2_2 0.31539156525252005f
#define SH_C2_3 -1.0925484305920792f
#define SH_C2_4 0.5462742152960396f
#define SH_C3_0 -0.5900435899266435f
#define SH_C3_1 2.890611442640554f
#define SH_C3_2 -0.4570457994644658f
#define SH_C3_3 0.3731763325901154f
#define SH_C3_4 -0.4570457994644658f
#define SH_C3_5 1.445305721320277f
#define SH_C3_6 -0.5900435899266435f

__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float B[16]) {
  B[0] = SH_C0;
  if (deg > 0) {
    B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      B[4] = SH_C2_0 * xy; B[5] = SH_C2_1 * yz; B[6] = SH_C2_2 * (2.f * zz - xx - yy);
      B[7] = SH_C2_3 * xz; B[8] = SH_C2_4 * (xx - yy);
      if (deg > 2) {
        B[9] = SH_C3_0 * y * (3.f * xx - yy);
        B[10] = SH_C3_1 * xy * z;
        B[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
        B[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
        B[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
        B[14] = SH_C3_5 * z * (xx - yy);
        B[15] = SH_C3_6 * x * (xx - 3.f * yy);
      }
    }
  }
}

// d(basis_k)/d(dir) contracted with a per-coefficient weight w[k] (= sum_ch sh[k][ch]*g[ch]).
__device__ __forceinline__ void sh_dir_grad(int deg, float x, float y, float z, const float w[16], float g[3]) {
  g[0] = g[1] = g[2] = 0.f;
  if (deg > 0) {
    g[1] += -SH_C1 * w[1]; g[2] += SH_C1 * w[2]; g[0] += -SH_C1 * w[3];
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      g[0] += SH_C2_0 * y * w[4]; g[1] += SH_C2_0 * x * w[4];
      g[1] += SH_C2_1 * z * w[5]; g[2] += SH_C2_1 * y * w[5];
      g[0] += SH_C2_2 * -2.f * x * w[6]; g[1] += SH_C2_2 * -2.f * y * w[6]; g[2] += SH_C2_2 * 4.f * z * w[6];
      g[0] += SH_C2_3 * z * w[7]; g[2] += SH_C2_3 * x * w[7];
      g[0] += SH_C2_4 * 2.f * x * w[8]; g[1] += SH_C2_4 * -2.f * y * w[8];
      if (deg > 2) {
        g[0] += SH_C3_0 * 6.f * xy * w[9]; g[1] += SH_C3_0 * (3.f * xx - 3.f * yy) * w[9];
        g[0] += SH_C3_1 * yz * w[10]; g[1] += SH_C3_1 * xz * w[10]; g[2] += SH_C3_1 * xy * w[10];
        g[0] += SH_C3_2 * -2.f * xy * w[11]; g[1] += SH_C3_2 * (4.f * zz - xx - 3.f * yy) * w[11]; g[2] += SH_C3_2 * 8.f * yz * w[11];
        g[0] += SH_C3_3 * -6.f * xz * w[12]; g[1] += SH_C3_3 * -6.f * yz * w[12]; g[2] += SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy) * w[12];
        g[0] += SH_C3_4 * (4.f * zz - 3.f * xx - yy) * w[13]; g[1] += SH_C3_4 * -2.f * xy * w[13]; g[2] += SH_C3_4 * 8.f * xz * w[13];
        g[0] += SH_C3_5 * 2.f * xz * w[14]; g[1] += SH_C3_5 * -2.f * yz * w[14]; g[2] += SH_C3_5 * (xx - yy) * w[14];
        g[0] += SH_C3_6 * (3.f * xx - 3.f * yy) * w[15]; g[1] += SH_C3_6 * -6.f * xy * w[15];
      }
    }
  }
}

// "Render glue" activations (gaussian_renderer/__init__.py:74-92, scene/gaussian_model.py:104-132)
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

struct GlueIn {
  float p[3];      // means3D
  float q[4];      // unit quaternion handed to the rasterizer
  float s[3];      // activated scales
  float es[3];     // exp(_scaling) (glue backward)
  float o;         // activated opacity
  float vnorm;     // |_rotation + d_rotation| (glue backward)
  float v[4];      // _rotation + d_rotation
};

}  // namespace riggs
