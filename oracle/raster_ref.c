/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C float32 CPU restatement of the tile-based differentiable 3D-Gaussian
 * rasterizer that RigGS calls through `diff_gaussian_rasterization`
 * (call sites: /root/reference/gaussian_renderer/__init__.py:14,57-72,133-141).
 *
 * PARITY UNPINNED AT THE SOURCE LEVEL: the rasterizer is a third-party git
 * submodule (github.com/ashawkey/diff-gaussian-rasterization, no commit pinned in
 * /root/reference/.gitmodules:1-6) whose source is ABSENT from /root/reference.
 * This file restates the *published* 3DGS algorithm (Kerbl et al. 2023) plus the
 * fork's extra depth/alpha outputs as specified in SURVEY.md Appendix B, and is
 * pinned by: (1) the SH / covariance golden vectors captured from the reference's
 * own Python (utils/sh_utils.py:57-112, utils/general_utils.py:137-170) in
 * tests/golden/glue_*.npz, (2) closed-form known-answer tests, (3) an independent
 * float64 dense autograd splat (tests/dense_splat.py) for forward AND gradients.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  Build: see oracle/Makefile (gcc -O2 -ffp-contract=off).
 *
 * Matrices are the reference's transposed row-vector 4x4 tensors read as
 * column-major (scene/cameras.py:61-71): view.z = m[2]x + m[6]y + m[10]z + m[14].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16
#define NEAR_Z 0.2f
#define ALPHA_MIN (1.0f / 255.0f)
#define ALPHA_MAX 0.99f
#define T_EPS 0.0001f

/* utils/sh_utils.py:26-43 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

static int g_threads = 1;
void rr_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

typedef struct {
  int N, deg, M, W, H;
  float tanfovx, tanfovy, scale_modifier;
  const float *bg, *view, *proj, *campos;
} rr_cfg;

/* ---- geometry of one Gaussian (shared by forward and backward recompute) ---- */
static void quat_to_R(const float *q, float R[9]) {
  /* unit quaternion assumed, NOT renormalised (Appendix B; same polynomial as
     utils/general_utils.py:137-157 build_rotation after its normalisation) */
  float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
  R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
  R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

static void cov3d_from_scale_rot(const float *s, float mod, const float *q, float c6[6]) {
  float R[9], M[9];
  quat_to_R(q, R);
  float sx = mod * s[0], sy = mod * s[1], sz = mod * s[2];
  for (int i = 0; i < 3; i++) { M[3 * i] = R[3 * i] * sx; M[3 * i + 1] = R[3 * i + 1] * sy; M[3 * i + 2] = R[3 * i + 2] * sz; }
  /* Sigma = M M^T */
  c6[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
  c6[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
  c6[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
  c6[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
  c6[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
  c6[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

typedef struct {
  float tx, ty, tz;       /* view-space mean, tx/ty after frustum clamp */
  int clamp_x, clamp_y;   /* 1 when the clamp was active (gradient zeroed) */
  float M2[6];            /* 2x3 = Jm * Wm */
  float J00, J02, J11, J12;
  float a, b, c;          /* cov2D incl. +0.3 */
} cov2d_t;

static void cov2d_eval(const float *p, const float *c6, const float *V, float fx, float fy, float tanx, float tany,
                       cov2d_t *o) {
  float tx = V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12];
  float ty = V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13];
  float tz = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
  float limx = 1.3f * tanx, limy = 1.3f * tany;
  float txtz = tx / tz, tytz = ty / tz;
  o->clamp_x = (txtz < -limx || txtz > limx);
  o->clamp_y = (tytz < -limy || tytz > limy);
  tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
  ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
  o->tx = tx; o->ty = ty; o->tz = tz;
  float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
  float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
  o->J00 = J00; o->J02 = J02; o->J11 = J11; o->J12 = J12;
  /* Wm[i][j] = V[j*4+i]; M2 = Jm * Wm */
  float *M2 = o->M2;
  M2[0] = J00 * V[0] + J02 * V[2];
  M2[1] = J00 * V[4] + J02 * V[6];
  M2[2] = J00 * V[8] + J02 * V[10];
  M2[3] = J11 * V[1] + J12 * V[2];
  M2[4] = J11 * V[5] + J12 * V[6];
  M2[5] = J11 * V[9] + J12 * V[10];
  /* S = Sigma * M2^T rows */
  float s00 = c6[0] * M2[0] + c6[1] * M2[1] + c6[2] * M2[2];
  float s10 = c6[1] * M2[0] + c6[3] * M2[1] + c6[4] * M2[2];
  float s20 = c6[2] * M2[0] + c6[4] * M2[1] + c6[5] * M2[2];
  float s01 = c6[0] * M2[3] + c6[1] * M2[4] + c6[2] * M2[5];
  float s11 = c6[1] * M2[3] + c6[3] * M2[4] + c6[4] * M2[5];
  float s21 = c6[2] * M2[3] + c6[4] * M2[4] + c6[5] * M2[5];
  o->a = (M2[0] * s00 + M2[1] * s10 + M2[2] * s20) + 0.3f;
  o->b = M2[0] * s01 + M2[1] * s11 + M2[2] * s21;
  o->c = (M2[3] * s01 + M2[4] * s11 + M2[5] * s21) + 0.3f;
}

static void sh_basis(int deg, float x, float y, float z, float *B) {
  B[0] = SH_C0;
  if (deg > 0) {
    B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (2.f * zz - xx - yy);
      B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
      if (deg > 2) {
        B[9] = SH_C3[0] * y * (3.f * xx - yy);
        B[10] = SH_C3[1] * xy * z;
        B[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
        B[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
        B[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
        B[14] = SH_C3[5] * z * (xx - yy);
        B[15] = SH_C3[6] * x * (xx - 3.f * yy);
      }
    }
  }
}

/* gradient of the basis w.r.t. the unit direction: dB[k][0..2] */
static void sh_basis_grad(int deg, float x, float y, float z, float dB[16][3]) {
  memset(dB, 0, sizeof(float) * 48);
  if (deg > 0) {
    dB[1][1] = -SH_C1; dB[2][2] = SH_C1; dB[3][0] = -SH_C1;
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      dB[4][0] = SH_C2[0] * y; dB[4][1] = SH_C2[0] * x;
      dB[5][1] = SH_C2[1] * z; dB[5][2] = SH_C2[1] * y;
      dB[6][0] = SH_C2[2] * -2.f * x; dB[6][1] = SH_C2[2] * -2.f * y; dB[6][2] = SH_C2[2] * 4.f * z;
      dB[7][0] = SH_C2[3] * z; dB[7][2] = SH_C2[3] * x;
      dB[8][0] = SH_C2[4] * 2.f * x; dB[8][1] = SH_C2[4] * -2.f * y;
      if (deg > 2) {
        dB[9][0] = SH_C3[0] * 6.f * xy; dB[9][1] = SH_C3[0] * (3.f * xx - 3.f * yy);
        dB[10][0] = SH_C3[1] * yz; dB[10][1] = SH_C3[1] * xz; dB[10][2] = SH_C3[1] * xy;
        dB[11][0] = SH_C3[2] * -2.f * xy; dB[11][1] = SH_C3[2] * (4.f * zz - xx - 3.f * yy); dB[11][2] = SH_C3[2] * 8.f * yz;
        dB[12][0] = SH_C3[3] * -6.f * xz; dB[12][1] = SH_C3[3] * -6.f * yz; dB[12][2] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
        dB[13][0] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); dB[13][1] = SH_C3[4] * -2.f * xy; dB[13][2] = SH_C3[4] * 8.f * xz;
        dB[14][0] = SH_C3[5] * 2.f * xz; dB[14][1] = SH_C3[5] * -2.f * yz; dB[14][2] = SH_C3[5] * (xx - yy);
        dB[15][0] = SH_C3[6] * (3.f * xx - 3.f * yy); dB[15][1] = SH_C3[6] * -6.f * xy;
      }
    }
  }
}

static inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ------------------------------------------------------------------ preprocess */
static void preprocess_one(const rr_cfg *c, int i, const float *means3D, const float *shs, const float *colors_precomp,
                           const float *opac, const float *scales, const float *rots, const float *cov3D_precomp,
                           float *depths, int *radii, float *xy, float *cov3D, float *conic_o, float *rgb,
                           uint8_t *clamped, uint32_t *tiles, int *rect) {
  const float *V = c->view, *P = c->proj;
  radii[i] = 0; tiles[i] = 0;
  rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;
  const float *p = means3D + 3 * i;
  float vz = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
  if (vz <= NEAR_Z) return;
  float hx = P[0] * p[0] + P[4] * p[1] + P[8] * p[2] + P[12];
  float hy = P[1] * p[0] + P[5] * p[1] + P[9] * p[2] + P[13];
  float hw = P[3] * p[0] + P[7] * p[1] + P[11] * p[2] + P[15];
  float pw = 1.0f / (hw + 0.0000001f);
  float ndcx = hx * pw, ndcy = hy * pw;
  float *c6 = cov3D + 6 * i;
  if (cov3D_precomp) memcpy(c6, cov3D_precomp + 6 * i, 24);
  else cov3d_from_scale_rot(scales + 3 * i, c->scale_modifier, rots + 4 * i, c6);
  float fx = c->W / (2.0f * c->tanfovx), fy = c->H / (2.0f * c->tanfovy);
  cov2d_t cv;
  cov2d_eval(p, c6, V, fx, fy, c->tanfovx, c->tanfovy, &cv);
  float det = cv.a * cv.c - cv.b * cv.b;
  if (det == 0.0f) return;
  float det_inv = 1.0f / det;
  float mid = 0.5f * (cv.a + cv.c);
  float root = sqrtf(fmaxf(0.1f, mid * mid - det));
  float lam1 = mid + root, lam2 = mid - root;
  float rad = ceilf(3.0f * sqrtf(fmaxf(lam1, lam2)));
  float px = ((ndcx + 1.0f) * c->W - 1.0f) * 0.5f;
  float py = ((ndcy + 1.0f) * c->H - 1.0f) * 0.5f;
  int gx = (c->W + TILE - 1) / TILE, gy = (c->H + TILE - 1) / TILE;
  int ir = (int)rad;
  int x0 = (int)((px - ir) / (float)TILE), y0 = (int)((py - ir) / (float)TILE);
  int x1 = (int)((px + ir + TILE - 1) / (float)TILE), y1 = (int)((py + ir + TILE - 1) / (float)TILE);
  x0 = x0 < 0 ? 0 : (x0 > gx ? gx : x0); x1 = x1 < 0 ? 0 : (x1 > gx ? gx : x1);
  y0 = y0 < 0 ? 0 : (y0 > gy ? gy : y0); y1 = y1 < 0 ? 0 : (y1 > gy ? gy : y1);
  if ((x1 - x0) * (y1 - y0) == 0) return;
  if (colors_precomp) {
    for (int ch = 0; ch < 3; ch++) { rgb[3 * i + ch] = colors_precomp[3 * i + ch]; clamped[3 * i + ch] = 0; }
  } else {
    float dx = p[0] - c->campos[0], dy = p[1] - c->campos[1], dz = p[2] - c->campos[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    dx = dx / len; dy = dy / len; dz = dz / len;
    float B[16];
    sh_basis(c->deg, dx, dy, dz, B);
    int nb = (c->deg + 1) * (c->deg + 1);
    const float *sh = shs + (size_t)i * c->M * 3;
    for (int ch = 0; ch < 3; ch++) {
      float r = 0.f;
      for (int k = 0; k < nb; k++) r += B[k] * sh[3 * k + ch];
      r += 0.5f;
      clamped[3 * i + ch] = (r < 0.f);
      rgb[3 * i + ch] = fmaxf(r, 0.f);
    }
  }
  depths[i] = vz; radii[i] = ir;
  xy[2 * i] = px; xy[2 * i + 1] = py;
  conic_o[4 * i] = cv.c * det_inv; conic_o[4 * i + 1] = -cv.b * det_inv; conic_o[4 * i + 2] = cv.a * det_inv;
  conic_o[4 * i + 3] = opac[i];
  tiles[i] = (uint32_t)((x1 - x0) * (y1 - y0));
  rect[4 * i] = x0; rect[4 * i + 1] = y0; rect[4 * i + 2] = x1; rect[4 * i + 3] = y1;
}

/* stable LSD radix sort of (key64, val32) pairs on bits [0, nbits) */
static void radix_sort_pairs(uint64_t *k, uint32_t *v, size_t n, int nbits) {
  uint64_t *k2 = (uint64_t *)malloc(n * 8 + 8);
  uint32_t *v2 = (uint32_t *)malloc(n * 4 + 4);
  uint64_t *src = k, *dst = k2; uint32_t *vs = v, *vd = v2;
  for (int sh = 0; sh < nbits; sh += 8) {
    size_t cnt[257]; memset(cnt, 0, sizeof(cnt));
    for (size_t i = 0; i < n; i++) cnt[((src[i] >> sh) & 255) + 1]++;
    for (int b = 0; b < 256; b++) cnt[b + 1] += cnt[b];
    for (size_t i = 0; i < n; i++) { size_t d = cnt[(src[i] >> sh) & 255]++; dst[d] = src[i]; vd[d] = vs[i]; }
    uint64_t *t = src; src = dst; dst = t; uint32_t *tv = vs; vs = vd; vd = tv;
  }
  if (src != k) { memcpy(k, src, n * 8); memcpy(v, vs, n * 4); }
  free(k2); free(v2);
}

/* Forward.  Returns R (number of tile instances), or -(needed) when `cap` is too small.
 * Saved state arrays are caller-allocated (sizes in comments). */
long rr_forward(int N, int deg, int M, int W, int H, const float *bg, const float *means3D, const float *shs,
                const float *colors_precomp, const float *opac, const float *scales, float scale_modifier,
                const float *rots, const float *cov3D_precomp, const float *view, const float *proj,
                const float *campos, float tanfovx, float tanfovy,
                float *out_color /*3HW*/, float *out_depth /*HW*/, float *out_alpha /*HW*/, int *radii /*N*/,
                float *depths /*N*/, float *xy /*2N*/, float *cov3D /*6N*/, float *conic_o /*4N*/, float *rgb /*3N*/,
                uint8_t *clamped /*3N*/, uint32_t *tiles_touched /*N*/, uint64_t *keys /*cap*/,
                uint32_t *point_list /*cap*/, long cap, uint32_t *ranges /*2T*/, float *final_T /*HW*/,
                uint32_t *n_contrib /*HW*/) {
  rr_cfg c = {N, deg, M, W, H, tanfovx, tanfovy, scale_modifier, bg, view, proj, campos};
  int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy;
  int *rect = (int *)malloc((size_t)N * 16 + 16);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int i = 0; i < N; i++)
    preprocess_one(&c, i, means3D, shs, colors_precomp, opac, scales, rots, cov3D_precomp, depths, radii, xy, cov3D,
                   conic_o, rgb, clamped, tiles_touched, rect);
  long R = 0;
  for (int i = 0; i < N; i++) R += tiles_touched[i];
  if (R > cap) { free(rect); return -R; }
  /* duplicateWithKeys: row-major over the rect (y outer, x inner) */
  long off = 0;
  for (int i = 0; i < N; i++) {
    if (radii[i] <= 0) continue;
    for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; y++)
      for (int x = rect[4 * i]; x < rect[4 * i + 2]; x++) {
        keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | fbits(depths[i]);
        point_list[off] = (uint32_t)i; off++;
      }
  }
  int tbits = 0; while ((1 << tbits) < T) tbits++;
  radix_sort_pairs(keys, point_list, (size_t)R, 32 + tbits + 1);
  memset(ranges, 0, (size_t)T * 8);
  for (long i = 0; i < R; i++) {
    uint32_t t = (uint32_t)(keys[i] >> 32);
    if (i == 0) ranges[2 * t] = 0;
    else { uint32_t pt = (uint32_t)(keys[i - 1] >> 32); if (pt != t) { ranges[2 * pt + 1] = (uint32_t)i; ranges[2 * t] = (uint32_t)i; } }
    if (i == R - 1) ranges[2 * t + 1] = (uint32_t)R;
  }
  /* render */
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
  for (int tile = 0; tile < T; tile++) {
    int tx0 = (tile % gx) * TILE, ty0 = (tile / gx) * TILE;
    uint32_t s = ranges[2 * tile], e = ranges[2 * tile + 1];
    for (int ly = 0; ly < TILE; ly++) for (int lx = 0; lx < TILE; lx++) {
      int pxi = tx0 + lx, pyi = ty0 + ly;
      if (pxi >= W || pyi >= H) continue;
      float pfx = (float)pxi, pfy = (float)pyi;
      float Tr = 1.0f, C0 = 0, C1 = 0, C2 = 0, D = 0, A = 0;
      uint32_t contributor = 0, last = 0;
      for (uint32_t k = s; k < e; k++) {
        contributor++;
        uint32_t id = point_list[k];
        float dx = xy[2 * id] - pfx, dy = xy[2 * id + 1] - pfy;
        const float *co = conic_o + 4 * id;
        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        float alpha = fminf(ALPHA_MAX, co[3] * expf(power));
        if (alpha < ALPHA_MIN) continue;
        float test_T = Tr * (1.0f - alpha);
        if (test_T < T_EPS) break; /* pixel done; this contributor is not counted */
        float w = alpha * Tr;
        C0 += rgb[3 * id] * w; C1 += rgb[3 * id + 1] * w; C2 += rgb[3 * id + 2] * w;
        D += depths[id] * w; A += w;
        Tr = test_T; last = contributor;
      }
      size_t pid = (size_t)pyi * W + pxi, HW = (size_t)H * W;
      final_T[pid] = Tr; n_contrib[pid] = last;
      out_color[pid] = C0 + Tr * bg[0]; out_color[HW + pid] = C1 + Tr * bg[1]; out_color[2 * HW + pid] = C2 + Tr * bg[2];
      out_depth[pid] = D; out_alpha[pid] = A;
    }
  }
  free(rect);
  return R;
}

/* The per-Gaussian sums over pixels are kept in DOUBLE: upstream adds them with float atomicAdd one pixel at a time, in
 * whatever order the hardware serves — for a splat that covers 10^5 pixels under a cotangent of changing sign the float sum
 * carries per-cent-level rounding (seen: 3-5 % on dL/dmean of a screen-filling near-camera Gaussian, tests/test_gpu_fuzz.py
 * case 20), i.e. the reference itself has no single float answer there.  A checker should sit at the exact sum. */
static inline void atomic_addd(double *p, double v) {
#pragma omp atomic
  *p += v;
}

/* Backward.  Gradient outputs are overwritten (zero-initialised here). */
void rr_backward(int N, int deg, int M, int W, int H, const float *bg, const float *means3D, const float *shs,
                 const float *colors_precomp, const float *opac, const float *scales, float scale_modifier,
                 const float *rots, const float *cov3D_precomp, const float *view, const float *proj,
                 const float *campos, float tanfovx, float tanfovy,
                 /* saved */ const int *radii, const float *depths, const float *xy, const float *cov3D,
                 const float *conic_o, const float *rgb, const uint8_t *clamped, const uint32_t *point_list, long R,
                 const uint32_t *ranges, const float *final_T, const uint32_t *n_contrib,
                 /* incoming */ const float *dL_dcolor /*3HW*/, const float *dL_ddepth_img /*HW or NULL*/,
                 const float *dL_dalpha_img /*HW or NULL*/,
                 /* out */ float *dL_dmeans3D /*3N*/, float *dL_dmeans2D /*3N*/, float *dL_dsh /*N*M*3*/,
                 float *dL_dcolors /*3N*/, float *dL_dopacity /*N*/, float *dL_dscales /*3N*/, float *dL_drots /*4N*/,
                 float *dL_dcov3D /*6N*/) {
  (void)R; (void)opac;
  int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy;
  size_t HW = (size_t)H * W;
  float *g_conic = (float *)calloc((size_t)N * 3 + 3, 4);  /* true dL/d(A,B,C) */
  float *g_depth = (float *)calloc((size_t)N + 1, 4);
  /* double accumulators of the compositing backward: [conic 3 | depth 1 | mean2D 2 | colour 3 | opacity 1] per Gaussian */
  double *acc = (double *)calloc((size_t)N * 10 + 10, 8);
  memset(dL_dmeans3D, 0, (size_t)N * 12); memset(dL_dmeans2D, 0, (size_t)N * 12);
  memset(dL_dcolors, 0, (size_t)N * 12); memset(dL_dopacity, 0, (size_t)N * 4);
  memset(dL_dscales, 0, (size_t)N * 12); memset(dL_drots, 0, (size_t)N * 16);
  memset(dL_dcov3D, 0, (size_t)N * 24);
  if (dL_dsh) memset(dL_dsh, 0, (size_t)N * M * 12);
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
  for (int tile = 0; tile < T; tile++) {
    int tx0 = (tile % gx) * TILE, ty0 = (tile / gx) * TILE;
    uint32_t s = ranges[2 * tile], e = ranges[2 * tile + 1];
    for (int ly = 0; ly < TILE; ly++) for (int lx = 0; lx < TILE; lx++) {
      int pxi = tx0 + lx, pyi = ty0 + ly;
      if (pxi >= W || pyi >= H) continue;
      size_t pid = (size_t)pyi * W + pxi;
      float pfx = (float)pxi, pfy = (float)pyi;
      float T_final = final_T[pid], Tr = T_final;
      uint32_t last = n_contrib[pid];
      float gC[3] = {dL_dcolor[pid], dL_dcolor[HW + pid], dL_dcolor[2 * HW + pid]};
      float gD = dL_ddepth_img ? dL_ddepth_img[pid] : 0.f;
      float gA = dL_dalpha_img ? dL_dalpha_img[pid] : 0.f;
      float accum[3] = {0, 0, 0}, last_color[3] = {0, 0, 0};
      float accum_d = 0, last_d = 0, accum_a = 0, last_alpha = 0;
      float bg_dot = bg[0] * gC[0] + bg[1] * gC[1] + bg[2] * gC[2];
      for (uint32_t pos = last; pos-- > 0;) { /* back-to-front over contributors [0, last) */
        uint32_t id = point_list[s + pos];
        float dx = xy[2 * id] - pfx, dy = xy[2 * id + 1] - pfy;
        const float *co = conic_o + 4 * id;
        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        float G = expf(power);
        float alpha = fminf(ALPHA_MAX, co[3] * G);
        if (alpha < ALPHA_MIN) continue;
        Tr = Tr / (1.0f - alpha);
        float w = alpha * Tr;
        float dL_dalpha = 0.f;
        for (int ch = 0; ch < 3; ch++) {
          float cc = rgb[3 * id + ch];
          accum[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum[ch];
          last_color[ch] = cc;
          dL_dalpha += (cc - accum[ch]) * gC[ch];
          atomic_addd(&acc[10 * (size_t)id + 6 + ch], (double)(w * gC[ch]));
        }
        float cd = depths[id];
        accum_d = last_alpha * last_d + (1.f - last_alpha) * accum_d; last_d = cd;
        dL_dalpha += (cd - accum_d) * gD;
        atomic_addd(&acc[10 * (size_t)id + 3], (double)(w * gD));
        accum_a = last_alpha * 1.0f + (1.f - last_alpha) * accum_a;
        dL_dalpha += (1.0f - accum_a) * gA;
        dL_dalpha *= Tr;
        last_alpha = alpha;
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
        float dL_dG = co[3] * dL_dalpha;
        float gdx = G * dx, gdy = G * dy;
        float dG_ddelx = -gdx * co[0] - gdy * co[1];
        float dG_ddely = -gdy * co[2] - gdx * co[1];
        atomic_addd(&acc[10 * (size_t)id + 4], (double)(dL_dG * dG_ddelx * ddelx_dx));
        atomic_addd(&acc[10 * (size_t)id + 5], (double)(dL_dG * dG_ddely * ddely_dy));
        atomic_addd(&acc[10 * (size_t)id + 0], (double)(-0.5f * gdx * dx * dL_dG));
        atomic_addd(&acc[10 * (size_t)id + 1], (double)(-gdx * dy * dL_dG));
        atomic_addd(&acc[10 * (size_t)id + 2], (double)(-0.5f * gdy * dy * dL_dG));
        atomic_addd(&acc[10 * (size_t)id + 9], (double)(G * dL_dalpha));
      }
      (void)e;
    }
  }

  for (int i = 0; i < N; i++) {  /* the sums, rounded once */
    const double *a = acc + 10 * (size_t)i;
    g_conic[3 * i] = (float)a[0]; g_conic[3 * i + 1] = (float)a[1]; g_conic[3 * i + 2] = (float)a[2];
    g_depth[i] = (float)a[3];
    dL_dmeans2D[3 * i] = (float)a[4]; dL_dmeans2D[3 * i + 1] = (float)a[5];
    dL_dcolors[3 * i] = (float)a[6]; dL_dcolors[3 * i + 1] = (float)a[7]; dL_dcolors[3 * i + 2] = (float)a[8];
    dL_dopacity[i] = (float)a[9];
  }
  free(acc);
  /* per-Gaussian backward */
  const float *V = view, *P = proj;
  float fx = W / (2.0f * tanfovx), fy = H / (2.0f * tanfovy);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int i = 0; i < N; i++) {
    if (radii[i] <= 0) continue;
    const float *p = means3D + 3 * i;
    const float *c6 = cov3D + 6 * i;
    cov2d_t cv;
    cov2d_eval(p, c6, V, fx, fy, tanfovx, tanfovy, &cv);
    float a = cv.a, b = cv.b, c = cv.c;
    float det = a * c - b * b;
    float d2inv = 1.0f / (det * det + 0.0000001f);
    float gA = g_conic[3 * i], gB = g_conic[3 * i + 1], gC = g_conic[3 * i + 2];
    float dL_da = d2inv * (-c * c * gA + b * c * gB - b * b * gC);
    float dL_db = d2inv * (2.f * b * c * gA - (det + 2.f * b * b) * gB + 2.f * a * b * gC);
    float dL_dc = d2inv * (-b * b * gA + a * b * gB - a * a * gC);
    /* D = [[dL_da, dL_db/2],[dL_db/2, dL_dc]] ; GS = M2^T D M2 (3x3 symmetric) */
    const float *M2 = cv.M2;
    float hb = 0.5f * dL_db;
    float DM[6]; /* D*M2 : 2x3 */
    for (int j = 0; j < 3; j++) { DM[j] = dL_da * M2[j] + hb * M2[3 + j]; DM[3 + j] = hb * M2[j] + dL_dc * M2[3 + j]; }
    float GS[9];
    for (int r = 0; r < 3; r++) for (int j = 0; j < 3; j++) GS[3 * r + j] = M2[r] * DM[j] + M2[3 + r] * DM[3 + j];
    /* dL/dM2 = 2 * D * M2 * Sigma */
    float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    float dM2[6];
    for (int r = 0; r < 2; r++) for (int j = 0; j < 3; j++)
      dM2[3 * r + j] = 2.f * (DM[3 * r] * S[j] + DM[3 * r + 1] * S[3 + j] + DM[3 * r + 2] * S[6 + j]);
    /* dL/dJm = dM2 * Wm^T ; Wm[i][j] = V[j*4+i] */
    float dJ00 = dM2[0] * V[0] + dM2[1] * V[4] + dM2[2] * V[8];
    float dJ02 = dM2[0] * V[2] + dM2[1] * V[6] + dM2[2] * V[10];
    float dJ11 = dM2[3] * V[1] + dM2[4] * V[5] + dM2[5] * V[9];
    float dJ12 = dM2[3] * V[2] + dM2[4] * V[6] + dM2[5] * V[10];
    float tz = 1.f / cv.tz, tz2 = tz * tz, tz3 = tz2 * tz;
    float dtx = (cv.clamp_x ? 0.f : 1.f) * (-fx * tz2 * dJ02);
    float dty = (cv.clamp_y ? 0.f : 1.f) * (-fy * tz2 * dJ12);
    float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * cv.tx) * tz3 * dJ02 + (2.f * fy * cv.ty) * tz3 * dJ12;
    float gm[3];
    gm[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
    gm[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
    gm[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;
    /* 2D mean (NDC-scaled gradient) through the perspective divide */
    float hx = P[0] * p[0] + P[4] * p[1] + P[8] * p[2] + P[12];
    float hy = P[1] * p[0] + P[5] * p[1] + P[9] * p[2] + P[13];
    float hw = P[3] * p[0] + P[7] * p[1] + P[11] * p[2] + P[15];
    float mw = 1.0f / (hw + 0.0000001f);
    float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
    float g2x = dL_dmeans2D[3 * i], g2y = dL_dmeans2D[3 * i + 1];
    gm[0] += (P[0] * mw - P[3] * mul1) * g2x + (P[1] * mw - P[3] * mul2) * g2y;
    gm[1] += (P[4] * mw - P[7] * mul1) * g2x + (P[5] * mw - P[7] * mul2) * g2y;
    gm[2] += (P[8] * mw - P[11] * mul1) * g2x + (P[9] * mw - P[11] * mul2) * g2y;
    /* depth = view z */
    float gd = g_depth[i];
    gm[0] += V[2] * gd; gm[1] += V[6] * gd; gm[2] += V[10] * gd;
    /* colour */
    if (!colors_precomp) {
      float dx = p[0] - campos[0], dy = p[1] - campos[1], dz = p[2] - campos[2];
      float len = sqrtf(dx * dx + dy * dy + dz * dz);
      float ux = dx / len, uy = dy / len, uz = dz / len;
      float B[16], dB[16][3];
      sh_basis(deg, ux, uy, uz, B);
      sh_basis_grad(deg, ux, uy, uz, dB);
      int nb = (deg + 1) * (deg + 1);
      const float *sh = shs + (size_t)i * M * 3;
      float *gsh = dL_dsh + (size_t)i * M * 3;
      float gdir[3] = {0, 0, 0};
      for (int ch = 0; ch < 3; ch++) {
        float gc = clamped[3 * i + ch] ? 0.f : dL_dcolors[3 * i + ch];
        for (int k = 0; k < nb; k++) {
          gsh[3 * k + ch] = B[k] * gc;
          gdir[0] += dB[k][0] * sh[3 * k + ch] * gc;
          gdir[1] += dB[k][1] * sh[3 * k + ch] * gc;
          gdir[2] += dB[k][2] * sh[3 * k + ch] * gc;
        }
      }
      float dot = ux * gdir[0] + uy * gdir[1] + uz * gdir[2];
      gm[0] += (gdir[0] - ux * dot) / len; gm[1] += (gdir[1] - uy * dot) / len; gm[2] += (gdir[2] - uz * dot) / len;
      dL_dcolors[3 * i] = dL_dcolors[3 * i + 1] = dL_dcolors[3 * i + 2] = 0.f; /* not an input in SH mode */
    }
    dL_dmeans3D[3 * i] = gm[0]; dL_dmeans3D[3 * i + 1] = gm[1]; dL_dmeans3D[3 * i + 2] = gm[2];
    /* Sigma3D -> scale / rotation (or precomputed covariance) */
    if (cov3D_precomp) {
      dL_dcov3D[6 * i] = GS[0]; dL_dcov3D[6 * i + 1] = 2.f * GS[1]; dL_dcov3D[6 * i + 2] = 2.f * GS[2];
      dL_dcov3D[6 * i + 3] = GS[4]; dL_dcov3D[6 * i + 4] = 2.f * GS[5]; dL_dcov3D[6 * i + 5] = GS[8];
    } else {
      float Rm[9];
      const float *q = rots + 4 * i; const float *sc = scales + 3 * i;
      quat_to_R(q, Rm);
      float s3[3] = {scale_modifier * sc[0], scale_modifier * sc[1], scale_modifier * sc[2]};
      float Mm[9], dMm[9];
      for (int r = 0; r < 3; r++) for (int j = 0; j < 3; j++) Mm[3 * r + j] = Rm[3 * r + j] * s3[j];
      for (int r = 0; r < 3; r++) for (int j = 0; j < 3; j++)
        dMm[3 * r + j] = 2.f * (GS[3 * r] * Mm[j] + GS[3 * r + 1] * Mm[3 + j] + GS[3 * r + 2] * Mm[6 + j]);
      float dR[9];
      for (int j = 0; j < 3; j++) {
        float ds = Rm[j] * dMm[j] + Rm[3 + j] * dMm[3 + j] + Rm[6 + j] * dMm[6 + j];
        dL_dscales[3 * i + j] = scale_modifier * ds;
        for (int r = 0; r < 3; r++) dR[3 * r + j] = s3[j] * dMm[3 * r + j];
      }
      float r = q[0], x = q[1], y = q[2], z = q[3];
      dL_drots[4 * i] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
      dL_drots[4 * i + 1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
      dL_drots[4 * i + 2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
      dL_drots[4 * i + 3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
    }
  }
  free(g_conic); free(g_depth);
}

/* Exact mean squared distance to the 3 nearest neighbours (simple_knn.distCUDA2,
 * call site /root/reference/scene/gaussian_model.py:170).  Brute force, double. */
void rr_dist2_knn3(int P, const float *pts, float *out) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int i = 0; i < P; i++) {
    float best[3] = {INFINITY, INFINITY, INFINITY};
    for (int j = 0; j < P; j++) {
      if (j == i) continue;
      float dx = pts[3 * j] - pts[3 * i], dy = pts[3 * j + 1] - pts[3 * i + 1], dz = pts[3 * j + 2] - pts[3 * i + 2];
      float d = dx * dx + dy * dy + dz * dz;
      if (d < best[2]) {
        if (d < best[1]) { best[2] = best[1]; if (d < best[0]) { best[1] = best[0]; best[0] = d; } else best[1] = d; }
        else best[2] = d;
      }
    }
    int k = P - 1 < 3 ? P - 1 : 3;
    float s = 0.f;
    for (int t = 0; t < k; t++) s += best[t];
    out[i] = k > 0 ? s / 3.0f : 0.f;
  }
}
