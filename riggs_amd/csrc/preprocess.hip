// Per-Gaussian stage of the rasterizer (SURVEY.md §8 A7+A8, A11): projection, covariance,
// SH colour, tile rectangle — forward and backward — with the optional fused "render glue".
//
// HBM-bound streaming kernels: one thread per Gaussian, 256-thread workgroups (4 waves),
// every output written once as 16-byte records where the layout is ours (xyd / conic_o / rgb
// float4 arrays) so that the compositing kernels gather whole 16-B words.
//
// This translation unit is compiled with FP contraction OFF (gauss_math.h) — geometry is
// bit-identical to oracle/raster_ref.c.
#include "color_job.h"
#include "raster_internal.h"

namespace riggs {

// A Gaussian's inputs in two steps: every global load first (no arithmetic, so that they are all in flight together), the
// arithmetic after.
struct RawIn {
  float p[3], dx[3], o, sc[3], dsc[3];
  float4 r, dr;
};
__device__ __forceinline__ void load_raw(const PreArgs& a, int i, RawIn& w, bool need_sr) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  w.p[0] = a.means3D[3 * i]; w.p[1] = a.means3D[3 * i + 1]; w.p[2] = a.means3D[3 * i + 2];
  w.dx[0] = w.dx[1] = w.dx[2] = 0.f; w.sc[0] = w.sc[1] = w.sc[2] = 0.f; w.dsc[0] = w.dsc[1] = w.dsc[2] = 0.f;
  w.r = z; w.dr = z;
  w.o = a.opac[i];
  if (a.glue && a.d_xyz) { w.dx[0] = a.d_xyz[3 * i]; w.dx[1] = a.d_xyz[3 * i + 1]; w.dx[2] = a.d_xyz[3 * i + 2]; }
  if (need_sr) {
    if (a.glue && a.isotropic) w.sc[0] = a.scales[i];
    else { w.sc[0] = a.scales[3 * i]; w.sc[1] = a.scales[3 * i + 1]; w.sc[2] = a.scales[3 * i + 2]; }
    if (a.glue && a.d_scaling) { w.dsc[0] = a.d_scaling[3 * i]; w.dsc[1] = a.d_scaling[3 * i + 1]; w.dsc[2] = a.d_scaling[3 * i + 2]; }
    w.r = reinterpret_cast<const float4*>(a.rots)[i];
    if (a.glue && a.d_rot) w.dr = reinterpret_cast<const float4*>(a.d_rot)[i];
  }
}
__device__ __forceinline__ void finish_inputs(const PreArgs& a, const RawIn& w, GlueIn& g, bool need_sr) {
  g.p[0] = w.p[0]; g.p[1] = w.p[1]; g.p[2] = w.p[2];
  if (a.glue) {
    if (a.d_xyz) { g.p[0] = g.p[0] + w.dx[0]; g.p[1] = g.p[1] + w.dx[1]; g.p[2] = g.p[2] + w.dx[2]; }
    g.o = sigmoidf_(w.o);
    if (need_sr) {
      if (a.isotropic) { float s = expf(w.sc[0]); g.es[0] = g.es[1] = g.es[2] = s; }
      else { g.es[0] = expf(w.sc[0]); g.es[1] = expf(w.sc[1]); g.es[2] = expf(w.sc[2]); }
      g.s[0] = g.es[0]; g.s[1] = g.es[1]; g.s[2] = g.es[2];
      if (a.d_scaling) { g.s[0] = g.s[0] + w.dsc[0]; g.s[1] = g.s[1] + w.dsc[1]; g.s[2] = g.s[2] + w.dsc[2]; }
      g.v[0] = w.r.x; g.v[1] = w.r.y; g.v[2] = w.r.z; g.v[3] = w.r.w;
      if (a.d_rot) { g.v[0] = g.v[0] + w.dr.x; g.v[1] = g.v[1] + w.dr.y; g.v[2] = g.v[2] + w.dr.z; g.v[3] = g.v[3] + w.dr.w; }
      float n = sqrtf(g.v[0] * g.v[0] + g.v[1] * g.v[1] + g.v[2] * g.v[2] + g.v[3] * g.v[3]);
      g.vnorm = fmaxf(n, 1e-12f);
      g.q[0] = g.v[0] / g.vnorm; g.q[1] = g.v[1] / g.vnorm; g.q[2] = g.v[2] / g.vnorm; g.q[3] = g.v[3] / g.vnorm;
    }
  } else {
    g.o = w.o;
    if (need_sr) {
      g.s[0] = w.sc[0]; g.s[1] = w.sc[1]; g.s[2] = w.sc[2];
      g.q[0] = w.r.x; g.q[1] = w.r.y; g.q[2] = w.r.z; g.q[3] = w.r.w;
      g.vnorm = 1.f;
    }
  }
}

// ---- SH staging (sh_lds_stride, sh_stage_in, sh_stage_dma: color_job.h) ----------------------------
// The same copy in two steps — every global load of the thread first (up to SH_IT independent 16-byte loads in flight),
// the LDS writes after — so that a workgroup's staging costs ONE memory round trip instead of one per 4 KB slice: written
// as a single loop the compiler waits for each load before its LDS write.  (Reading each Gaussian's 180-byte row straight
// into registers with dword-aligned 16-byte loads, which would free the LDS and let every workgroup be resident at once,
// is 2x SLOWER: 64 lanes x 64 different cache lines per load instruction.)
#define SH_IT 12  // 256 Gaussians x <= 48 floats / (256 threads x 4 floats)
__device__ __forceinline__ void sh_stage_load(const float* __restrict__ src, int total, float4 (&v)[SH_IT]) {
#pragma unroll
  for (int it = 0; it < SH_IT; it++) {
    const int e = threadIdx.x * 4 + it * 1024;
    v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e + 3 < total) v[it] = *reinterpret_cast<const float4*>(src + e);
  }
}
__device__ __forceinline__ void sh_stage_store(const float* __restrict__ src, int per, int total, const float4 (&v)[SH_IT], float* lds) {
  const int stride = sh_lds_stride(per);
  const int q1024 = 1024 / per, r1024 = 1024 % per;
  int e = threadIdx.x * 4;
  int g = e / per, k = e % per;
#pragma unroll
  for (int it = 0; it < SH_IT; it++, e += 1024) {
    if (e + 3 < total) {
      const float w[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        int kk = k + q, gg = g;
        if (kk >= per) { kk -= per; gg++; }
        lds[gg * stride + kk] = w[q];
      }
    }
    k += r1024; g += q1024;
    if (k >= per) { k -= per; g++; }
  }
  // ragged end of the last workgroup (total not a multiple of 4): at most three elements
  if (threadIdx.x < (total & 3)) {
    const int t = (total & ~3) + threadIdx.x;
    lds[(t / per) * stride + (t % per)] = src[t];
  }
}

__device__ __forceinline__ void sh_stage_out(float* __restrict__ dst, int per, int count, const float* lds) {
  const int total = count * per;
  const int stride = sh_lds_stride(per);
  const bool aligned = ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
  const int q1024 = 1024 / per, r1024 = 1024 % per;
  int e = threadIdx.x * 4;
  int g = e / per, k = e % per;
  for (; e < total; e += 1024) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int kk = k + q, gg = g;
      if (kk >= per) { kk -= per; gg++; }
      v[q] = (e + q < total) ? lds[gg * stride + kk] : 0.f;
    }
    if (aligned && e + 3 < total) *reinterpret_cast<float4*>(dst + e) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
      for (int q = 0; q < 4; q++) if (e + q < total) dst[e + q] = v[q];
    }
    k += r1024; g += q1024;
    if (k >= per) { k -= per; g++; }
  }
}

__global__ __launch_bounds__(256) void preprocess_fwd_kernel(PreArgs a) {
  extern __shared__ float s_sh[];
  const int i = blockIdx.x * 256 + threadIdx.x;
  // SH records of this workgroup -> LDS (uniform; before any per-Gaussian exit)
  // (a.defer_color: the colours are evaluated by extra workgroups of the tile sort's scatter launch — color_job.h — and this
  // kernel neither stages nor reads a coefficient; either way it leaves the job's record for that launch: N = 0 = nothing to do)
  if (a.job_rec && blockIdx.x == 0 && threadIdx.x == 0) {
    ColorJob r;
    r.N = a.defer_color ? a.N : 0; r.deg = a.deg; r.M = a.M; r.pad_ = 0;
    r.shs = a.shs; r.shs_rest = a.shs_rest; r.means3D = a.means3D; r.d_xyz = (a.glue && a.d_xyz) ? a.d_xyz : nullptr;
    r.campos = a.campos; r.radii = a.radii; r.rgb = a.rgb; r.clamped = a.clamped;
    *a.job_rec = r;
  }
  const bool sh_mode = (a.colors_precomp == nullptr) && !a.defer_color;
  const int sh_per = a.shs_rest ? (a.M - 1) * 3 : a.M * 3;  // floats per Gaussian in the staged array
  // every global load of the thread is issued before anything waits: its slices of the workgroup's SH run and its own
  // Gaussian's parameters (the compiler otherwise serialises them: one round trip per load)
  const bool need_sr = (a.cov3D_precomp == nullptr);
  const int sh_first = blockIdx.x * 256, sh_total = min(256, a.N - sh_first) * sh_per;
  const float* sh_src = sh_mode ? (a.shs_rest ? a.shs_rest : a.shs) + (size_t)sh_first * sh_per : nullptr;
  const bool sh_fast = sh_mode && sh_per > 0 && sh_per <= 48 && ((reinterpret_cast<uintptr_t>(sh_src) & 15) == 0);
  const bool sh_dma = sh_fast && (sh_per & 1);  // (odd record length: the unpadded run is the conflict-free LDS image)
  float4 shq[SH_IT];
  RawIn raw;
  if (sh_dma) sh_stage_dma(sh_src, sh_total, s_sh);
  else if (sh_fast) sh_stage_load(sh_src, sh_total, shq);
  if (i < a.N) load_raw(a, i, raw, need_sr);
  float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < a.N && a.cov3D_precomp) {
#pragma unroll
    for (int k = 0; k < 6; k++) c6[k] = a.cov3D_precomp[6 * i + k];
  }
  float cpre[3] = {0.f, 0.f, 0.f}, dc0[3] = {0.f, 0.f, 0.f};
  if (i < a.N && a.colors_precomp) { cpre[0] = a.colors_precomp[3 * i]; cpre[1] = a.colors_precomp[3 * i + 1]; cpre[2] = a.colors_precomp[3 * i + 2]; }
  if (i < a.N && sh_mode && a.shs_rest) { dc0[0] = a.shs[3 * i]; dc0[1] = a.shs[3 * i + 1]; dc0[2] = a.shs[3 * i + 2]; }
  __builtin_amdgcn_sched_barrier(0);
  if (sh_mode && sh_per > 0) {
    if (sh_dma) {}
    else if (sh_fast) sh_stage_store(sh_src, sh_per, sh_total, shq, s_sh);
    else sh_stage_in(sh_src, sh_per, min(256, a.N - sh_first), s_sh);
    __syncthreads();
  }
  uint32_t my_tiles = 0u, my_top = 0xFFFFFFFFu;  // (top byte of the depth key of a visible Gaussian, else none)
  ushort4 my_rect = make_ushort4(0, 0, 0, 0);   // (stays empty for a culled Gaussian: the tile sorts read the rectangles alone)
  do {
  if (i >= a.N) break;
  const float* __restrict__ V = a.view;
  const float* __restrict__ P = a.proj;
  a.radii[i] = 0;
  a.tiles[i] = 0;
  a.depth_key[i] = 0xFFFFFFFFu;  // culled Gaussians sort last
  GlueIn g;
  finish_inputs(a, raw, g, need_sr);
  const float* p = g.p;
  float vz = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
  if (vz <= RIGGS_NEAR_Z) break;
  float hx = P[0] * p[0] + P[4] * p[1] + P[8] * p[2] + P[12];
  float hy = P[1] * p[0] + P[5] * p[1] + P[9] * p[2] + P[13];
  float hw = P[3] * p[0] + P[7] * p[1] + P[11] * p[2] + P[15];
  float pw = 1.0f / (hw + 0.0000001f);
  float ndcx = hx * pw, ndcy = hy * pw;
  if (!a.cov3D_precomp) cov3d_from_scale_rot(g.s, a.mod, g.q, c6);
  float fx = a.W / (2.0f * a.tanx), fy = a.H / (2.0f * a.tany);
  Cov2D cv;
  cov2d_eval(p, c6, V, fx, fy, a.tanx, a.tany, cv);
  float det = cv.a * cv.c - cv.b * cv.b;
  if (det == 0.0f) break;
  float det_inv = 1.0f / det;
  float mid = 0.5f * (cv.a + cv.c);
  float root = sqrtf(fmaxf(0.1f, mid * mid - det));
  float lam1 = mid + root, lam2 = mid - root;
  float rad = ceilf(3.0f * sqrtf(fmaxf(lam1, lam2)));
  float px = ((ndcx + 1.0f) * a.W - 1.0f) * 0.5f;
  float py = ((ndcy + 1.0f) * a.H - 1.0f) * 0.5f;
  int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE, gy = (a.H + RIGGS_TILE - 1) / RIGGS_TILE;
  int ir = (int)rad;
  int x0 = (int)((px - ir) / (float)RIGGS_TILE), y0 = (int)((py - ir) / (float)RIGGS_TILE);
  int x1 = (int)((px + ir + RIGGS_TILE - 1) / (float)RIGGS_TILE), y1 = (int)((py + ir + RIGGS_TILE - 1) / (float)RIGGS_TILE);
  x0 = min(gx, max(0, x0)); x1 = min(gx, max(0, x1));
  y0 = min(gy, max(0, y0)); y1 = min(gy, max(0, y1));
  if ((x1 - x0) * (y1 - y0) == 0) break;

  float rgbv[3] = {0.f, 0.f, 0.f};
  uint8_t cl = 0;
  if (a.colors_precomp) {
    rgbv[0] = cpre[0]; rgbv[1] = cpre[1]; rgbv[2] = cpre[2];
  } else if (!a.defer_color) {
    cl = sh_color(a.deg, p, a.campos, a.shs_rest != nullptr, dc0, s_sh + threadIdx.x * sh_lds_stride(sh_per), rgbv);
  }
  // axis-aligned half extents of the region where alpha = o * exp(power) can reach 1/255 (the compositing
  // kernels cull instances against pixel blocks with it): q(d) <= tau = 2 ln(255 o), x-extent sqrt(tau * Sxx).
  // Conservative: tau carries a +0.05 margin (2.5 % in alpha) — far above any rounding of cov / conic / exp.
  float cull_hx = -1e30f, cull_hy = -1e30f;
  {
    const float tau = 2.0f * logf(255.0f * g.o) + 0.05f;
    if (tau > 0.0f) { cull_hx = sqrtf(tau * cv.a) + 0.01f; cull_hy = sqrtf(tau * cv.c) + 0.01f; }
    if (!(cull_hx == cull_hx) || !(cull_hy == cull_hy)) { cull_hx = 1e30f; cull_hy = 1e30f; }  // NaN: never cull
  }
  if (a.tight) {
    // cfg.tight_lists: the tile rectangle is cut down to the tiles whose PIXEL CENTRES (integer coordinates 16 t .. 16 t + 15)
    // the alpha >= 1/255 box [p - h, p + h] reaches — the extents above carry their margins, so no tile with a reachable pixel
    // is dropped.  A Gaussian that reaches none (opacity below 1/255, or a box that falls between tiles at the image border)
    // keeps its radius (it is "visible" as upstream defines it) and gets no instance.
    if (cull_hx < 0.0f) { x1 = x0; y1 = y0; }
    else {
      x0 = max(x0, (int)floorf((px - cull_hx) * (1.0f / RIGGS_TILE))); x1 = min(x1, (int)floorf((px + cull_hx) * (1.0f / RIGGS_TILE)) + 1);
      y0 = max(y0, (int)floorf((py - cull_hy) * (1.0f / RIGGS_TILE))); y1 = min(y1, (int)floorf((py + cull_hy) * (1.0f / RIGGS_TILE)) + 1);
      if (x1 <= x0 || y1 <= y0) { x1 = x0; y1 = y0; }
    }
  }
  a.radii[i] = ir;
  a.xyd[i] = make_float4(px, py, vz, cull_hx);
  a.conic_o[i] = make_float4(cv.c * det_inv, -cv.b * det_inv, cv.a * det_inv, g.o);
  a.rgb[i] = make_float4(rgbv[0], rgbv[1], rgbv[2], cull_hy);
#pragma unroll
  for (int k = 0; k < 6; k++) a.cov3D[6 * i + k] = c6[k];
  if (!a.defer_color) a.clamped[i] = cl;
  a.tiles[i] = (uint32_t)((x1 - x0) * (y1 - y0));
  my_rect = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1, (unsigned short)y1);
  a.depth_key[i] = __float_as_uint(vz);
  my_top = __float_as_uint(vz) >> 24;
  my_tiles = (uint32_t)((x1 - x0) * (y1 - y0));
  } while (0);
  if (i < a.N) a.rect[i] = my_rect;
  // instance count R = sum of tiles_touched: wave reduction -> per-workgroup partial (summed by the first kernel of
  // the depth sort; thousands of same-address atomics would serialise in L2)
  // ... and the range of the top bytes of the visible depth keys (the depth sort skips its third pass when they all agree)
  __shared__ uint32_t s_tiles[4], s_lo[4], s_hi[4];
  uint32_t lo = (my_top == 0xFFFFFFFFu) ? 0xFFu : my_top, hi = (my_top == 0xFFFFFFFFu) ? 0u : my_top;
  for (int o = 32; o > 0; o >>= 1) {
    my_tiles += (uint32_t)__shfl_xor((int)my_tiles, o);
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
  }
  if ((threadIdx.x & 63) == 0) { s_tiles[threadIdx.x >> 6] = my_tiles; s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.block_tiles[blockIdx.x] = s_tiles[0] + s_tiles[1] + s_tiles[2] + s_tiles[3];
    a.block_tiles[gridDim.x + blockIdx.x] = (min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])) << 8) | max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
  }
}

// ---------------------------------------------------------------------------- backward
// One Gaussian's backward arithmetic (cov2D / Sigma3D / SH / projection + the glue's chain rule), shared by the two launch forms
// of preprocess_bwd below.
// the exact-zero gradient of a Gaussian whose accumulator record is all zeros: every per-Gaussian output except its dL/dsh row
__device__ __forceinline__ void bwd_zero_row(const PreBwdArgs& b, const PreArgs& a, const int i0, const bool sh_mode) {
    b.dL_dmeans3D[3 * i0] = 0.f; b.dL_dmeans3D[3 * i0 + 1] = 0.f; b.dL_dmeans3D[3 * i0 + 2] = 0.f;
    b.dL_dmeans2D[3 * i0] = 0.f; b.dL_dmeans2D[3 * i0 + 1] = 0.f; b.dL_dmeans2D[3 * i0 + 2] = 0.f;
    if (b.dL_dcolors) { b.dL_dcolors[3 * i0] = 0.f; b.dL_dcolors[3 * i0 + 1] = 0.f; b.dL_dcolors[3 * i0 + 2] = 0.f; }
    b.dL_dopac[i0] = 0.f;
    if (b.dL_dscales) {
      if (a.glue && a.isotropic) b.dL_dscales[i0] = 0.f;
      else { b.dL_dscales[3 * i0] = 0.f; b.dL_dscales[3 * i0 + 1] = 0.f; b.dL_dscales[3 * i0 + 2] = 0.f; }
    }
    if (a.glue && b.dL_dscales && b.dL_dd_scaling) { b.dL_dd_scaling[3 * i0] = 0.f; b.dL_dd_scaling[3 * i0 + 1] = 0.f; b.dL_dd_scaling[3 * i0 + 2] = 0.f; }
    if (b.dL_drots) reinterpret_cast<float4*>(b.dL_drots)[i0] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b.dL_dcov3D) {
#pragma unroll
      for (int k = 0; k < 6; k++) b.dL_dcov3D[6 * i0 + k] = 0.f;
    }
    if (sh_mode && a.shs_rest) { b.dL_dsh[3 * i0] = 0.f; b.dL_dsh[3 * i0 + 1] = 0.f; b.dL_dsh[3 * i0 + 2] = 0.f; }
}

// (q0, q1, q2: the Gaussian's accumulator record of the compositing backward; c6: its Sigma3D; cl: clamp bits; shv: its coefficients)
__device__ __forceinline__ void bwd_math(const PreArgs& a, const bool sh_mode, const bool need_sr, const RawIn& raw, const float4 q0,
                                         const float4 q1, const float4 q2, const float (&c6)[6], const uint8_t cl,
                                         const float (&shv)[48], GlueIn& g, float (&gm)[3], float& g2x, float& g2y, float& g_op,
                                         float (&gcol)[3], float (&gs)[3], float (&gq)[4], float (&gcov)[6], float (&Bk)[16],
                                         float (&gcs)[3]) {
  const float* __restrict__ V = a.view;
  const float* __restrict__ P = a.proj;
  {
    finish_inputs(a, raw, g, need_sr);
    const float* p = g.p;
    g2x = q0.x; g2y = q0.y;
    const float gA = q0.z, gB = q0.w, gC = q1.x;
    g_op = q1.y;
    gcol[0] = q1.z; gcol[1] = q1.w; gcol[2] = q2.x;
    const float gd = q2.y;
    float fx = a.W / (2.0f * a.tanx), fy = a.H / (2.0f * a.tany);
    Cov2D cv;
    cov2d_eval(p, c6, V, fx, fy, a.tanx, a.tany, cv);
    const float ca = cv.a, cb = cv.b, cc = cv.c;
    const float det = ca * cc - cb * cb;
    const float d2inv = 1.0f / (det * det + 0.0000001f);
    const float dL_da = d2inv * (-cc * cc * gA + cb * cc * gB - cb * cb * gC);
    const float dL_db = d2inv * (2.f * cb * cc * gA - (det + 2.f * cb * cb) * gB + 2.f * ca * cb * gC);
    const float dL_dc = d2inv * (-cb * cb * gA + ca * cb * gB - ca * ca * gC);
    const float* M2 = cv.M2;
    const float hb = 0.5f * dL_db;
    float DM[6];
#pragma unroll
    for (int j = 0; j < 3; j++) { DM[j] = dL_da * M2[j] + hb * M2[3 + j]; DM[3 + j] = hb * M2[j] + dL_dc * M2[3 + j]; }
    float GS[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int j = 0; j < 3; j++) GS[3 * r + j] = M2[r] * DM[j] + M2[3 + r] * DM[3 + j];
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    float dM2[6];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        dM2[3 * r + j] = 2.f * (DM[3 * r] * S[j] + DM[3 * r + 1] * S[3 + j] + DM[3 * r + 2] * S[6 + j]);
    const float dJ00 = dM2[0] * V[0] + dM2[1] * V[4] + dM2[2] * V[8];
    const float dJ02 = dM2[0] * V[2] + dM2[1] * V[6] + dM2[2] * V[10];
    const float dJ11 = dM2[3] * V[1] + dM2[4] * V[5] + dM2[5] * V[9];
    const float dJ12 = dM2[3] * V[2] + dM2[4] * V[6] + dM2[5] * V[10];
    const float tz = 1.f / cv.tz, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = (cv.clamp_x ? 0.f : 1.f) * (-fx * tz2 * dJ02);
    const float dty = (cv.clamp_y ? 0.f : 1.f) * (-fy * tz2 * dJ12);
    const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * cv.tx) * tz3 * dJ02 + (2.f * fy * cv.ty) * tz3 * dJ12;
    gm[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
    gm[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
    gm[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;
    // 2-D mean (NDC-scaled) through the perspective divide
    const float hx = P[0] * p[0] + P[4] * p[1] + P[8] * p[2] + P[12];
    const float hy = P[1] * p[0] + P[5] * p[1] + P[9] * p[2] + P[13];
    const float hw = P[3] * p[0] + P[7] * p[1] + P[11] * p[2] + P[15];
    const float mw = 1.0f / (hw + 0.0000001f);
    const float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
    gm[0] += (P[0] * mw - P[3] * mul1) * g2x + (P[1] * mw - P[3] * mul2) * g2y;
    gm[1] += (P[4] * mw - P[7] * mul1) * g2x + (P[5] * mw - P[7] * mul2) * g2y;
    gm[2] += (P[8] * mw - P[11] * mul1) * g2x + (P[9] * mw - P[11] * mul2) * g2y;
    gm[0] += V[2] * gd; gm[1] += V[6] * gd; gm[2] += V[10] * gd;

    if (sh_mode) {
      float dx = p[0] - a.campos[0], dy = p[1] - a.campos[1], dz = p[2] - a.campos[2];
      const float len = sqrtf(dx * dx + dy * dy + dz * dz);
      const float ux = dx / len, uy = dy / len, uz = dz / len;
      sh_basis(a.deg, ux, uy, uz, Bk);
      const int nb = (a.deg + 1) * (a.deg + 1);
#pragma unroll
      for (int k = 0; k < 16; k++) if (k >= nb) Bk[k] = 0.f;
      gcs[0] = (cl & 1) ? 0.f : gcol[0]; gcs[1] = (cl & 2) ? 0.f : gcol[1]; gcs[2] = (cl & 4) ? 0.f : gcol[2];
      float w[16];
#pragma unroll
      for (int k = 0; k < 16; k++) {
        w[k] = 0.f;
        if (k < nb) w[k] = shv[3 * k] * gcs[0] + shv[3 * k + 1] * gcs[1] + shv[3 * k + 2] * gcs[2];
      }
      float gdir[3];
      sh_dir_grad(a.deg, ux, uy, uz, w, gdir);
      const float dot = ux * gdir[0] + uy * gdir[1] + uz * gdir[2];
      gm[0] += (gdir[0] - ux * dot) / len; gm[1] += (gdir[1] - uy * dot) / len; gm[2] += (gdir[2] - uz * dot) / len;
    }
    if (!need_sr) {
      gcov[0] = GS[0]; gcov[1] = 2.f * GS[1]; gcov[2] = 2.f * GS[2];
      gcov[3] = GS[4]; gcov[4] = 2.f * GS[5]; gcov[5] = GS[8];
    } else {
      float Rm[9];
      quat_to_R(g.q, Rm);
      const float s3[3] = {a.mod * g.s[0], a.mod * g.s[1], a.mod * g.s[2]};
      float Mm[9], dMm[9], dR[9];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) Mm[3 * r + j] = Rm[3 * r + j] * s3[j];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++)
          dMm[3 * r + j] = 2.f * (GS[3 * r] * Mm[j] + GS[3 * r + 1] * Mm[3 + j] + GS[3 * r + 2] * Mm[6 + j]);
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const float ds = Rm[j] * dMm[j] + Rm[3 + j] * dMm[3 + j] + Rm[6 + j] * dMm[6 + j];
        gs[j] = a.mod * ds;
#pragma unroll
        for (int r = 0; r < 3; r++) dR[3 * r + j] = s3[j] * dMm[3 * r + j];
      }
      const float r = g.q[0], x = g.q[1], y = g.q[2], z = g.q[3];
      gq[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
      gq[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
      gq[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
      gq[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
    }
  }
}
// every per-Gaussian output except the rows of dL/dsh (the callers stage those): with the chain rule of the render glue when fused
__device__ __forceinline__ void bwd_outputs(const PreBwdArgs& b, const PreArgs& a, const int i, const bool sh_mode, const bool visible,
                                            const GlueIn& g, const float (&gm)[3], const float g2x, const float g2y, const float g_op,
                                            const float (&gcol)[3], const float (&gs)[3], const float (&gq)[4], const float (&gcov)[6]) {
  // ---- outputs (with the chain rule of the render glue when fused) ----
  b.dL_dmeans3D[3 * i] = gm[0]; b.dL_dmeans3D[3 * i + 1] = gm[1]; b.dL_dmeans3D[3 * i + 2] = gm[2];
  b.dL_dmeans2D[3 * i] = g2x; b.dL_dmeans2D[3 * i + 1] = g2y; b.dL_dmeans2D[3 * i + 2] = 0.f;
  if (b.dL_dcolors) {
    const bool pc = !sh_mode;
    b.dL_dcolors[3 * i] = pc ? gcol[0] : 0.f; b.dL_dcolors[3 * i + 1] = pc ? gcol[1] : 0.f; b.dL_dcolors[3 * i + 2] = pc ? gcol[2] : 0.f;
  }
  if (a.glue) {
    // opacity = sigmoid(raw): d/draw = o (1 - o)
    b.dL_dopac[i] = visible ? g_op * g.o * (1.f - g.o) : 0.f;
    if (b.dL_dscales) {
      // scales = exp(raw): d/draw = s ; isotropic: the three columns collapse onto column 0
      if (a.isotropic) b.dL_dscales[i] = visible ? (gs[0] * g.es[0] + gs[1] * g.es[1] + gs[2] * g.es[2]) : 0.f;
      else {
        b.dL_dscales[3 * i] = visible ? gs[0] * g.es[0] : 0.f;
        b.dL_dscales[3 * i + 1] = visible ? gs[1] * g.es[1] : 0.f;
        b.dL_dscales[3 * i + 2] = visible ? gs[2] * g.es[2] : 0.f;
      }
      if (b.dL_dd_scaling) { b.dL_dd_scaling[3 * i] = gs[0]; b.dL_dd_scaling[3 * i + 1] = gs[1]; b.dL_dd_scaling[3 * i + 2] = gs[2]; }
    }
    if (b.dL_drots) {
      // q = v / |v| : dL/dv = (gq - q (q . gq)) / |v|
      float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (visible) {
        const float dot = g.q[0] * gq[0] + g.q[1] * gq[1] + g.q[2] * gq[2] + g.q[3] * gq[3];
        o4 = make_float4((gq[0] - g.q[0] * dot) / g.vnorm, (gq[1] - g.q[1] * dot) / g.vnorm,
                         (gq[2] - g.q[2] * dot) / g.vnorm, (gq[3] - g.q[3] * dot) / g.vnorm);
      }
      reinterpret_cast<float4*>(b.dL_drots)[i] = o4;
    }
  } else {
    b.dL_dopac[i] = g_op;
    if (b.dL_dscales) { b.dL_dscales[3 * i] = gs[0]; b.dL_dscales[3 * i + 1] = gs[1]; b.dL_dscales[3 * i + 2] = gs[2]; }
    if (b.dL_drots) reinterpret_cast<float4*>(b.dL_drots)[i] = make_float4(gq[0], gq[1], gq[2], gq[3]);
  }
  if (b.dL_dcov3D) {
#pragma unroll
    for (int k = 0; k < 6; k++) b.dL_dcov3D[6 * i + k] = gcov[k];
  }
}

__global__ __launch_bounds__(256, 3) void preprocess_bwd_kernel(PreBwdArgs b) {
  extern __shared__ float s_sh[];
  __shared__ unsigned char s_list[256];
  __shared__ int s_wcount[4];
  const PreArgs& a = b.f;
  const bool sh_mode = (a.colors_precomp == nullptr);
  const int sh_per = a.shs_rest ? (a.M - 1) * 3 : a.M * 3;
  const int sh_first = blockIdx.x * 256, sh_count = min(256, a.N - sh_first);
  // ---- which Gaussians of this block received any gradient from the compositing backward?  A Gaussian whose
  // accumulator record is all zeros (invisible, behind saturated pixels, or never at alpha >= 1/255 in a tile it
  // overlaps: 93 % of the bench scene) has an exactly zero gradient: its thread writes the zeros right away, and
  // the others are COMPACTED onto the first threads of the block, so that the long arithmetic below runs on full
  // lanes of few waves instead of a few lanes of every wave, and nothing is read for the rest.
  const int t0 = threadIdx.x, lane0 = t0 & 63, wave0 = t0 >> 6;
  const int i0 = blockIdx.x * 256 + t0;
  bool touched0 = false;
  // (a frame whose instance arena overflowed composited truncated lists: its gradients are undefined, so every
  // Gaussian is treated as untouched and the optimizer sees zeros until the host notices the flag and re-renders)
  const bool overflowed = b.counters != nullptr && b.counters[1] != 0u;
  if (i0 < a.N && a.radii[i0] > 0) {
    float4* acc4 = reinterpret_cast<float4*>(b.gacc + (size_t)i0 * RIGGS_GACC);
    const float4 q0 = acc4[0], q1 = acc4[1], q2 = acc4[2];
    touched0 = (q0.x != 0.f) || (q0.y != 0.f) || (q0.z != 0.f) || (q0.w != 0.f) || (q1.x != 0.f) || (q1.y != 0.f) ||
               (q1.z != 0.f) || (q1.w != 0.f) || (q2.x != 0.f) || (q2.y != 0.f);
    if (touched0 && overflowed) {  // nobody will consume (and clear) this record below
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      acc4[0] = z; acc4[1] = z; acc4[2] = z;
      touched0 = false;
    }
  }
  const uint64_t tmask = __builtin_amdgcn_ballot_w64(touched0);
  // cfg.sparse_zero: the gradient buffers still hold what the previous call with this workspace left — zeros outside the
  // rows it listed (its bits are still in the workspace) — so only the rows of (previous | current) need a store: the
  // zero fill of the other 86-93 % was two thirds of this kernel's HBM traffic
  uint64_t pmask = ~0ull;
  if (b.sparse_zero) pmask = (blockIdx.x * 256 + wave0 * 64 < a.N) ? b.touched_bits[blockIdx.x * 4 + wave0] : 0ull;
  const bool was0 = (pmask >> lane0) & 1ull;
  __shared__ unsigned long long s_need[4];
  if (lane0 == 0) s_need[wave0] = tmask | pmask;
  if (lane0 == 0) s_wcount[wave0] = __builtin_popcountll(tmask);
  __syncthreads();
  int tbase = 0;
  for (int w = 0; w < wave0; w++) tbase += s_wcount[w];
  const int n_work = s_wcount[0] + s_wcount[1] + s_wcount[2] + s_wcount[3];
  if (lane0 == 0 && blockIdx.x * 256 + wave0 * 64 < a.N) b.touched_bits[blockIdx.x * 4 + wave0] = tmask;
  if (t0 == 0) b.block_touched[blockIdx.x] = (uint32_t)n_work;
  if (touched0) s_list[tbase + __builtin_popcountll(tmask & ((1ull << lane0) - 1ull))] = (unsigned char)t0;
  if (i0 < a.N && !touched0 && was0) bwd_zero_row(b, a, i0, sh_mode);
  // the block's coefficients are staged through LDS when most of it has work, else the few read their own rows
  const bool staged = sh_mode && sh_per > 0 && n_work > 64;
  if (staged) {
    const float* sh_src = (a.shs_rest ? a.shs_rest : a.shs) + (size_t)sh_first * sh_per;
    // (odd record length + 16-byte aligned run: direct-to-LDS loads, as in the forward; else through registers)
    if ((sh_per & 1) && sh_per <= 48 && ((reinterpret_cast<uintptr_t>(sh_src) & 15) == 0)) sh_stage_dma(sh_src, sh_count * sh_per, s_sh);
    else sh_stage_in(sh_src, sh_per, sh_count, s_sh);
  }
  __syncthreads();  // s_list (and the staged coefficients)
  const bool in_range = t0 < n_work;                       // from here on: "this thread has a Gaussian to work on"
  const int slot = in_range ? (int)s_list[t0] : t0;        // its place in the block (LDS row of its coefficients)
  const int i = blockIdx.x * 256 + slot;
  const bool visible = in_range;
  float Bk[16], gcs[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; k++) Bk[k] = 0.f;
  const bool need_sr = (a.cov3D_precomp == nullptr);
  float gm[3] = {0.f, 0.f, 0.f};
  float g2x = 0.f, g2y = 0.f, g_op = 0.f;
  float gcol[3] = {0.f, 0.f, 0.f};
  float gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f}, gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  GlueIn g;
  g.vnorm = 1.f;
  // ---- every global load of this Gaussian up front, nothing used yet: left to itself the compiler sinks each load
  // next to its use and the arithmetic below becomes a chain of ~45 exposed round trips (25 us for a wave that has
  // the SIMD to itself, and few waves have work here)
  RawIn raw;
  float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
  float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint8_t cl = 0;
  float shv[48];  // coefficient k, channel c at [3 k + c] whatever the parameter layout
#pragma unroll
  for (int e = 0; e < 48; e++) shv[e] = 0.f;
  if (visible) {
    load_raw(a, i, raw, need_sr);
    // the accumulators are SELF-CLEANING: whoever consumes a touched record puts the zeros back (7 % of the records in
    // the bench scene), so the compositing backward of the next frame finds them cleared without a 14 MB fill per frame
    float4* acc4 = reinterpret_cast<float4*>(b.gacc + (size_t)i * RIGGS_GACC);
    q0 = acc4[0]; q1 = acc4[1]; q2 = acc4[2];
    { const float4 z = make_float4(0.f, 0.f, 0.f, 0.f); acc4[0] = z; acc4[1] = z; acc4[2] = z; }
#pragma unroll
    for (int k = 0; k < 6; k++) c6[k] = a.cov3D[6 * i + k];
    if (sh_mode) {
      cl = a.clamped[i];
      const float* mine = staged ? s_sh + slot * sh_lds_stride(sh_per)
                                 : (a.shs_rest ? a.shs_rest : a.shs) + (size_t)i * sh_per;
      if (a.shs_rest) {
        shv[0] = a.shs[3 * i]; shv[1] = a.shs[3 * i + 1]; shv[2] = a.shs[3 * i + 2];
#pragma unroll
        for (int e = 0; e < 45; e++) if (e < sh_per) shv[3 + e] = mine[e];
      } else {
#pragma unroll
        for (int e = 0; e < 48; e++) if (e < sh_per) shv[e] = mine[e];
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (visible) bwd_math(a, sh_mode, need_sr, raw, q0, q1, q2, c6, cl, shv, g, gm, g2x, g2y, g_op, gcol, gs, gq, gcov, Bk, gcs);
  // ---- dL/dsh: per-thread records -> LDS -> full-line stores (zeros for invisible Gaussians)
  if (sh_mode) {
    const int koff = a.shs_rest ? 3 : 0;
    if (sh_per > 0) {
      __syncthreads();  // every thread is done reading the staged coefficients
      {  // zero rows for the Gaussians without work
        float* row = s_sh + t0 * sh_lds_stride(sh_per);
        for (int k = 0; k < sh_per; k++) row[k] = 0.f;
      }
      __syncthreads();
      if (in_range) {
        float* mine = s_sh + slot * sh_lds_stride(sh_per);
#pragma unroll
        for (int k = 0; k < 16; k++) {
          if (k < a.M && 3 * k >= koff) {
            mine[3 * k - koff] = Bk[k] * gcs[0]; mine[3 * k + 1 - koff] = Bk[k] * gcs[1]; mine[3 * k + 2 - koff] = Bk[k] * gcs[2];
          }
        }
      }
      __syncthreads();
      float* dst = (a.shs_rest ? b.dL_dsh_rest : b.dL_dsh) + (size_t)sh_first * sh_per;
      if (b.sparse_zero) {  // only the rows listed now or last time: a wave stores its own 64 Gaussians' rows, lane = float
        uint64_t m = s_need[wave0];
        const int stride = sh_lds_stride(sh_per);
        while (m) {
          const int row = wave0 * 64 + __builtin_ctzll(m);
          m &= m - 1ull;
          if (row < sh_count)
            for (int c = lane0; c < sh_per; c += 64) dst[(size_t)row * sh_per + c] = s_sh[row * stride + c];
        }
      } else sh_stage_out(dst, sh_per, sh_count, s_sh);
    }
    if (a.shs_rest && in_range) {
      b.dL_dsh[3 * i] = Bk[0] * gcs[0]; b.dL_dsh[3 * i + 1] = Bk[0] * gcs[1]; b.dL_dsh[3 * i + 2] = Bk[0] * gcs[2];
    }
  }
  if (!in_range) return;
  bwd_outputs(b, a, i, sh_mode, visible, g, gm, g2x, g2y, g_op, gcol, gs, gq, gcov);
}

// The same backward as ONE WAVE per 256 Gaussians — for frames whose gradient rows are sparse (cfg.sparse_zero: a captured frame
// whose owner keeps the rows' history; 7 - 12 % of the bench scene's Gaussians receive a gradient).  The kernel above keeps 46 KB
// of LDS (the block's dL/dsh rows) and 157 registers: three workgroups per CU, 1.5 rounds of workgroups at 300 k Gaussians, and in
// each of them the long arithmetic runs on ONE wave with a third of its lanes while the other three wait at its barriers — the
// kernel trace reads 9.4 us for the screening alone and 26 us with the arithmetic.  Here a workgroup IS that one wave: it screens
// its 256 Gaussians four to a lane (all records in flight together), works through the listed ones 64 at a time, and stages only
// those rows (12 KB); twelve workgroups fit a CU: every block of the frame is resident at once.
// TWO waves per block since round 6: each screens half of the block (two records per lane) and the listed Gaussians are dealt to
// the waves in batches of 64 — a scene whose Gaussians mostly receive a gradient (the opaque-skin scene lists 28 %: 72 of a
// block's 256) used to walk two or three batches one behind the other on the one wave, each a full round of scattered loads,
// ~1 000 vector instructions and stores.
#define PBL_WAVES 2
__global__ __launch_bounds__(64 * PBL_WAVES) void preprocess_bwd_lean_kernel(PreBwdArgs b) {
  __shared__ unsigned char s_list[256];
  __shared__ int s_cnt[PBL_WAVES];
  const PreArgs& a = b.f;
  const bool sh_mode = (a.colors_precomp == nullptr);
  const int sh_per = a.shs_rest ? (a.M - 1) * 3 : a.M * 3;
  const bool need_sr = (a.cov3D_precomp == nullptr);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int SUBS = 4 / PBL_WAVES;  // runs of 64 Gaussians a wave screens
  const int first = blockIdx.x * 256, count = min(256, a.N - first);
  const bool overflowed = b.counters != nullptr && b.counters[1] != 0u;
  // ---- screening: which of the block's Gaussians received a gradient (see the kernel above)
  // (one round trip for everything the screening reads: the four records of the lane, their radii — a culled Gaussian's record is
  // zero, the radius only spares the compare — and the previous call's bits)
  float4 rq[SUBS][3];
  uint64_t prev[SUBS];
#pragma unroll
  for (int ss = 0; ss < SUBS; ss++) {
    const int sub = wave * SUBS + ss;
    const int i0 = first + sub * 64 + lane;
    rq[ss][0] = rq[ss][1] = rq[ss][2] = make_float4(0.f, 0.f, 0.f, 0.f);
    prev[ss] = ~0ull;
    if (b.sparse_zero) prev[ss] = (sub * 64 < count) ? b.touched_bits[blockIdx.x * 4 + sub] : 0ull;
    if (i0 < a.N) {
      const float4* acc4 = reinterpret_cast<const float4*>(b.gacc + (size_t)i0 * RIGGS_GACC);
      rq[ss][0] = acc4[0]; rq[ss][1] = acc4[1]; rq[ss][2] = acc4[2];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  uint64_t tmask[SUBS], need[SUBS];
  bool tch[SUBS];
  int n_mine = 0;  // listed by this wave
#pragma unroll
  for (int ss = 0; ss < SUBS; ss++) {
    const int sub = wave * SUBS + ss;
    const int i0 = first + sub * 64 + lane;
    const float4 q0 = rq[ss][0], q1 = rq[ss][1], q2 = rq[ss][2];
    bool touched = (q0.x != 0.f) || (q0.y != 0.f) || (q0.z != 0.f) || (q0.w != 0.f) || (q1.x != 0.f) || (q1.y != 0.f) ||
                   (q1.z != 0.f) || (q1.w != 0.f) || (q2.x != 0.f) || (q2.y != 0.f);
    if (touched && overflowed) {  // (an overflowed frame back-propagates exact zeros; nobody will consume this record)
      float4* acc4 = reinterpret_cast<float4*>(b.gacc + (size_t)i0 * RIGGS_GACC);
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      acc4[0] = z; acc4[1] = z; acc4[2] = z;
      touched = false;
    }
    const uint64_t tm = __builtin_amdgcn_ballot_w64(touched);
    const uint64_t pm = prev[ss];
    tmask[ss] = tm;
    need[ss] = tm | pm;
    tch[ss] = touched;
    n_mine += __builtin_popcountll(tm);
    if (lane == 0 && sub * 64 < count) b.touched_bits[blockIdx.x * 4 + sub] = tm;
    if (i0 < a.N && !touched && ((pm >> lane) & 1ull)) bwd_zero_row(b, a, i0, sh_mode);
  }
  if (lane == 0) s_cnt[wave] = n_mine;
  __syncthreads();
  int n_work = 0, my_off = 0;
#pragma unroll
  for (int w = 0; w < PBL_WAVES; w++) { if (w < wave) my_off += s_cnt[w]; n_work += s_cnt[w]; }
  {
    int run = my_off;
#pragma unroll
    for (int ss = 0; ss < SUBS; ss++) {
      if (tch[ss]) s_list[run + __builtin_popcountll(tmask[ss] & ((1ull << lane) - 1ull))] = (unsigned char)((wave * SUBS + ss) * 64 + lane);
      run += __builtin_popcountll(tmask[ss]);
    }
  }
  if (threadIdx.x == 0) b.block_touched[blockIdx.x] = (uint32_t)n_work;
  __syncthreads();  // s_list
  float* const dst_rest = sh_mode && sh_per > 0 ? (a.shs_rest ? b.dL_dsh_rest : b.dL_dsh) + (size_t)first * sh_per : nullptr;
  const int koff = a.shs_rest ? 3 : 0;
  // ---- the listed Gaussians, 64 at a time, batches dealt to the waves
  for (int base = wave * 64; base < n_work; base += 64 * PBL_WAVES) {
    const int tq = base + lane;
    const bool in_range = tq < n_work;
    const int slot = in_range ? (int)s_list[tq] : 0;
    const int i = first + slot;
    float Bk[16], gcs[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; k++) Bk[k] = 0.f;
    float gm[3] = {0.f, 0.f, 0.f};
    float g2x = 0.f, g2y = 0.f, g_op = 0.f;
    float gcol[3] = {0.f, 0.f, 0.f};
    float gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f}, gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    GlueIn g;
    g.vnorm = 1.f;
    RawIn raw;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
    float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint8_t cl = 0;
    float shv[48];
#pragma unroll
    for (int e = 0; e < 48; e++) shv[e] = 0.f;
    if (in_range) {  // every global load of the Gaussian up front (see the kernel above)
      load_raw(a, i, raw, need_sr);
      float4* acc4 = reinterpret_cast<float4*>(b.gacc + (size_t)i * RIGGS_GACC);
      q0 = acc4[0]; q1 = acc4[1]; q2 = acc4[2];
      { const float4 z = make_float4(0.f, 0.f, 0.f, 0.f); acc4[0] = z; acc4[1] = z; acc4[2] = z; }  // (self-cleaning accumulators)
#pragma unroll
      for (int k = 0; k < 6; k++) c6[k] = a.cov3D[6 * i + k];
      if (sh_mode) {
        cl = a.clamped[i];
        // (its own coefficient row, 45 loads of 64 different lines each: staging the batch's rows through LDS with a row per
        // direct-to-LDS load instruction measured the same, 22.7 against 22.9 us, and cost two barriers)
        const float* mine = (a.shs_rest ? a.shs_rest : a.shs) + (size_t)i * sh_per;
        if (a.shs_rest) {
          shv[0] = a.shs[3 * i]; shv[1] = a.shs[3 * i + 1]; shv[2] = a.shs[3 * i + 2];
#pragma unroll
          for (int e = 0; e < 45; e++) if (e < sh_per) shv[3 + e] = mine[e];
        } else {
#pragma unroll
          for (int e = 0; e < 48; e++) if (e < sh_per) shv[e] = mine[e];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (in_range) bwd_math(a, sh_mode, need_sr, raw, q0, q1, q2, c6, cl, shv, g, gm, g2x, g2y, g_op, gcol, gs, gq, gcov, Bk, gcs);
    if (sh_mode) {
      if (sh_per > 0) {
        // dL/dsh rows of the batch: every listed lane stores its own row (45 stores nobody waits for; through LDS and a row per
        // store instruction the wave stood at two barriers and 2 x 20 LDS round trips: 6.5 us of a 22.7 us kernel)
        if (in_range) {
          float* mine = dst_rest + (size_t)slot * sh_per;
#pragma unroll
          for (int k = 0; k < 16; k++) {
            if (k < a.M && 3 * k >= koff) {
              mine[3 * k - koff] = Bk[k] * gcs[0]; mine[3 * k + 1 - koff] = Bk[k] * gcs[1]; mine[3 * k + 2 - koff] = Bk[k] * gcs[2];
            }
          }
        }
      }
      if (a.shs_rest && in_range) {
        b.dL_dsh[3 * i] = Bk[0] * gcs[0]; b.dL_dsh[3 * i + 1] = Bk[0] * gcs[1]; b.dL_dsh[3 * i + 2] = Bk[0] * gcs[2];
      }
    }
    if (in_range) bwd_outputs(b, a, i, sh_mode, true, g, gm, g2x, g2y, g_op, gcol, gs, gq, gcov);
  }
  // ---- zero rows of dL/dsh for the Gaussians that need a store and have no gradient (listed by the previous call, or — without
  // cfg.sparse_zero — every other one)
  if (dst_rest) {
#pragma unroll
    for (int ss = 0; ss < SUBS; ss++) {
      const int sub = wave * SUBS + ss;
      uint64_t m = need[ss] & ~tmask[ss];
      while (m) {
        const int row = sub * 64 + __builtin_ctzll(m);
        m &= m - 1ull;
        if (row < count)
          for (int c = lane; c < sh_per; c += 64) dst_rest[(size_t)row * sh_per + c] = 0.f;
      }
    }
  }
}

// host-side launchers (called from capi.hip)
int launch_preprocess_fwd(const PreArgs& a, hipStream_t s) {
  if (a.N == 0) return 0;
  const int per = a.shs_rest ? (a.M - 1) * 3 : a.M * 3;
  const size_t lds = (a.colors_precomp || a.defer_color) ? 0 : (size_t)256 * (per | 1) * sizeof(float);
  hipLaunchKernelGGL(preprocess_fwd_kernel, dim3((a.N + 255) / 256), dim3(256), lds, s, a);
  return 0;
}
int launch_preprocess_bwd(const PreBwdArgs& b, hipStream_t s) {
  if (b.f.N == 0) return 0;
  const int per = b.f.shs_rest ? (b.f.M - 1) * 3 : b.f.M * 3;
  if (option(OPT_PREPROCESS_BWD_LEAN) == 1 || (option(OPT_PREPROCESS_BWD_LEAN) < 0 && b.sparse_zero)) {
    hipLaunchKernelGGL(preprocess_bwd_lean_kernel, dim3((b.f.N + 255) / 256), dim3(64 * PBL_WAVES), 0, s, b);
    return 0;
  }
  const size_t lds = b.f.colors_precomp ? 0 : (size_t)256 * (per | 1) * sizeof(float);
  hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((b.f.N + 255) / 256), dim3(256), lds, s, b);
  return 0;
}

}  // namespace riggs
