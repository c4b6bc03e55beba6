// Image loss of the trainer: L1 and SSIM (11x11 Gaussian window, sigma 1.5, zero padding) of a
// rendered image against the ground truth, and the gradient w.r.t. the rendered image that feeds the rasterizer backward.
//   l1_loss / ssim / _ssim   /root/reference/utils/loss_utils.py:17-18, 33-77 ;  used at /root/reference/train_rig.py:508-509
// The reference evaluates 5 grouped 11x11 conv2d (mu1, mu2, E[x^2], E[y^2], E[xy]) plus ~15 elementwise passes, and autograd
// replays them backwards.  Here: ONE forward launch (tile of 32x32 pixels + 5-pixel halo staged in LDS, separable window,
// the five moments kept in registers) that emits the two scalars' partial sums and three derivative maps
//   d(ssim)/d(mu1), d(ssim)/d(E[x^2]), d(ssim)/d(E[xy])
// and ONE backward launch that convolves the three maps with the (symmetric) window and combines them with the L1 sign:
//   dL/dx = g_l1 * sign(x - y) / n  +  g_ssim / n * ( G*dmu + 2 x (G*de11) + y (G*de12) ).
// Traffic: forward reads 2 images and writes 3 maps, backward reads 3 maps + 2 images and writes 1 (61 MB at 800 x 800: 8 us of
// HBM time); what the launches take is their vector instructions — see the packed forms below.
#include "common.h"

namespace riggs {

#define LS_T 32            // output tile: 32 columns ...
#define LS_TH 64           // ... x 64 rows, 512 threads, 4 outputs each in either pass
#define LS_NT 512
#define LS_R 5             // window radius (11 taps)
#define LS_S (LS_T + 2 * LS_R)    // staged columns: 42
#define LS_SH (LS_TH + 2 * LS_R)  // staged rows: 74

struct LossArgs {
  int C, H, W;
  const float *x, *y;      // rendered image, ground truth: (C, H, W)
  float* maps;             // (3, C, H, W): dm/dmu1, dm/dE[x^2], dm/dE[xy]
  float* partial;          // [blocks][2]: sum |x - y|, sum ssim_map
  float* out2;             // [l1 mean, ssim mean]
  const float *g_l1, *g_ssim;  // backward: upstream gradients of the two scalars (device scalars)
  const float* g_loss;         // ... and of the combined loss (1 - lambda) l1 + lambda (1 - ssim)
  float lambda_dssim;
  float* dx;               // (C, H, W)
  float win[2 * LS_R + 1];
};

__device__ __forceinline__ float ld_pad(const float* __restrict__ p, int yy, int xx, int H, int W) {
  return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? p[(size_t)yy * W + xx] : 0.f;
}

// Packed fp32 throughout (v_pk_mul / v_pk_fma_f32: two fp32 operations per lane and instruction): the staged images travel as
// (x, y) PAIRS, the moments as the pairs (mu1, mu2) and (E[x^2], E[y^2]) plus the lone E[xy] — three instructions per tap and
// output instead of five multiply-adds and three products, the products of a staged pixel formed once instead of once per
// output it serves.  The kernel was bound by its vector instructions (~1 000 per thread), not by the LDS.  Same operations in
// the same order per element as the scalar form: bit-identical results.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v pk_fma(float w, f2v a, f2v c) { return __builtin_elementwise_fma(f2v{w, w}, a, c); }
#define LS_PX 44  // row pitch of the staged pairs (in pairs): rows start 16-byte aligned

__global__ __launch_bounds__(LS_NT) void l1_ssim_forward_kernel(LossArgs a) {
  // 32 x 64 outputs per workgroup of 512 threads.  Both passes are register blocked: a thread produces 4 adjacent outputs from
  // 14 staged inputs (instead of 4 x 11).  The tall tile: 75 KB of LDS = two workgroups = four waves per SIMD (32 x 32 tiles of
  // 256 threads: 42 KB, three workgroups, three waves per SIMD, and 1875 workgroups = 2.4 rounds of the chip at 800 x 800
  // against 975 = 1.9 rounds here), halo overhead 1.5x instead of 1.7x.
  __shared__ f2v s_xy[LS_SH][LS_PX];
  __shared__ f2v s_m[LS_SH][LS_T + 1], s_e[LS_SH][LS_T + 1];  // after the horizontal pass: (mu1, mu2), (E[x^2], E[y^2])
  __shared__ float s_c[LS_SH][LS_T + 1];                      // ... E[xy]
  __shared__ float s_red[2][LS_NT / 64];
  const int c = blockIdx.z, tx0 = blockIdx.x * LS_T, ty0 = blockIdx.y * LS_TH;
  const int tid = threadIdx.x;
  const float* X = a.x + (size_t)c * a.H * a.W;
  const float* Y = a.y + (size_t)c * a.H * a.W;
  {
    constexpr int NST = (LS_SH * LS_S + LS_NT - 1) / LS_NT;
    float gx[NST], gy[NST];
#pragma unroll
    for (int i = 0; i < NST; i++) {  // (all of the thread's loads in flight before the first LDS write)
      const int e = tid + LS_NT * i, r = e / LS_S, q = e % LS_S;
      const bool in = e < LS_SH * LS_S;
      gx[i] = in ? ld_pad(X, ty0 + r - LS_R, tx0 + q - LS_R, a.H, a.W) : 0.f;
      gy[i] = in ? ld_pad(Y, ty0 + r - LS_R, tx0 + q - LS_R, a.H, a.W) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int e = tid + LS_NT * i, r = e / LS_S, q = e % LS_S;
      if (e < LS_SH * LS_S) s_xy[r][q] = f2v{gx[i], gy[i]};
    }
  }
  __syncthreads();
  float win[2 * LS_R + 1];
#pragma unroll
  for (int k = 0; k <= 2 * LS_R; k++) win[k] = a.win[k];
  // horizontal pass: 74 rows x 8 groups of 4 columns
  for (int e = tid; e < LS_SH * (LS_T / 4); e += LS_NT) {
    const int r = e >> 3, q0 = (e & 7) * 4;
    f2v p[14], pp[14];
    float pc[14];
#pragma unroll
    for (int k = 0; k < 14; k++) {
      p[k] = s_xy[r][q0 + k];
      pp[k] = p[k] * p[k];
      pc[k] = p[k].x * p[k].y;
    }
#pragma unroll
    for (int o = 0; o < 4; o++) {
      f2v m = f2v{0.f, 0.f}, ee = f2v{0.f, 0.f};
      float e12 = 0.f;
#pragma unroll
      for (int k = 0; k <= 2 * LS_R; k++) {
        m = pk_fma(win[k], p[o + k], m);
        ee = pk_fma(win[k], pp[o + k], ee);
        e12 = fmaf(win[k], pc[o + k], e12);
      }
      s_m[r][q0 + o] = m; s_e[r][q0 + o] = ee; s_c[r][q0 + o] = e12;
    }
  }
  __syncthreads();
  // vertical pass: thread = (column lx, 4 consecutive rows ly0..ly0+3)
  const int lx = tid & 31, ly0 = (tid >> 5) * 4;
  f2v mo_m[4], mo_e[4];
  float mo_c[4];
  {
    f2v cm[14], ce[14];
    float cc[14];
#pragma unroll
    for (int k = 0; k < 14; k++) { cm[k] = s_m[ly0 + k][lx]; ce[k] = s_e[ly0 + k][lx]; cc[k] = s_c[ly0 + k][lx]; }
#pragma unroll
    for (int o = 0; o < 4; o++) {
      f2v m = f2v{0.f, 0.f}, ee = f2v{0.f, 0.f};
      float e12 = 0.f;
#pragma unroll
      for (int k = 0; k <= 2 * LS_R; k++) {
        m = pk_fma(win[k], cm[o + k], m);
        ee = pk_fma(win[k], ce[o + k], ee);
        e12 = fmaf(win[k], cc[o + k], e12);
      }
      mo_m[o] = m; mo_e[o] = ee; mo_c[o] = e12;
    }
  }
  float ssim_sum = 0.f, ad_sum = 0.f;
  const size_t plane = (size_t)a.C * a.H * a.W;
  const int px = tx0 + lx;
#pragma unroll
  for (int o = 0; o < 4; o++) {
    const int py = ty0 + ly0 + o;
    if (px < a.W && py < a.H) {
      const float m1 = mo_m[o].x, m2 = mo_m[o].y, e11 = mo_e[o].x, e22 = mo_e[o].y, e12 = mo_c[o];
      const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;  // loss_utils.py:67-68
      const float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
      const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
      const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1 + s2 + C2;
      const float inv = 1.0f / (Cc * D);
      const float ssim = A * B * inv;
      const size_t idx = ((size_t)c * a.H + py) * a.W + px;
      a.maps[idx] = 2.f * m2 * (B - A) * inv - ssim * 2.f * m1 * (D - Cc) * inv;  // d/dmu1 (through sigma1^2, sigma12 too)
      a.maps[plane + idx] = -ssim / D;                                            // d/dE[x^2]
      a.maps[2 * plane + idx] = 2.f * A * inv;                                    // d/dE[xy]
      ssim_sum += ssim;
      const f2v ctr = s_xy[ly0 + o + LS_R][lx + LS_R];
      ad_sum += fabsf(ctr.x - ctr.y);
    }
  }
  const float ad = wave_sum(ad_sum), ss = wave_sum(ssim_sum);
  if ((tid & 63) == 63) { s_red[0][tid >> 6] = ad; s_red[1][tid >> 6] = ss; }
  __syncthreads();
  if (tid == 0) {
    const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    a.partial[2 * b] = ((s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3])) + ((s_red[0][4] + s_red[0][5]) + (s_red[0][6] + s_red[0][7]));
    a.partial[2 * b + 1] = ((s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3])) + ((s_red[1][4] + s_red[1][5]) + (s_red[1][6] + s_red[1][7]));
  }
}

// fixed-order sum of the per-workgroup partials (deterministic), / n
__global__ __launch_bounds__(1024) void l1_ssim_finish_kernel(int n_blocks, const float* __restrict__ partial, double inv_n,
                                                              float lambda_dssim, float* __restrict__ out2) {
  __shared__ double s_w[2][16];
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < n_blocks; i += 1024) { s0 += (double)partial[2 * i]; s1 += (double)partial[2 * i + 1]; }
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
  if ((threadIdx.x & 63) == 0) { s_w[0][threadIdx.x >> 6] = s0; s_w[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int w = 0; w < 16; w++) { t0 += s_w[0][w]; t1 += s_w[1][w]; }
    const float l1 = (float)(t0 * inv_n), ss = (float)(t1 * inv_n);
    out2[0] = l1; out2[1] = ss;
    out2[2] = (1.0f - lambda_dssim) * l1 + lambda_dssim * (1.0f - ss);  // train_rig.py:509
  }
}

__global__ __launch_bounds__(LS_NT) void l1_ssim_backward_kernel(LossArgs a) {
  // (packed like the forward: the maps d/dmu1 and d/dE[x^2] travel as a pair, d/dE[xy] alone)
  __shared__ f2v s_ab[LS_SH][LS_PX];
  __shared__ float s_cc[LS_SH][LS_S + 1];
  __shared__ f2v s_hab[LS_SH][LS_T + 1];
  __shared__ float s_hc[LS_SH][LS_T + 1];
  const int c = blockIdx.z, tx0 = blockIdx.x * LS_T, ty0 = blockIdx.y * LS_TH;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.C * a.H * a.W, chan = (size_t)c * a.H * a.W;
  {
    constexpr int NST = (LS_SH * LS_S + LS_NT - 1) / LS_NT;
    float g0[NST], g1[NST], g2[NST];
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int e = tid + LS_NT * i, r = e / LS_S, q = e % LS_S;
      const bool in = e < LS_SH * LS_S;
      g0[i] = in ? ld_pad(a.maps + chan, ty0 + r - LS_R, tx0 + q - LS_R, a.H, a.W) : 0.f;
      g1[i] = in ? ld_pad(a.maps + plane + chan, ty0 + r - LS_R, tx0 + q - LS_R, a.H, a.W) : 0.f;
      g2[i] = in ? ld_pad(a.maps + 2 * plane + chan, ty0 + r - LS_R, tx0 + q - LS_R, a.H, a.W) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NST; i++) {
      const int e = tid + LS_NT * i, r = e / LS_S, q = e % LS_S;
      if (e < LS_SH * LS_S) { s_ab[r][q] = f2v{g0[i], g1[i]}; s_cc[r][q] = g2[i]; }
    }
  }
  // (this thread's four pixels of both images: asked for here, used behind the two passes)
  const int lx = tid & 31, ly0 = (tid >> 5) * 4;
  const int px = tx0 + lx;
  float xs[4], ys[4];
#pragma unroll
  for (int o = 0; o < 4; o++) {
    const int py = ty0 + ly0 + o;
    const bool in = px < a.W && py < a.H;
    const size_t idx = chan + (size_t)py * a.W + px;
    xs[o] = in ? a.x[idx] : 0.f;
    ys[o] = in ? a.y[idx] : 0.f;
  }
  __syncthreads();
  float win[2 * LS_R + 1];
#pragma unroll
  for (int k = 0; k <= 2 * LS_R; k++) win[k] = a.win[k];
  for (int e = tid; e < LS_SH * (LS_T / 4); e += LS_NT) {
    const int r = e >> 3, q0 = (e & 7) * 4;
    f2v u[14];
    float w[14];
#pragma unroll
    for (int k = 0; k < 14; k++) { u[k] = s_ab[r][q0 + k]; w[k] = s_cc[r][q0 + k]; }
#pragma unroll
    for (int o = 0; o < 4; o++) {
      f2v acc = f2v{0.f, 0.f};
      float ac = 0.f;
#pragma unroll
      for (int k = 0; k <= 2 * LS_R; k++) { acc = pk_fma(win[k], u[o + k], acc); ac = fmaf(win[k], w[o + k], ac); }
      s_hab[r][q0 + o] = acc; s_hc[r][q0 + o] = ac;
    }
  }
  __syncthreads();
  f2v cv_ab[4];
  float cv_c[4];
  {
    f2v col[14];
    float cc[14];
#pragma unroll
    for (int k = 0; k < 14; k++) { col[k] = s_hab[ly0 + k][lx]; cc[k] = s_hc[ly0 + k][lx]; }
#pragma unroll
    for (int o = 0; o < 4; o++) {
      f2v acc = f2v{0.f, 0.f};
      float ac = 0.f;
#pragma unroll
      for (int k = 0; k <= 2 * LS_R; k++) { acc = pk_fma(win[k], col[o + k], acc); ac = fmaf(win[k], cc[o + k], ac); }
      cv_ab[o] = acc; cv_c[o] = ac;
    }
  }
  const float inv_n = 1.0f / (float)plane;
  const float gt_ = a.g_loss ? a.g_loss[0] : 0.f;
  const float gl = (a.g_l1 ? a.g_l1[0] : 0.f) + (1.0f - a.lambda_dssim) * gt_;
  const float gs = (a.g_ssim ? a.g_ssim[0] : 0.f) - a.lambda_dssim * gt_;
#pragma unroll
  for (int o = 0; o < 4; o++) {
    const int py = ty0 + ly0 + o;
    if (px < a.W && py < a.H) {
      const size_t idx = chan + (size_t)py * a.W + px;
      const float x = xs[o], y = ys[o];
      const float d = x - y;
      const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);  // torch.abs backward: sign(0) = 0
      a.dx[idx] = gl * sgn * inv_n + gs * inv_n * (cv_ab[o].x + 2.f * x * cv_ab[o].y + y * cv_c[o]);
    }
  }
}

static void fill_window(LossArgs& a) {
  // gaussian(11, 1.5) of loss_utils.py:33-35: float32 tensor of the exps, divided by its float32 sum
  float g[2 * LS_R + 1], s = 0.f;
  for (int i = 0; i <= 2 * LS_R; i++) { g[i] = (float)exp(-(double)((i - LS_R) * (i - LS_R)) / (2.0 * 1.5 * 1.5)); s += g[i]; }
  for (int i = 0; i <= 2 * LS_R; i++) a.win[i] = g[i] / s;
}

}  // namespace riggs

using namespace riggs;

extern "C" {

static size_t ls_blocks(int C, int H, int W) { return (size_t)C * ((H + LS_TH - 1) / LS_TH) * ((W + LS_T - 1) / LS_T); }

size_t riggs_l1_ssim_state_floats(int32_t C, int32_t H, int32_t W) {
  return 3 * (size_t)C * H * W + 2 * ls_blocks(C, H, W);
}

int riggs_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float* image, const float* gt, float lambda_dssim,
                          float* state, float* out2, riggs_stream stream) {
  RIGGS_REQUIRE(C >= 1 && H >= 1 && W >= 1 && C <= 65535, "bad image shape");
  RIGGS_REQUIRE(image && gt && state && out2, "NULL buffer");
  LossArgs a;
  memset(&a, 0, sizeof(a));
  a.C = C; a.H = H; a.W = W; a.x = image; a.y = gt;
  a.maps = state; a.partial = state + 3 * (size_t)C * H * W; a.out2 = out2;
  fill_window(a);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((W + LS_T - 1) / LS_T, (H + LS_TH - 1) / LS_TH, C);
  {
    ProfScope ps(PROF_LOSS_FWD, s);
    hipLaunchKernelGGL(l1_ssim_forward_kernel, grid, dim3(LS_NT), 0, s, a);
    hipLaunchKernelGGL(l1_ssim_finish_kernel, dim3(1), dim3(1024), 0, s, (int)ls_blocks(C, H, W), a.partial,
                       1.0 / ((double)C * H * W), lambda_dssim, out2);
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float* image, const float* gt, const float* state,
                           float lambda_dssim, const float* g_l1, const float* g_ssim, const float* g_loss,
                           float* dL_dimage, riggs_stream stream) {
  RIGGS_REQUIRE(C >= 1 && H >= 1 && W >= 1 && C <= 65535, "bad image shape");
  RIGGS_REQUIRE(image && gt && state && dL_dimage, "NULL buffer");
  LossArgs a;
  memset(&a, 0, sizeof(a));
  a.C = C; a.H = H; a.W = W; a.x = image; a.y = gt;
  a.maps = const_cast<float*>(state); a.g_l1 = g_l1; a.g_ssim = g_ssim; a.g_loss = g_loss; a.lambda_dssim = lambda_dssim;
  a.dx = dL_dimage;
  fill_window(a);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((W + LS_T - 1) / LS_T, (H + LS_TH - 1) / LS_TH, C);
  {
    ProfScope ps(PROF_LOSS_BWD, s);
    hipLaunchKernelGGL(l1_ssim_backward_kernel, grid, dim3(LS_NT), 0, s, a);
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
