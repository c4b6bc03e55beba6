// Per-Gaussian MLP heads (second half): WeightMLP / DeformMLP
// (skeleton_utils/network_utils.py:6-112) as ONE fused launch per direction on the CDNA4 matrix cores.
//
//   x_emb (N, in_ch) fp32 -> D x [Linear(256) + ReLU], the embedding re-concatenated in front of the hidden vector
//   after layer `skip` -> Linear(out_ch)
//
// bf16 operands, fp32 accumulation (v_mfma_f32_32x32x16_bf16): the reference computes in fp32, so this path is
// opt-in and its parity bar is the bf16 one ("parity tolerance must be renegotiated for bf16").
//
// Forward: a workgroup owns 64 rows (Gaussians).  The hidden vector of those rows lives in LDS as bf16
// [64][256 (+pad)], the embedding as bf16 [64][in_pad (+pad)]; a layer is  H <- relu(H W^T + b)  with the weights
// streamed from L2 (128 KB per layer, shared by all workgroups).  Wave w computes output columns [64 w, 64 w + 64)
// of all 64 rows: 2 x 2 tiles of 32 x 32, K in steps of 16; A fragments (8 consecutive k of one row) are 16-byte LDS
// reads, B fragments (8 consecutive k of one output column = 16 contiguous bytes of a weight row) are 16-byte global
// loads.  The post-ReLU activations go back to LDS (in place, behind a barrier) and — for the backward — to HBM as
// bf16 with full-line stores.
//
// Backward (data gradient): the same tiling with the transposed weights,  dH_{l-1} = (dH_l * relu'(H_l)) W_l ;
// every layer's masked gradient is stored as bf16 for the weight gradients, which are plain (256 x N)·(N x K) GEMMs
// left to the library (hipBLASLt through torch, bf16 in / fp32 out).  The inputs of both heads are detached in the
// reference (positions and pose), so no gradient flows past the first layer.
#include "common.h"

namespace riggs {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MLP_W 256          // hidden width (the reference's W)
#define MLP_HS (MLP_W + 8) // LDS row stride of the hidden vector in bf16 (16-byte pad: conflict-free 16-byte column reads)
#define MLP_MAX_IN 128     // padded embedding width supported

__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((uint32_t)h << 16); }
// The 16-bit operand format is a template parameter of every kernel: bf16 (fp32's range, 8 significand bits) or fp16
// (11 significand bits: an eighth of bf16's rounding error per operand — parameter gradients within 1-2 % of the fp32
// mirror instead of 3-11 % — at the same MFMA rate; its narrow range is handled by the caller: the incoming gradient is
// scaled by a power of two on the device (riggs_mlp_backward: g_scale) and the parameter gradients scaled back).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool H16> __device__ __forceinline__ unsigned short f2h(float f) {
  if constexpr (H16) return __builtin_bit_cast(unsigned short, (_Float16)f);  // v_cvt_f16_f32: round to nearest even
  else return f2bf(f);
}
template <bool H16> __device__ __forceinline__ float h2f(unsigned short h) {
  if constexpr (H16) return (float)__builtin_bit_cast(_Float16, h);
  else return bf2f(h);
}
template <bool H16> __device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (H16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

struct MlpDesc {
  int N, in_ch, in_pad, out_ch, depth, skip;
  const unsigned short* Wp[10];   // packed bf16 weights, layer l: [256][K_l], K_0 = in_pad, K_{skip+1} = in_pad + 256, else 256
  const float* bias[10];          // [256]
  const unsigned short* Wout;     // [32][256] (rows >= out_ch are zero)
  const float* bout;              // [out_ch]
};

__device__ __forceinline__ int mlp_k(const MlpDesc& d, int l) { return l == 0 ? d.in_pad : (l == d.skip + 1 ? d.in_pad + MLP_W : MLP_W); }

// acc[rt][ct] += A(rows 32 rt .. +31, k) * B(k, cols col0 + 32 ct .. +31) over k in [0, K): A from LDS (row stride
// `as` bf16), B[k][n] = Wrow[n][k] with row stride `ws` bf16 in global memory
template <int RT, int CT, bool H16>
__device__ __forceinline__ void mlp_gemm_part(f32x16 (&acc)[RT][CT], const unsigned short* A, int as, const unsigned short* Wg,
                                              int ws, int K, int lane) {
  const int r = lane & 31, kq = (lane >> 5) * 8;
  // software-pipelined by one K-step: the fragments of step k + 1 are requested before the MFMAs of step k are issued
  // (two waves per SIMD cannot hide an L2 round trip per step on their own)
  bf16x8 a[RT], b[CT], an[RT], bn[CT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) a[rt] = *reinterpret_cast<const bf16x8*>(A + (size_t)(32 * rt + r) * as + kq);
#pragma unroll
  for (int ct = 0; ct < CT; ct++) b[ct] = *reinterpret_cast<const bf16x8*>(Wg + (size_t)(32 * ct + r) * ws + kq);
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int kn = (k0 + 16 < K) ? k0 + 16 : k0;  // (the last trip re-reads its own step: harmless, keeps the loop branch-free)
#pragma unroll
    for (int ct = 0; ct < CT; ct++) bn[ct] = *reinterpret_cast<const bf16x8*>(Wg + (size_t)(32 * ct + r) * ws + kn + kq);
#pragma unroll
    for (int rt = 0; rt < RT; rt++) an[rt] = *reinterpret_cast<const bf16x8*>(A + (size_t)(32 * rt + r) * as + kn + kq);
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ct = 0; ct < CT; ct++) acc[rt][ct] = mfma16<H16>(a[rt], b[ct], acc[rt][ct]);
#pragma unroll
    for (int rt = 0; rt < RT; rt++) a[rt] = an[rt];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) b[ct] = bn[ct];
  }
}

// C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int mlp_c_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <int RT, bool H16>  // 32-row tiles per workgroup (rows per workgroup = 32 RT); every wave owns 64 output columns of all of them
__global__ __launch_bounds__(256, RT == 2 ? 3 : 2) void mlp_forward_kernel(MlpDesc d, const unsigned short* __restrict__ xb,
                                                                            unsigned short* __restrict__ acts /* [depth][N][256] or NULL */,
                                                                            uint4* __restrict__ masks /* [depth][workgroups][256] or NULL */,
                                                                            float* __restrict__ out /* [N][out_ch] */) {
  constexpr int ROWS = 32 * RT;
  __shared__ unsigned short s_h[ROWS * MLP_HS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * ROWS;
  // the embedding is read from its bf16 copy in HBM ((rows rounded up to 128) x in_pad, zero padded): it is an A operand
  // of two layers only, and keeping it out of LDS lets two 128-row workgroups share a CU
  const unsigned short* xrow = xb + (size_t)row0 * d.in_pad;
  const int col0 = wave * 64;
  for (int l = 0; l < d.depth; l++) {
    f32x16 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[rt][ct][e] = 0.f;
    const int K = mlp_k(d, l);
    const unsigned short* Wl = d.Wp[l] + (size_t)col0 * K;
    if (l == 0) mlp_gemm_part<RT, 2, H16>(acc, xrow, d.in_pad, Wl, K, d.in_pad, lane);
    else if (l == d.skip + 1) {
      mlp_gemm_part<RT, 2, H16>(acc, xrow, d.in_pad, Wl, K, d.in_pad, lane);
      mlp_gemm_part<RT, 2, H16>(acc, s_h, MLP_HS, Wl + d.in_pad, K, MLP_W, lane);
    } else mlp_gemm_part<RT, 2, H16>(acc, s_h, MLP_HS, Wl, K, MLP_W, lane);
    __syncthreads();  // every wave is done reading the previous hidden vector
    uint32_t mbits[4] = {0u, 0u, 0u, 0u};  // ReLU mask of this lane's accumulator elements: bit (rt * 2 + ct) * 16 + e
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      const int col = col0 + 32 * ct + (lane & 31);
      const float b = d.bias[l][col];
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int row = 32 * rt + mlp_c_row(e, lane);
          const float v = acc[rt][ct][e] + b;
          const unsigned short hv = f2h<H16>(fmaxf(v, 0.f));
          s_h[row * MLP_HS + col] = hv;
          if ((hv & 0x7FFFu) != 0u) mbits[(rt * 2 + ct) >> 1] |= 1u << ((((rt * 2 + ct) & 1) << 4) + e);
        }
    }
    // (the data-gradient kernel uses the same tiling, so the mask travels in the accumulator layout: 16 bytes per lane
    // and layer instead of re-reading the layer's activations)
    if (masks) masks[((size_t)l * gridDim.x + blockIdx.x) * 256 + tid] = make_uint4(mbits[0], mbits[1], mbits[2], mbits[3]);
    __syncthreads();
    if (acts) {  // full-line stores of the layer's activations (operand of the weight gradients)
      unsigned short* dst = acts + ((size_t)l * d.N + row0) * MLP_W;
      for (int e = tid; e < ROWS * (MLP_W / 8); e += 256) {
        const int r = e / (MLP_W / 8), c8 = e - r * (MLP_W / 8);
        if (row0 + r < d.N)
          *reinterpret_cast<bf16x8*>(dst + (size_t)r * MLP_W + 8 * c8) = *reinterpret_cast<const bf16x8*>(s_h + r * MLP_HS + 8 * c8);
      }
    }
  }
  // output head: 32 RT rows x 32 (padded) columns, K = 256: wave w < RT takes the 32-row tile w
  if (wave < RT) {
    f32x16 acc[1][1];
#pragma unroll
    for (int e = 0; e < 16; e++) acc[0][0][e] = 0.f;
    mlp_gemm_part<1, 1, H16>(acc, s_h + (size_t)(32 * wave) * MLP_HS, MLP_HS, d.Wout, MLP_W, MLP_W, lane);
    const int col = lane & 31;
    if (col < d.out_ch) {
      const float b = d.bout[col];
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int row = row0 + 32 * wave + mlp_c_row(e, lane);
        if (row < d.N) out[(size_t)row * d.out_ch + col] = acc[0][0][e] + b;
      }
    }
  }
}

// Data-gradient pass.  d_post of the last hidden layer comes from the head (g_out W_out, K = 32 padded); then for
// l = depth-1 .. 0:  d_pre_l = d_post_l * [act_l > 0]  (cooperative 16-byte pass: LDS gradient x global activation ->
// LDS, and -> HBM as bf16 for the weight gradients), and for l >= 1  d_post_{l-1} = d_pre_l W_l[:, hidden part]
// with the transposed bf16 copy of the weights as the B operand (row = input feature k, 8 consecutive output neurons
// n contiguous).  No gradient leaves the first layer: both heads' inputs are detached in the reference.
struct MlpBwdDesc {
  int N, out_ch, depth, skip;
  const unsigned short* Wt[10];   // l >= 1: [256 (k)][256 (n)] bf16 = W_l[:, hidden part]^T
  const unsigned short* Wout_t;   // [256 (k)][32 (c)] bf16, columns >= out_ch zero
};

template <int RT, bool H16>
__global__ __launch_bounds__(256, RT == 2 ? 3 : 2) void mlp_backward_kernel(MlpBwdDesc d, const float* __restrict__ g_out,
                                                           const float* __restrict__ g_scale /* device scalar or NULL */,
                                                           const uint4* __restrict__ masks /* [depth][workgroups][256], from the forward */,
                                                           unsigned short* __restrict__ dpre /* [depth][N][256] */,
                                                           float* __restrict__ db_part /* [workgroups][depth][256] */) {
  constexpr int ROWS = 32 * RT;
  __shared__ unsigned short s_d[ROWS * MLP_HS];
  __shared__ unsigned short s_g[ROWS * 40];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * ROWS;
  const float gs = g_scale ? g_scale[0] : 1.0f;
  for (int e = tid; e < ROWS * 32; e += 256) {
    const int r = e >> 5, c = e & 31;
    float v = 0.f;
    if (row0 + r < d.N && c < d.out_ch) v = g_out[(size_t)(row0 + r) * d.out_ch + c] * gs;
    s_g[r * 40 + c] = f2h<H16>(v);
  }
  __syncthreads();
  const int col0 = wave * 64;
  for (int l = d.depth - 1; l >= 0; l--) {
    // ---- d_post_l for this wave's 64 columns
    f32x16 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[rt][ct][e] = 0.f;
    if (l == d.depth - 1) mlp_gemm_part<RT, 2, H16>(acc, s_g, 40, d.Wout_t + (size_t)col0 * 32, 32, 32, lane);
    else mlp_gemm_part<RT, 2, H16>(acc, s_d, MLP_HS, d.Wt[l + 1] + (size_t)col0 * MLP_W, MLP_W, MLP_W, lane);
    __syncthreads();  // every wave is done reading d_pre_{l+1}
    // ---- d_pre_l = d_post_l where the forward's activation was positive (mask bits in this lane's accumulator layout)
    const uint4 mk = masks[((size_t)l * gridDim.x + blockIdx.x) * 256 + tid];
    const uint32_t mbits[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      const int col = col0 + 32 * ct + (lane & 31);
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
        const uint32_t m16 = mbits[(rt * 2 + ct) >> 1] >> (((rt * 2 + ct) & 1) << 4);
#pragma unroll
        for (int e = 0; e < 16; e++)
          s_d[(32 * rt + mlp_c_row(e, lane)) * MLP_HS + col] = ((m16 >> e) & 1u) ? f2h<H16>(acc[rt][ct][e]) : (unsigned short)0;
      }
    }
    __syncthreads();
    // ---- out to HBM (operand of the weight gradients), full lines
    unsigned short* dl = dpre + ((size_t)l * d.N + row0) * MLP_W;
    for (int e = tid; e < ROWS * (MLP_W / 8); e += 256) {
      const int r = e / (MLP_W / 8), c8 = e - r * (MLP_W / 8);
      if (row0 + r < d.N)
        *reinterpret_cast<bf16x8*>(dl + (size_t)r * MLP_W + 8 * c8) = *reinterpret_cast<const bf16x8*>(s_d + r * MLP_HS + 8 * c8);
    }
    // bias gradient of the layer: this workgroup's column sums (thread = column; summed over the workgroups in a fixed
    // order by the caller — deterministic, and cheaper than a column reduction of the (N, 256) tensor)
    {
      float sum = 0.f;
#pragma unroll 8
      for (int r = 0; r < ROWS; r++) sum += h2f<H16>(s_d[r * MLP_HS + tid]);
      db_part[((size_t)blockIdx.x * d.depth + l) * MLP_W + tid] = sum;
    }
  }
}

// Positional encoding straight into the kernels' operand: row n = [x_n, sin(2^k x_n), cos(2^k x_n) (k < multires), tail, 0 ...]
// as bf16, (rows rounded up to 128) x in_pad — get_embedder of utils/time_utils.py:208-256 followed by the concatenation
// with a per-call constant vector (DeformMLP's pose), instead of ~45 elementwise launches and a 75 MB fp32 intermediate.
template <bool H16>
__global__ __launch_bounds__(256) void mlp_embed_kernel(int N, int n_rows, int multires, int n_tail, int in_pad,
                                                        const float* __restrict__ x, const float* __restrict__ tail,
                                                        unsigned short* __restrict__ out) {
  // thread = 8 consecutive columns of one row: 16-byte stores, consecutive threads -> consecutive addresses
  const int segs = in_pad >> 3;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)n_rows * segs) return;
  const int n = (int)(t / segs), c0 = (int)(t - (size_t)n * segs) * 8;
  const int pe = 3 * (1 + 2 * multires);
  bf16x8 v;
#pragma unroll
  for (int q = 0; q < 8; q++) v[q] = 0;
  if (n < N) {
    const float xv[3] = {x[3 * n], x[3 * n + 1], x[3 * n + 2]};
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int c = c0 + q;
      float f = 0.f;
      if (c < 3) f = xv[c];
      else if (c < pe) {
        const int k = (c - 3) / 6, w = (c - 3) - 6 * k;
        const float a = xv[w % 3] * (float)(1 << k);
        f = (w < 3) ? sinf(a) : cosf(a);
      } else if (c < pe + n_tail) f = tail[c - pe];
      v[q] = (short)f2h<H16>(f);
    }
  }
  *reinterpret_cast<bf16x8*>(out + (size_t)n * in_pad + c0) = v;
}

// fp32 master weights -> the bf16 operand layouts of both kernels, in ONE launch (the masters change every optimizer step,
// and ~60 small conversion / padding / transposition launches per MLP were a tenth of a heads-on training iteration)
struct MlpPackDesc {
  int in_ch, in_pad, out_ch, depth, skip;
  const float* W[10];            // (256, K_true): K_true = in_ch, in_ch + 256 (layer skip + 1) or 256
  const float* Wout;             // (out_ch, 256)
  unsigned short* Wp[10];        // (256, K_pad)
  unsigned short* Wt[10];        // l >= 1: (256 k, 256 n) = hidden part transposed
  unsigned short* Wout_p;        // (32, 256)
  unsigned short* Wout_t;        // (256, 32)
};
template <bool H16>
__global__ __launch_bounds__(256) void mlp_pack_kernel(MlpPackDesc d) {
  const int l = blockIdx.y;  // depth = the head
  const int tid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  if (l == d.depth) {
    for (int e = tid; e < 32 * MLP_W; e += stride) {
      const int c = e / MLP_W, k = e - c * MLP_W;
      d.Wout_p[e] = f2h<H16>(c < d.out_ch ? d.Wout[(size_t)c * MLP_W + k] : 0.f);
    }
    for (int e = tid; e < MLP_W * 32; e += stride) {
      const int k = e >> 5, c = e & 31;
      d.Wout_t[e] = f2h<H16>(c < d.out_ch ? d.Wout[(size_t)c * MLP_W + k] : 0.f);
    }
    return;
  }
  const bool first = (l == 0), sk = (l == d.skip + 1);
  const int k_true = first ? d.in_ch : (sk ? d.in_ch + MLP_W : MLP_W);
  const int k_pad = first ? d.in_pad : (sk ? d.in_pad + MLP_W : MLP_W);
  const int hoff_true = sk ? d.in_ch : 0, hoff_pad = sk ? d.in_pad : 0;  // where the hidden part starts
  const float* W = d.W[l];
  for (int e = tid; e < MLP_W * k_pad; e += stride) {
    const int n = e / k_pad, k = e - n * k_pad;
    float v = 0.f;
    if (first || (sk && k < hoff_pad)) { if (k < d.in_ch) v = W[(size_t)n * k_true + k]; }
    else v = W[(size_t)n * k_true + hoff_true + (k - hoff_pad)];
    d.Wp[l][e] = f2h<H16>(v);
  }
  if (!first)
    for (int e = tid; e < MLP_W * MLP_W; e += stride) {
      const int k = e >> 8, n = e & 255;
      d.Wt[l][e] = f2h<H16>(W[(size_t)n * k_true + hoff_true + k]);
    }
}

// Self-test of the fragment layouts this file assumes (A = identity against an ASYMMETRIC B): D must equal B.
__global__ __launch_bounds__(64) void mlp_layout_probe_kernel(float* __restrict__ out /* [32][32] */) {
  __shared__ unsigned short s_a[32 * 24], s_b[32 * 24];  // A[i][k] (i < 32, k < 16), Bt[n][k] = B[k][n]
  const int lane = threadIdx.x;
  for (int e = lane; e < 32 * 16; e += 64) {
    const int i = e / 16, k = e % 16;
    s_a[i * 24 + k] = f2bf(i == k ? 1.f : 0.f);               // A = [I_16; 0]
    s_b[i * 24 + k] = f2bf(k < 8 ? (float)(32 * k + i) : -(float)(32 * (k - 8) + i + 1));  // B[k][n], n = i: distinct, exact in bf16
  }
  __syncthreads();
  f32x16 acc[1][1];
  for (int e = 0; e < 16; e++) acc[0][0][e] = 0.f;
  const int r = lane & 31, kq = (lane >> 5) * 8;
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(s_a + r * 24 + kq);
  const bf16x8 b = *reinterpret_cast<const bf16x8*>(s_b + r * 24 + kq);
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0][0], 0, 0, 0);
  for (int e = 0; e < 16; e++) out[mlp_c_row(e, lane) * 32 + (lane & 31)] = acc[0][0][e];
}

}  // namespace riggs

using namespace riggs;

extern "C" {

int riggs_mlp_layout_probe(float* out32x32, riggs_stream stream) {
  hipLaunchKernelGGL(mlp_layout_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out32x32);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

// 128-row workgroups: 4 x 2 tiles of 32 x 32 per wave (64-row workgroups with 2 x 2 tiles cost 0.84 / 1.01 ms against
// 0.66 / 0.63 ms: twice the weight loads per MFMA)
#define MLP_RT 4

static int mlp_fill(MlpDesc& d, int32_t N, int32_t in_ch, int32_t out_ch, int32_t depth, int32_t skip, const void* const* Wp,
                    const float* const* bias, const void* Wout, const float* bout) {
  RIGGS_REQUIRE(N >= 0 && depth >= 1 && depth <= 10, "MLP depth out of range");
  RIGGS_REQUIRE(in_ch >= 1 && in_ch <= MLP_MAX_IN, "MLP input width must be <= 128");
  RIGGS_REQUIRE(out_ch >= 1 && out_ch <= 32, "MLP output width must be <= 32");
  RIGGS_REQUIRE(skip >= 0 && skip < depth - 1, "MLP skip layer out of range");
  d.N = N; d.in_ch = in_ch; d.in_pad = (in_ch + 31) & ~31; d.out_ch = out_ch; d.depth = depth; d.skip = skip;
  for (int l = 0; l < depth; l++) { d.Wp[l] = (const unsigned short*)Wp[l]; d.bias[l] = bias[l]; RIGGS_REQUIRE(Wp[l] && bias[l], "MLP layer pointers"); }
  d.Wout = (const unsigned short*)Wout; d.bout = bout;
  RIGGS_REQUIRE(Wout && bout, "MLP head pointers");
  return 0;
}

int riggs_mlp_forward(int32_t N, int32_t in_ch, int32_t out_ch, int32_t depth, int32_t skip, const void* const* weights_bf16,
                      const float* const* biases, const void* w_out_bf16, const float* b_out, const void* x_emb_bf16,
                      void* acts_bf16, void* relu_masks, float* out, int32_t fp16, riggs_stream stream) {
  MlpDesc d;
  int rc = mlp_fill(d, N, in_ch, out_ch, depth, skip, weights_bf16, biases, w_out_bf16, b_out);
  if (rc) return rc;
  if (N == 0) return 0;
  RIGGS_REQUIRE(x_emb_bf16 && out, "MLP input / output pointers");
  if (fp16) hipLaunchKernelGGL((mlp_forward_kernel<MLP_RT, true>), dim3((N + 127) / 128), dim3(256), 0, (hipStream_t)stream, d,
                               (const unsigned short*)x_emb_bf16, (unsigned short*)acts_bf16, (uint4*)relu_masks, out);
  else hipLaunchKernelGGL((mlp_forward_kernel<MLP_RT, false>), dim3((N + 127) / 128), dim3(256), 0, (hipStream_t)stream, d,
                          (const unsigned short*)x_emb_bf16, (unsigned short*)acts_bf16, (uint4*)relu_masks, out);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_mlp_backward(int32_t N, int32_t out_ch, int32_t depth, int32_t skip, const void* const* weights_t_bf16,
                       const void* w_out_t_bf16, const float* g_out, const float* g_scale, const void* relu_masks, void* dpre_bf16,
                       float* db_partial, int32_t fp16, riggs_stream stream) {
  RIGGS_REQUIRE(N >= 0 && depth >= 1 && depth <= 10, "MLP depth out of range");
  RIGGS_REQUIRE(out_ch >= 1 && out_ch <= 32, "MLP output width must be <= 32");
  if (N == 0) return 0;
  MlpBwdDesc d;
  d.N = N; d.out_ch = out_ch; d.depth = depth; d.skip = skip;
  for (int l = 0; l < depth; l++) { d.Wt[l] = (const unsigned short*)weights_t_bf16[l]; RIGGS_REQUIRE(l == 0 || d.Wt[l], "MLP transposed weights"); }
  d.Wout_t = (const unsigned short*)w_out_t_bf16;
  RIGGS_REQUIRE(d.Wout_t && g_out && relu_masks && dpre_bf16 && db_partial, "MLP backward pointers");
  if (fp16) hipLaunchKernelGGL((mlp_backward_kernel<MLP_RT, true>), dim3((N + 127) / 128), dim3(256), 0, (hipStream_t)stream, d, g_out,
                               g_scale, (const uint4*)relu_masks, (unsigned short*)dpre_bf16, db_partial);
  else hipLaunchKernelGGL((mlp_backward_kernel<MLP_RT, false>), dim3((N + 127) / 128), dim3(256), 0, (hipStream_t)stream, d, g_out,
                          g_scale, (const uint4*)relu_masks, (unsigned short*)dpre_bf16, db_partial);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_mlp_embed(int32_t N, int32_t multires, int32_t n_tail, const float* x, const float* tail, void* out_bf16,
                    int32_t fp16, riggs_stream stream) {
  RIGGS_REQUIRE(N >= 0 && multires >= 0 && n_tail >= 0, "MLP embedding arguments");
  const int in_ch = 3 * (1 + 2 * multires) + n_tail, in_pad = (in_ch + 31) & ~31;
  RIGGS_REQUIRE(in_ch <= MLP_MAX_IN, "MLP input width must be <= 128");
  const int n_rows = (N + 127) / 128 * 128;
  if (n_rows == 0) return 0;
  RIGGS_REQUIRE(x && out_bf16 && (n_tail == 0 || tail), "MLP embedding pointers");
  const size_t n_thr = (size_t)n_rows * (in_pad >> 3);
  if (fp16) hipLaunchKernelGGL(mlp_embed_kernel<true>, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, n_rows,
                               multires, n_tail, in_pad, x, tail, (unsigned short*)out_bf16);
  else hipLaunchKernelGGL(mlp_embed_kernel<false>, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, n_rows,
                          multires, n_tail, in_pad, x, tail, (unsigned short*)out_bf16);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_mlp_pack(int32_t in_ch, int32_t out_ch, int32_t depth, int32_t skip, const float* const* weights, const float* w_out,
                   void* const* weights_bf16, void* const* weights_t_bf16, void* w_out_bf16, void* w_out_t_bf16,
                   int32_t fp16, riggs_stream stream) {
  RIGGS_REQUIRE(depth >= 1 && depth <= 10 && in_ch >= 1 && in_ch <= MLP_MAX_IN && out_ch >= 1 && out_ch <= 32 && skip >= 0 &&
                skip < depth - 1, "MLP shape out of range");
  MlpPackDesc d;
  d.in_ch = in_ch; d.in_pad = (in_ch + 31) & ~31; d.out_ch = out_ch; d.depth = depth; d.skip = skip;
  for (int l = 0; l < depth; l++) {
    d.W[l] = weights[l]; d.Wp[l] = (unsigned short*)weights_bf16[l]; d.Wt[l] = (unsigned short*)weights_t_bf16[l];
    RIGGS_REQUIRE(d.W[l] && d.Wp[l] && (l == 0 || d.Wt[l]), "MLP pack pointers");
  }
  d.Wout = w_out; d.Wout_p = (unsigned short*)w_out_bf16; d.Wout_t = (unsigned short*)w_out_t_bf16;
  RIGGS_REQUIRE(d.Wout && d.Wout_p && d.Wout_t, "MLP pack head pointers");
  if (fp16) hipLaunchKernelGGL(mlp_pack_kernel<true>, dim3(32, depth + 1), dim3(256), 0, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(mlp_pack_kernel<false>, dim3(32, depth + 1), dim3(256), 0, (hipStream_t)stream, d);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

/* rows per workgroup of the MLP kernels = rows that one slice of riggs_mlp_backward's db_partial covers */
int32_t riggs_mlp_rows_per_workgroup(void) { return 32 * MLP_RT; }

}  // extern "C"
