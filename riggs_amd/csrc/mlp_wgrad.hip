// Weight gradients of the per-Gaussian MLP heads (mlp.hip):  dW_l = dpre_l^T · input_l  over all N Gaussians, every layer of one
// MLP in ONE launch, followed by one launch that sums the split partials into the parameters' gradient tensors.
//
// The products are HBM-bound by a wide margin: a hidden layer reads 2 x N x 512 B for 2 x 256 x 256 x N flops — per CU and
// 32 Gaussians that is 32 KB (≈ 4000 shader cycles at the chip's streaming rate) against 1024 cycles of MFMA work per SIMD.
// So the kernel is built around the stream, not the matrix pipe:
//   * a workgroup (8 waves) owns ONE product and a contiguous range of Gaussians, keeps the whole (M x K) fp32 result in its
//     accumulators and writes it once (split partial);  256 workgroups = one per CU, splits handed out by bytes per Gaussian;
//   * the operands arrive by LDS-direct loads (global_load_lds_dwordx4: no registers, no LDS-write pass) into a ring of four
//     32-Gaussian stages, three stages (96 KB per CU, 24 MB over the chip) in flight, waited for with a COUNTED s_waitcnt
//     vmcnt behind a raw s_barrier — every load instruction is unconditional (rows past N read a zero page, short stages are
//     padded with dummy loads) so that the count is exact for every wave;
//   * both operands are stored row-major (Gaussian, feature) while the MFMA wants 8 consecutive GAUSSIANS of one feature per
//     lane (the reduction runs over Gaussians): the fragments come out of LDS through the transposing read
//     (ds_read_b64_tr_b16, two per fragment; as 16-bit reads they were eight, and the kernel ran at 4.6 TB/s).
// Job classes (per-wave tiles TM x TK of 32 x 32, waves WM x WK):  hidden 256 x 256 (2x4, 4x2) · embedding 256 x 64 / 256 x 128
// (1x2 / 1x4, 8x1) · head 32 x 256 (1x1, 1x8).  A product's workgroups also sum the columns of its gradient operand — the layer's
// bias gradient — from the staged rows (thread = column).
#include "common.h"

namespace riggs {

typedef short wg_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 wg_f8 __attribute__((ext_vector_type(8)));
typedef float wg_acc __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* wg_gptr;
typedef __attribute__((address_space(3))) void* wg_lptr;

#define WG_ROWS 32      // Gaussians per stage (two MFMA K-steps of 16)
#define WG_STAGES 4
#define WG_WAVES 8
#define WG_MAX_JOBS 12
#define WG_DUMMY 1024   // bytes of LDS behind the ring that the padding loads land in
#define WG_ZERO_BYTES 1024

enum { WG_CLS_HIDDEN = 0, WG_CLS_EMB64 = 1, WG_CLS_EMB128 = 2, WG_CLS_HEAD = 3 };

struct WgJob {
  const unsigned short* d;  // (rows, M) 16-bit: the gradient operand (M = its columns)
  const unsigned short* a;  // (rows, K) 16-bit: the layer's input
  float* part;              // (splits, M, K) fp32
  float* colsum;            // (splits, M) fp32 or NULL: the column sums of d (the layer's bias gradient)
  int cls, first_wg, splits, pad;
};
struct WgDesc {
  int N, steps, njobs, pad;       // N, steps: the host's row count (the buffers' capacity when n_dev is given)
  const int32_t* n_dev;           // NULL, or a device word: only rows < min(*n_dev, N) exist (row-sparse backward: mlp.hip)
  const unsigned short* zeros;  // WG_ZERO_BYTES of zeros (source of the rows past N and of the padding loads)
  WgJob job[WG_MAX_JOBS];
};

template <bool H16> __device__ __forceinline__ wg_acc wg_mfma(wg_h8 a, wg_h8 b, wg_acc c) {
  if constexpr (H16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wg_f8, a), __builtin_bit_cast(wg_f8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool H16> __device__ __forceinline__ float wg_h2f(unsigned short h) {
  if constexpr (H16) return (float)__builtin_bit_cast(_Float16, h);
  else return __uint_as_float((uint32_t)h << 16);
}

// The LDS image of a stage is row-major (Gaussian, feature) — what an LDS-direct load writes — and read with gfx950's
// transposing LDS read (ds_read_b64_tr_b16): a 16-lane group hands in the addresses of a [4 Gaussians][16 features] block (four
// lanes per row, 8 bytes each) and every lane receives one feature's four consecutive GAUSSIANS — half an MFMA fragment.  Four
// rows of one block are RB bytes apart, i.e. on the same banks for RB >= 256: the 16-byte granules of row n are stored XOR-ed
// with wg_swz(n) (applied to the SOURCE address of the LDS-direct load: its LDS side is lane-linear), which puts the four rows
// of a 32-lane access on four different 64-byte quarters of the bank row.
template <int RB> __device__ __forceinline__ int wg_swz(int n) {
  if constexpr (RB >= 256) return (n & 3) << 2;
  else if constexpr (RB == 128) return ((n >> 1) & 1) << 2;
  else return 0;
}
// (Issued as inline asm: behind the builtin the compiler puts s_waitcnt vmcnt(0) in front of the stage's first read — an LDS
// read it cannot tell apart from the LDS-direct loads in flight — which drains the ring every stage.  The waits are then ours:
// one lgkmcnt(0) behind a K-step's reads, every fragment register passed through an empty asm behind it so that no MFMA can
// be scheduled above the wait.)
typedef short wg_h4 __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ wg_h4 wg_tr(uint32_t lds_addr) {
  wg_h4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t wg_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(const char*)p;
}
// one K-step (16 Gaussians, the KB-th of the stage): fragments of TM x TK tiles, then the products
template <int TM, int TK, int M, int K, int KB, bool H16>
__device__ __forceinline__ void wg_kstep(wg_acc (&acc)[TM][TK], const uint32_t (&adD)[TM], const uint32_t (&adA)[TK], uint32_t stage_d,
                                         uint32_t stage_a) {
  wg_h4 alo[TM], ahi[TM], blo[TK], bhi[TK];
#pragma unroll
  for (int i = 0; i < TM; i++) {
    alo[i] = wg_tr<KB * 16 * M * 2>(stage_d + adD[i]);
    ahi[i] = wg_tr<KB * 16 * M * 2 + 4 * M * 2>(stage_d + adD[i]);
  }
#pragma unroll
  for (int i = 0; i < TK; i++) {
    blo[i] = wg_tr<KB * 16 * K * 2>(stage_a + adA[i]);
    bhi[i] = wg_tr<KB * 16 * K * 2 + 4 * K * 2>(stage_a + adA[i]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wg_h8 af[TM], bf[TK];
#pragma unroll
  for (int i = 0; i < TM; i++) {
    asm volatile("" : "+v"(alo[i]), "+v"(ahi[i]));
    af[i] = __builtin_shufflevector(alo[i], ahi[i], 0, 1, 2, 3, 4, 5, 6, 7);
  }
#pragma unroll
  for (int i = 0; i < TK; i++) {
    asm volatile("" : "+v"(blo[i]), "+v"(bhi[i]));
    bf[i] = __builtin_shufflevector(blo[i], bhi[i], 0, 1, 2, 3, 4, 5, 6, 7);
  }
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TK; j++) acc[i][j] = wg_mfma<H16>(af[i], bf[j], acc[i][j]);
}

template <int N> __device__ __forceinline__ void wg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One workgroup's product: stages [t0, t1) of job `jb`.
template <int TM, int TK, int WM, int WK, bool H16>
__device__ __forceinline__ void wg_body(const WgDesc& D, const WgJob& jb, int split, int t0, int t1, int n_rows, char* lds) {
  constexpr int M = 32 * TM * WM, K = 32 * TK * WK;
  constexpr int DB = WG_ROWS * M * 2, AB = WG_ROWS * K * 2, STAGE = DB + AB;      // bytes
  constexpr int RBD = 2 * M, RBA = 2 * K;  // bytes per row of the two images
  constexpr int CD = DB / 1024, CH = (DB + AB) / 1024, PER = (CH + WG_WAVES - 1) / WG_WAVES;
  static_assert(WM * WK == WG_WAVES, "eight waves");
  static_assert(WG_STAGES * STAGE + WG_DUMMY <= 160 * 1024, "LDS");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WK, wk = wave - wm * WK;
  const int r = lane & 31, kg = lane >> 5;
  // (everything the loop needs out of the descriptor goes into registers HERE: the waits below are asm statements with a
  // memory clobber, behind which the compiler would re-load descriptor fields — an ordinary load inside the loop makes it wait
  // vmcnt(0), draining the ring)
  const char* dz = reinterpret_cast<const char*>(D.zeros) + lane * 16;
  const char* jd = reinterpret_cast<const char*>(jb.d);
  const char* ja = reinterpret_cast<const char*>(jb.a);
  float* const jcol = jb.colsum;
  float* const jpart = jb.part;
  const uint32_t lds0 = wg_lds_addr(lds);

  auto issue = [&](int buf, int t) {
    char* base = lds + buf * STAGE;
    const int row0 = t * WG_ROWS;
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const int c = i * WG_WAVES + wave;  // 1 KB piece of the stage (wave-uniform)
      const char* src = dz;
      char* dst = lds + WG_STAGES * STAGE;
      if (c < CD) {
        const int byte = c * 1024 + lane * 16, rl = byte / RBD, lg = ((byte % RBD) >> 4) ^ wg_swz<RBD>(rl);
        if (row0 + rl < n_rows) src = jd + (size_t)(row0 + rl) * RBD + lg * 16;
        dst = base + c * 1024;
      } else if (c < CH) {
        const int byte = (c - CD) * 1024 + lane * 16, rl = byte / RBA, lg = ((byte % RBA) >> 4) ^ wg_swz<RBA>(rl);
        if (row0 + rl < n_rows) src = ja + (size_t)(row0 + rl) * RBA + lg * 16;
        dst = base + c * 1024;
      }
      __builtin_amdgcn_global_load_lds((wg_gptr)src, (wg_lptr)dst, 16, 0, 0);
    }
  };

  wg_acc acc[TM][TK];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TK; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  float colsum = 0.f;
  const bool want_colsum = jcol != nullptr && tid < M;
  // this lane's place in the transposing reads: row (of the 8 of its K-half: kg * 8 + (q >> 2), + 4 for the second read) and
  // column 16 (g & 1) + 4 (q & 3) of a 32-column tile; the tile's granules XOR-ed like the row they are in
  uint32_t offD[TM], offA[TK];  // bytes
  {
    const int g = lane >> 4, q = lane & 15;
    const int nl = kg * 8 + (q >> 2), cl = 16 * (g & 1) + 4 * (q & 3);
#pragma unroll
    for (int i = 0; i < TM; i++) offD[i] = 2u * (uint32_t)(nl * M + ((((wm * TM + i) * 4 + (cl >> 3)) ^ wg_swz<RBD>(nl)) << 3) + (cl & 7));
#pragma unroll
    for (int i = 0; i < TK; i++) offA[i] = 2u * (uint32_t)(nl * K + ((((wk * TK + i) * 4 + (cl >> 3)) ^ wg_swz<RBA>(nl)) << 3) + (cl & 7));
  }

  const int ns = t1 - t0;
#pragma unroll
  for (int p = 0; p < WG_STAGES - 1; p++)
    if (p < ns) issue(p, t0 + p);
  for (int s = 0; s < ns; s++) {
    const int pending = min(WG_STAGES - 2, ns - 1 - s);  // stages issued behind this one
    if (pending >= 2) wg_wait_vm<2 * PER>();
    else if (pending == 1) wg_wait_vm<PER>();
    else wg_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the reads of the stage about to be overwritten have returned)
    __builtin_amdgcn_s_barrier();
    if (s + WG_STAGES - 1 < ns) issue((s + WG_STAGES - 1) % WG_STAGES, t0 + s + WG_STAGES - 1);
    const unsigned short* dI = reinterpret_cast<const unsigned short*>(lds + (s % WG_STAGES) * STAGE);
    const uint32_t st_d = lds0 + (uint32_t)((s % WG_STAGES) * STAGE), st_a = st_d + DB;
    wg_kstep<TM, TK, M, K, 0, H16>(acc, offD, offA, st_d, st_a);
    wg_kstep<TM, TK, M, K, 1, H16>(acc, offD, offA, st_d, st_a);
    if (want_colsum) {  // (thread = column of d; consecutive lanes read consecutive 16-bit values: conflict-free)
#pragma unroll 8
      for (int n = 0; n < WG_ROWS; n++) colsum += wg_h2f<H16>(dI[n * M + (((tid >> 3) ^ wg_swz<RBD>(n)) << 3) + (tid & 7)]);
    }
  }
  float* part = jpart + (size_t)split * (M * K);
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TK; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg, k = (wk * TK + j) * 32 + r;
        part[(size_t)m * K + k] = acc[i][j][e];
      }
  if (want_colsum) jcol[(size_t)split * M + tid] = colsum;
}

template <bool H16>
__global__ __launch_bounds__(512) void mlp_wgrad_kernel(WgDesc D) {
  extern __shared__ __attribute__((aligned(1024))) char wg_lds[];
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < D.njobs && b >= D.job[j + 1].first_wg) j++;
  const WgJob jb = D.job[__builtin_amdgcn_readfirstlane(j)];
  const int split = b - jb.first_wg;
  // (the row count may live on the device: read it once, here — an ordinary load inside wg_body's loop would drain its ring)
  int n_rows = D.N, steps = D.steps;
  if (D.n_dev) {
    n_rows = __builtin_amdgcn_readfirstlane(min(D.n_dev[0], D.N));
    steps = (n_rows + WG_ROWS - 1) / WG_ROWS;
  }
  const int t0 = (int)((long long)steps * split / jb.splits), t1 = (int)((long long)steps * (split + 1) / jb.splits);
  switch (jb.cls) {
    case WG_CLS_HIDDEN: wg_body<2, 4, 4, 2, H16>(D, jb, split, t0, t1, n_rows, wg_lds); break;
    case WG_CLS_EMB64: wg_body<1, 2, 8, 1, H16>(D, jb, split, t0, t1, n_rows, wg_lds); break;
    case WG_CLS_EMB128: wg_body<1, 4, 8, 1, H16>(D, jb, split, t0, t1, n_rows, wg_lds); break;
    default: wg_body<1, 1, 1, 8, H16>(D, jb, split, t0, t1, n_rows, wg_lds); break;
  }
}

// g_out (N, out_ch) fp32 x scale -> the head product's gradient operand (rows rounded up to 32, 32 columns, zero padded) in the
// 16-bit format; block 0 also clears the zero page
template <bool H16>
__global__ __launch_bounds__(256) void mlp_gob_kernel(int N, int rows, int out_ch, const float* __restrict__ g_out,
                                                      const float* __restrict__ g_scale, unsigned short* __restrict__ gob,
                                                      uint32_t* __restrict__ zeros, const int32_t* __restrict__ n_dev) {
  const float gs = g_scale ? g_scale[0] : 1.0f;
  if (n_dev) N = min(n_dev[0], N);
  if (blockIdx.x == 0 && threadIdx.x < WG_ZERO_BYTES / 4) zeros[threadIdx.x] = 0u;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;  // eight values (16 bytes) per thread
  if (e >= (size_t)rows * 4) return;
  const int n = (int)(e >> 2), c0 = (int)(e & 3) * 8;
  wg_h8 v;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    float f = 0.f;
    if (n < N && c0 + j < out_ch) f = g_out[(size_t)n * out_ch + c0 + j] * gs;
    unsigned short h;
    if constexpr (H16) h = __builtin_bit_cast(unsigned short, (_Float16)f);
    else { uint32_t u = __float_as_uint(f); u += 0x7FFFu + ((u >> 16) & 1u); h = (unsigned short)(u >> 16); }
    v[j] = (short)h;
  }
  *reinterpret_cast<wg_h8*>(gob + e * 8) = v;
}

// max |g| over a tensor -> the power of two that lifts it to ~2^10 (riggs_mlp_grad_scale): the maximum is taken on the bit
// patterns (non-negative floats order like unsigned integers) into a word that is zero between calls; a second, one-thread launch
// turns it into the scale and clears the word again
__global__ __launch_bounds__(256) void mlp_amax_kernel(int64_t n, const float* __restrict__ g, uint32_t* __restrict__ word) {
  uint32_t m = 0u;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const uint4 v = reinterpret_cast<const uint4*>(g)[i];
    m = max(max(m, v.x & 0x7FFFFFFFu), max(max(v.y & 0x7FFFFFFFu, v.z & 0x7FFFFFFFu), v.w & 0x7FFFFFFFu));
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) m = max(m, __float_as_uint(g[(n4 << 2) + threadIdx.x]) & 0x7FFFFFFFu);
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  // ONE atomic per workgroup, a few hundred workgroups: atomics on one address are served one after the other at the memory
  // side (one per wave of 2 048 workgroups were 8 192 of them: 70 us for a 6 us read)
  __shared__ uint32_t s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
    if (m) atomicMax(word, m);
  }
}
__global__ void mlp_scale_kernel(uint32_t* __restrict__ word, float* __restrict__ scale) {
  const float amax = fmaxf(__uint_as_float(word[0]), 1e-30f);
  scale[0] = exp2f(floorf(log2f(1024.0f / amax)));
  word[0] = 0u;
}


// The DeformMLP's L2 regulariser folded into its backward (train_rig.py:446-454: lambda * mean(template_offsets^2) over ALL
// Gaussians, x1e3 on the template frame): the cotangent the data-gradient pass reads becomes  g_eff = g + coef * out  with
// coef = 2 lambda / (3 N) a device scalar — written by the launch that takes max|g| for the fp16 gradient scale anyway, so the
// term costs no launch of its own — and the launch that turns the maximum into the scale also finishes  mean(out^2)  (the value
// the reference logs) from the per-workgroup partial sums, in a fixed order.
// (generalised in round 6 — riggs_mlp_cotangent: the cotangent may also be the SUM of a direct term and a per-row-masked one,
//   g_eff = (g + g_rows * row_mask[row]) * [s (1 - s)] + coef * out ,
// every piece optional: g_rows * row_mask is what reaches the template offsets through  d_xyz = blend + offsets * motion_mask
// (skeleton_warp.py:152-161; was a torch mul in front), s (1 - s) the sigmoid folded into the WeightMLP's head (was a
// sigmoid_backward launch).)
struct MlpCot {
  int64_t n;
  int out_ch;
  const float *g, *g_rows, *row_mask, *sig, *out, *coef;
  float* g_eff;
};
__device__ __forceinline__ float mlp_cot_value(const MlpCot& c, int64_t i, float coef) {
  float v = c.g ? c.g[i] : 0.f;
  if (c.g_rows) v += c.row_mask ? c.g_rows[i] * c.row_mask[(uint32_t)i / (uint32_t)c.out_ch] : c.g_rows[i];
  if (c.sig) { const float sg = c.sig[i]; v *= sg * (1.0f - sg); }
  if (c.out) v += coef * c.out[i];
  return v;
}
__global__ __launch_bounds__(256) void mlp_l2_amax_kernel(MlpCot c, uint32_t* __restrict__ word, float* __restrict__ partials) {
  const float coef = c.coef ? c.coef[0] : 0.f;
  uint32_t m = 0u;
  float ss = 0.f;
  const int64_t n4 = c.n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 e;
    if (!c.g_rows && !c.sig && c.g && c.out) {  // the plain L2 form: whole vectors
      const float4 gv = reinterpret_cast<const float4*>(c.g)[i], ov = reinterpret_cast<const float4*>(c.out)[i];
      e.x = gv.x + coef * ov.x; e.y = gv.y + coef * ov.y; e.z = gv.z + coef * ov.z; e.w = gv.w + coef * ov.w;
    } else {
      e.x = mlp_cot_value(c, 4 * i, coef); e.y = mlp_cot_value(c, 4 * i + 1, coef);
      e.z = mlp_cot_value(c, 4 * i + 2, coef); e.w = mlp_cot_value(c, 4 * i + 3, coef);
    }
    reinterpret_cast<float4*>(c.g_eff)[i] = e;
    m = max(max(m, __float_as_uint(e.x) & 0x7FFFFFFFu), max(max(__float_as_uint(e.y) & 0x7FFFFFFFu, __float_as_uint(e.z) & 0x7FFFFFFFu),
                                                             __float_as_uint(e.w) & 0x7FFFFFFFu));
    if (c.out) {
      const float4 ov = reinterpret_cast<const float4*>(c.out)[i];
      ss += (ov.x * ov.x + ov.y * ov.y) + (ov.z * ov.z + ov.w * ov.w);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(c.n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float e = mlp_cot_value(c, i, coef);
    c.g_eff[i] = e;
    m = max(m, __float_as_uint(e) & 0x7FFFFFFFu);
    if (c.out) ss += c.out[i] * c.out[i];
  }
  for (int o = 32; o > 0; o >>= 1) { m = max(m, (uint32_t)__shfl_xor((int)m, o)); ss += __shfl_xor(ss, o); }
  __shared__ uint32_t s_m[4];
  __shared__ float s_s[4];
  if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_s[threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
    if (m) atomicMax(word, m);
    partials[blockIdx.x] = (s_s[0] + s_s[1]) + (s_s[2] + s_s[3]);
  }
}
__global__ __launch_bounds__(64) void mlp_l2_scale_kernel(uint32_t* __restrict__ word, float* __restrict__ scale,
                                                          const float* __restrict__ partials, int n_partials, float inv_count,
                                                          float* __restrict__ mean_sq) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n_partials; i += 64) s += partials[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) {
    const float amax = fmaxf(__uint_as_float(word[0]), 1e-30f);
    scale[0] = exp2f(floorf(log2f(1024.0f / amax)));
    word[0] = 0u;
    if (mean_sq) mean_sq[0] = s * inv_count;
  }
}

struct WgOut {
  const float* part;  // (splits, M, K)
  float* dst;         // row-major, leading dimension ld, first column col_off
  int splits, M, K, rows, cols, ld, col_off, pad;
};
struct WgBias {
  const float* part;  // (splits, M)
  float* dst;         // (count)
  int splits, M, count, pad;
};
// the masters' columns of a constant input tail (riggs_mlp_wgrad_tail): dW[n][col_off + k] = db[n] * tail[k] — the row sums the
// product's workgroups left beside their partials, times the tail
struct WgRank1 {
  const float* part;  // (splits, M): the column sums of the layer's gradient operand
  float* dst;         // the layer's weight gradient, leading dimension ld
  int splits, M, ld, col_off;
};
struct WgReduceDesc {
  int n, nb, nr1, tail_ch;
  const float* g_scale;  // the gradients are divided by it (a power of two: exact)
  const float* tail;
  WgOut o[WG_MAX_JOBS];
  WgBias b[WG_MAX_JOBS];
  WgRank1 r1[2];
};
__global__ __launch_bounds__(256) void mlp_wgrad_reduce_kernel(WgReduceDesc D) {
  const float inv = D.g_scale ? 1.0f / D.g_scale[0] : 1.0f;
  const int j = blockIdx.y;
  if (j == D.n + 1) {  // the constant tail's columns: a wave per row (256 rows over 64 workgroups of four waves)
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int q1 = 0; q1 < D.nr1; q1++) {
      const WgRank1& r = D.r1[q1];
      float db = 0.f;
      for (int q = 0; q < r.splits; q++) db += r.part[(size_t)q * r.M + n];  // (the order of the bias gradient's own sum)
      db *= inv;
      for (int k = lane; k < D.tail_ch; k += 64) r.dst[(size_t)n * r.ld + r.col_off + k] = db * D.tail[k];
    }
    return;
  }
  if (j == D.n) {  // the bias gradients: blockIdx.x = which one, thread = column
    if ((int)blockIdx.x >= D.nb) return;
    const WgBias& bb = D.b[blockIdx.x];
    const int c = threadIdx.x;
    if (c < bb.count) {
      float s = 0.f;
      for (int q = 0; q < bb.splits; q++) s += bb.part[(size_t)q * bb.M + c];
      bb.dst[c] = s * inv;
    }
    return;
  }
  const WgOut& o = D.o[j];
  const int total = o.M * o.K;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int r = e / o.K, c = e - r * o.K;
    if (r >= o.rows || c >= o.cols) continue;
    const float* p = o.part + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int q = 0;
    for (; q + 4 <= o.splits; q += 4) {
      s0 += p[(size_t)q * total];
      s1 += p[(size_t)(q + 1) * total];
      s2 += p[(size_t)(q + 2) * total];
      s3 += p[(size_t)(q + 3) * total];
    }
    for (; q < o.splits; q++) s0 += p[(size_t)q * total];
    o.dst[(size_t)r * o.ld + o.col_off + c] = ((s0 + s1) + (s2 + s3)) * inv;
  }
}

// ---- the plan: which products, how many splits each, where their partials live in the workspace
struct WgPlan {
  int njobs, steps, rows32, total_wg;
  int cls[WG_MAX_JOBS], splits[WG_MAX_JOBS], first[WG_MAX_JOBS], M[WG_MAX_JOBS], K[WG_MAX_JOBS];
  int layer[WG_MAX_JOBS];      // hidden / embedding: the layer; head: depth
  size_t part_off[WG_MAX_JOBS], colsum_off[WG_MAX_JOBS];  // floats
  size_t gob_off /* bytes */, zeros_off /* bytes */, bytes;
};
static int wg_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
  int& c = cus[dev & 63];
  if (c == 0 && (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0)) { (void)hipGetLastError(); c = 256; }
  return c;
}
static void wg_plan(WgPlan& P, int N, int in_ch, int depth, int skip, int cus) {
  const int in_pad = (in_ch + 63) & ~63;
  P.steps = (N + WG_ROWS - 1) / WG_ROWS;
  P.rows32 = P.steps * WG_ROWS;
  int n = 0;
  for (int l = 1; l < depth; l++) { P.cls[n] = WG_CLS_HIDDEN; P.M[n] = 256; P.K[n] = 256; P.layer[n++] = l; }
  const int ecls = in_pad == 64 ? WG_CLS_EMB64 : WG_CLS_EMB128;
  P.cls[n] = ecls; P.M[n] = 256; P.K[n] = in_pad; P.layer[n++] = 0;
  P.cls[n] = ecls; P.M[n] = 256; P.K[n] = in_pad; P.layer[n++] = skip + 1;
  P.cls[n] = WG_CLS_HEAD; P.M[n] = 32; P.K[n] = 256; P.layer[n++] = depth;
  P.njobs = n;
  // splits by bytes per Gaussian (the head's a little more: its stage is padded with dummy loads), one workgroup per CU
  double w[WG_MAX_JOBS], tot = 0.0;
  for (int j = 0; j < n; j++) { w[j] = 2.0 * (P.M[j] + P.K[j]) * (P.cls[j] == WG_CLS_HIDDEN ? 1.0 : 1.25); tot += w[j]; }
  const int target = cus > n ? cus : n;
  int used = 0;
  for (int j = 0; j < n; j++) {
    int s = (int)(target * w[j] / tot);
    if (s < 1) s = 1;
    if (s > P.steps) s = P.steps > 0 ? P.steps : 1;
    P.splits[j] = s;
    used += s;
  }
  for (int j = 0; used < target && j < n; j = (j + 1) % n) {  // leftovers to the big products first
    if (P.splits[j] < P.steps) { P.splits[j]++; used++; }
    else {
      bool any = false;
      for (int q = 0; q < n; q++) any |= P.splits[q] < P.steps;
      if (!any) break;
    }
  }
  size_t off = 0;
  int first = 0;
  for (int j = 0; j < n; j++) {
    P.first[j] = first; first += P.splits[j];
    P.part_off[j] = off; off += (size_t)P.splits[j] * P.M[j] * P.K[j];
  }
  P.total_wg = first;
  for (int j = 0; j < n; j++) { P.colsum_off[j] = off; off += (size_t)P.splits[j] * P.M[j]; }
  size_t bytes = (off * 4 + 1023) & ~(size_t)1023;
  P.gob_off = bytes; bytes += (size_t)P.rows32 * 32 * 2;
  bytes = (bytes + 1023) & ~(size_t)1023;
  P.zeros_off = bytes; bytes += WG_ZERO_BYTES;
  P.bytes = bytes;
}

}  // namespace riggs

using namespace riggs;

extern "C" {

int riggs_mlp_grad_scale(int64_t n, const float* g, float* scale, uint32_t* zero_word, riggs_stream stream) {
  RIGGS_REQUIRE(n >= 0 && scale && zero_word && (n == 0 || g), "riggs_mlp_grad_scale: bad arguments");
  RIGGS_REQUIRE(((uintptr_t)g & 15) == 0, "riggs_mlp_grad_scale: the tensor must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (n > 0) {
    const int64_t want = ((n >> 2) + 255) / 256;
    hipLaunchKernelGGL(mlp_amax_kernel, dim3((unsigned)(want < 512 ? (want > 0 ? want : 1) : 512)), dim3(256), 0, s, n, g, zero_word);
    RIGGS_HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(mlp_scale_kernel, dim3(1), dim3(1), 0, s, zero_word, scale);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}


int riggs_mlp_cotangent(int32_t N, int32_t out_ch, const float* g, const float* g_rows, const float* row_mask, const float* sigmoid_out,
                        const float* l2_out, const float* l2_coef, float* g_eff, float* scale, uint32_t* zero_word, float* partials512,
                        float* mean_sq, riggs_stream stream) {
  RIGGS_REQUIRE(N >= 0 && out_ch >= 1 && out_ch <= 32 && scale && zero_word && partials512, "riggs_mlp_cotangent: bad arguments");
  RIGGS_REQUIRE((int64_t)N * out_ch < (1ll << 31), "riggs_mlp_cotangent: N x out_ch must be < 2^31");
  RIGGS_REQUIRE(N == 0 || ((g || g_rows) && g_eff), "riggs_mlp_cotangent: tensors");
  RIGGS_REQUIRE((l2_out == nullptr) == (l2_coef == nullptr), "riggs_mlp_cotangent: l2_out and l2_coef come together");
  RIGGS_REQUIRE(row_mask == nullptr || g_rows != nullptr, "riggs_mlp_cotangent: row_mask without g_rows");
  RIGGS_REQUIRE((((uintptr_t)g | (uintptr_t)l2_out | (uintptr_t)g_eff) & 15) == 0, "riggs_mlp_cotangent: g, l2_out and g_eff must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)N * out_ch;
  int blocks = 0;
  if (n > 0) {
    const int64_t want = ((n >> 2) + 255) / 256;
    blocks = (int)(want < 512 ? (want > 0 ? want : 1) : 512);
    MlpCot c;
    c.n = n; c.out_ch = out_ch; c.g = g; c.g_rows = g_rows; c.row_mask = row_mask; c.sig = sigmoid_out; c.out = l2_out; c.coef = l2_coef;
    c.g_eff = g_eff;
    hipLaunchKernelGGL(mlp_l2_amax_kernel, dim3(blocks), dim3(256), 0, s, c, zero_word, partials512);
    RIGGS_HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(mlp_l2_scale_kernel, dim3(1), dim3(64), 0, s, zero_word, scale, partials512, blocks, n > 0 ? 1.0f / (float)n : 0.f,
                     l2_out ? mean_sq : nullptr);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_mlp_l2_grad_scale(int64_t n, const float* g, const float* out, const float* coef, float* g_eff, float* scale,
                            uint32_t* zero_word, float* partials512, float* mean_sq, riggs_stream stream) {
  RIGGS_REQUIRE(n >= 0 && scale && zero_word && partials512 && coef, "riggs_mlp_l2_grad_scale: bad arguments");
  RIGGS_REQUIRE(n == 0 || (g && out && g_eff), "riggs_mlp_l2_grad_scale: tensors");
  RIGGS_REQUIRE((((uintptr_t)g | (uintptr_t)out | (uintptr_t)g_eff) & 15) == 0, "riggs_mlp_l2_grad_scale: the tensors must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  int blocks = 0;
  if (n > 0) {
    const int64_t want = ((n >> 2) + 255) / 256;
    blocks = (int)(want < 512 ? (want > 0 ? want : 1) : 512);
    MlpCot c;
    c.n = n; c.out_ch = 1; c.g = g; c.g_rows = nullptr; c.row_mask = nullptr; c.sig = nullptr; c.out = out; c.coef = coef; c.g_eff = g_eff;
    hipLaunchKernelGGL(mlp_l2_amax_kernel, dim3(blocks), dim3(256), 0, s, c, zero_word, partials512);
    RIGGS_HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(mlp_l2_scale_kernel, dim3(1), dim3(64), 0, s, zero_word, scale, partials512, blocks, n > 0 ? 1.0f / (float)n : 0.f, mean_sq);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

size_t riggs_mlp_wgrad_workspace_bytes(int32_t N, int32_t in_ch, int32_t depth, int32_t skip) {
  if (N <= 0 || depth < 1 || depth > 10 || in_ch < 1 || in_ch > 128 || skip < 0 || skip >= depth - 1) return 0;
  WgPlan P;
  wg_plan(P, N, in_ch, depth, skip, wg_cus());
  return P.bytes;
}

int riggs_mlp_wgrad(int32_t N, int32_t in_ch, int32_t out_ch, int32_t depth, int32_t skip, const void* x_emb_bf16,
                    const void* acts_bf16, const void* dpre_bf16, const float* g_out, const float* g_scale, void* workspace,
                    size_t workspace_bytes, float* const* grad_weights, float* const* grad_biases, float* grad_w_out,
                    float* grad_b_out, const int32_t* n_rows_dev, int32_t fp16, riggs_stream stream) {
  return riggs_mlp_wgrad_tail(N, in_ch, 0, nullptr, out_ch, depth, skip, x_emb_bf16, acts_bf16, dpre_bf16, g_out, g_scale, workspace,
                              workspace_bytes, grad_weights, grad_biases, grad_w_out, grad_b_out, n_rows_dev, fp16, stream);
}

int riggs_mlp_wgrad_tail(int32_t N, int32_t in_ch, int32_t tail_ch, const float* tail, int32_t out_ch, int32_t depth, int32_t skip,
                         const void* x_emb_bf16, const void* acts_bf16, const void* dpre_bf16, const float* g_out, const float* g_scale,
                         void* workspace, size_t workspace_bytes, float* const* grad_weights, float* const* grad_biases,
                         float* grad_w_out, float* grad_b_out, const int32_t* n_rows_dev, int32_t fp16, riggs_stream stream) {
  RIGGS_REQUIRE(N >= 0 && depth >= 2 && depth <= 10, "MLP depth out of range");
  RIGGS_REQUIRE(tail_ch >= 0 && tail_ch <= 4096 && (tail_ch == 0 || tail), "MLP weight gradients: the constant tail");
  const int in_true = in_ch + tail_ch;  // the masters' input columns
  RIGGS_REQUIRE(in_ch >= 1 && in_ch <= 128 && out_ch >= 1 && out_ch <= 32, "MLP width out of range");
  RIGGS_REQUIRE(skip >= 0 && skip < depth - 1, "MLP skip layer out of range");
  if (N == 0) return 0;
  RIGGS_REQUIRE(x_emb_bf16 && acts_bf16 && dpre_bf16 && g_out && workspace && grad_weights && grad_biases && grad_w_out && grad_b_out,
                "MLP weight-gradient pointers");
  RIGGS_REQUIRE(((uintptr_t)workspace & 255) == 0, "MLP weight-gradient workspace must be 256-byte aligned");
  WgPlan P;
  wg_plan(P, N, in_ch, depth, skip, wg_cus());
  RIGGS_REQUIRE(workspace_bytes >= P.bytes, "MLP weight-gradient workspace too small (riggs_mlp_wgrad_workspace_bytes)");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  float* parts = (float*)ws;
  unsigned short* gob = (unsigned short*)(ws + P.gob_off);
  const int in_pad = (in_ch + 63) & ~63;
  const unsigned short* xb = (const unsigned short*)x_emb_bf16;
  const unsigned short* acts = (const unsigned short*)acts_bf16;
  const unsigned short* dpre = (const unsigned short*)dpre_bf16;
  const size_t slab = (size_t)N * 256;

  {
    const size_t n_el = (size_t)P.rows32 * 4;
    if (fp16) hipLaunchKernelGGL(mlp_gob_kernel<true>, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, s, N, P.rows32, out_ch, g_out, g_scale,
                                 gob, (uint32_t*)(ws + P.zeros_off), n_rows_dev);
    else hipLaunchKernelGGL(mlp_gob_kernel<false>, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, s, N, P.rows32, out_ch, g_out, g_scale,
                            gob, (uint32_t*)(ws + P.zeros_off), n_rows_dev);
    RIGGS_HIP_CHECK(hipGetLastError());
  }
  WgDesc D;
  WgReduceDesc R;
  D.N = N; D.steps = P.steps; D.njobs = P.njobs; D.pad = 0; D.n_dev = n_rows_dev;
  D.zeros = (const unsigned short*)(ws + P.zeros_off);
  R.n = P.njobs; R.nb = 0; R.nr1 = 0; R.tail_ch = tail_ch; R.tail = tail; R.g_scale = g_scale;
  for (int j = 0; j < P.njobs; j++) {
    WgJob& jb = D.job[j];
    WgOut& o = R.o[j];
    const int l = P.layer[j];
    jb.cls = P.cls[j]; jb.first_wg = P.first[j]; jb.splits = P.splits[j]; jb.pad = 0;
    jb.part = parts + P.part_off[j];
    jb.colsum = parts + P.colsum_off[j];
    float* bias_dst = nullptr;
    int bias_count = 256;
    o.part = jb.part; o.splits = P.splits[j]; o.M = P.M[j]; o.K = P.K[j]; o.pad = 0;
    if (P.cls[j] == WG_CLS_HIDDEN) {
      jb.d = dpre + (size_t)l * slab; jb.a = acts + (size_t)(l - 1) * slab;
      RIGGS_REQUIRE(grad_weights[l], "MLP weight-gradient outputs");
      const bool sk = (l == skip + 1);
      o.dst = grad_weights[l]; o.rows = 256; o.cols = 256; o.ld = sk ? in_true + 256 : 256; o.col_off = sk ? in_true : 0;
      bias_dst = grad_biases[l];
      RIGGS_REQUIRE(bias_dst, "MLP bias-gradient outputs");
    } else if (P.cls[j] == WG_CLS_HEAD) {
      jb.d = gob; jb.a = acts + (size_t)(depth - 1) * slab;
      o.dst = grad_w_out; o.rows = out_ch; o.cols = 256; o.ld = 256; o.col_off = 0;
      bias_dst = grad_b_out; bias_count = out_ch;
    } else {
      jb.d = dpre + (size_t)l * slab; jb.a = xb;
      RIGGS_REQUIRE(grad_weights[l], "MLP weight-gradient outputs");
      o.dst = grad_weights[l]; o.rows = 256; o.cols = in_ch; o.ld = (l == 0) ? in_true : in_true + 256; o.col_off = 0;
      if (l == 0) { bias_dst = grad_biases[0]; RIGGS_REQUIRE(bias_dst, "MLP bias-gradient outputs"); }  // (layer skip + 1's comes from its hidden product)
      (void)in_pad;
    }
    if (bias_dst) {
      WgBias& bb = R.b[R.nb++];
      bb.part = jb.colsum; bb.dst = bias_dst; bb.splits = P.splits[j]; bb.M = P.M[j]; bb.count = bias_count; bb.pad = 0;
      if (tail_ch > 0 && P.cls[j] != WG_CLS_HEAD && (l == 0 || l == skip + 1)) {  // (the product whose row sums are the layer's bias gradient)
        RIGGS_REQUIRE(R.nr1 < 2, "MLP weight gradients: more than two layers read the input");
        WgRank1& r1 = R.r1[R.nr1++];
        r1.part = jb.colsum; r1.dst = grad_weights[l]; r1.splits = P.splits[j]; r1.M = P.M[j];
        r1.ld = (l == 0) ? in_true : in_true + 256; r1.col_off = in_ch;
      }
    } else jb.colsum = nullptr;
  }
  static unsigned long long attr_done = 0ull;
  const int lds_bytes = WG_STAGES * (WG_ROWS * 512 * 2) + WG_DUMMY;
  if (once_per_device(attr_done)) {
    RIGGS_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    RIGGS_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  }
  if (fp16) hipLaunchKernelGGL(mlp_wgrad_kernel<true>, dim3(P.total_wg), dim3(512), lds_bytes, s, D);
  else hipLaunchKernelGGL(mlp_wgrad_kernel<false>, dim3(P.total_wg), dim3(512), lds_bytes, s, D);
  RIGGS_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3(64, P.njobs + 1 + (R.nr1 > 0 ? 1 : 0)), dim3(256), 0, s, R);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
