// Per-Gaussian MLP heads (second half): WeightMLP / DeformMLP
// (skeleton_utils/network_utils.py:6-112) as ONE fused launch per direction on the CDNA4 matrix cores.
//
//   x_emb (N, in_ch) fp32 -> D x [Linear(256) + ReLU], the embedding re-concatenated in front of the hidden vector
//   after layer `skip` -> Linear(out_ch)
//
// bf16 operands, fp32 accumulation (v_mfma_f32_32x32x16_bf16): the reference computes in fp32, so this path is
// opt-in and its parity bar is the bf16 one ("parity tolerance must be renegotiated for bf16").
//
// Forward: a workgroup owns 128 rows (Gaussians).  The hidden vector of those rows lives in LDS as 16-bit
// [128][256 (+pad)]; a layer is  H^T <- relu(W H^T + b)  — the product is formed TRANSPOSED: the weights are the
// MFMA's A operand (rows = output neurons), the hidden vectors its B operand (columns = Gaussians).  Wave w computes
// neurons [64 w, 64 w + 64) of all 128 Gaussians: 2 x 4 tiles of 32 x 32, K in steps of 16.  The weights are stored
// FRAGMENT-MAJOR (riggs_mlp_pack): the 64 x 16 bytes a wave loads for (neuron tile, K-step) are one contiguous 1 KB run
// — eight cache lines per load instruction instead of a 16-byte piece of each of 32 weight rows (32 lines), which kept
// the texture path as busy as the matrix pipe; the hidden fragments (8 consecutive k of one Gaussian) are 16-byte LDS
// reads.  In the transposed accumulator a lane holds four CONSECUTIVE neurons of one Gaussian per register group, so the
// post-ReLU activations go back to LDS as 8-byte stores (32 per lane and layer instead of 128 two-byte ones), in place,
// behind a barrier, and — for the backward — to HBM as 16-bit values with full-line stores.
//
// Backward (data gradient): the same tiling with the transposed weights,  dH_{l-1} = (dH_l * relu'(H_l)) W_l ;
// every layer's masked gradient is stored as bf16 for the weight gradients (mlp_wgrad.hip: every product of one MLP in one
// launch, streamed at the HBM rate).  The inputs of both heads are detached in the
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
  const unsigned short* Wp[10];   // packed 16-bit weights of layer l, FRAGMENT-MAJOR (below): 8 neuron tiles x K_l / 16 steps x 512,
                                  // K_0 = in_pad, K_{skip+1} = in_pad + 256 (embedding columns first), else 256
  const float* bias[10];          // [256]
  const unsigned short* Wout;     // the head, fragment-major: 1 tile (outputs >= out_ch are zero) x 16 steps x 512
  const float* bout;              // [out_ch]
  const int32_t* n_dev;           // NULL, or a device word: only rows < min(*n_dev, N) exist (N stays the buffers' row stride)
  // output epilogue (riggs_mlp_epilogue): the head's value through a sigmoid (WeightMLP: network_utils.py:107) and / or joined
  // with a residual — res_out = res_base + out * res_mask[row] (DeformMLP: skeleton_warp.py:152-161, the template offsets enter
  // the blended position in front of the motion mask) — in the launch that has the value in its hands anyway
  int out_sigmoid;
  const float* res_base;
  const float* res_mask;
  float* res_out;
};

__device__ __forceinline__ int mlp_k(const MlpDesc& d, int l) { return l == 0 ? d.in_pad : (l == d.skip + 1 ? d.in_pad + MLP_W : MLP_W); }

// Fragment-major weights: for neuron tile T (32 neurons) and K-step s (16 inputs) the wave's A operand is the 1 KB run
//   Wf[((T * S + s) * 64 + lane) * 8 + j] = W[32 T + (lane & 31)][16 s + 8 (lane >> 5) + j],   S = K / 16 K-steps of the layer
// (riggs_mlp_pack writes it; the transposed copies of the backward alike).
#define MLP_FRAG 512  // 16-bit values per (tile, K-step) fragment block

// acc[nt][gt] += Wtile(nt, k) * H(gaussian tile gt, k)^T over `steps` K-steps: the weights (A operand) from their
// fragment-major run `Wf` (this wave's first tile at its first step; `tile_stride` values between neuron tiles), the hidden
// vectors (B operand: 8 consecutive k of one Gaussian) from `Bsrc` (LDS or the embedding copy), row stride `bs`
// The weights come from L2 (a layer's 128 KB stay there, shared by every workgroup): ~1000 shader cycles per load, against
// 256 cycles of MFMA work per K-step and wave — one step of look-ahead left every step waiting for its fragments (a layer
// took 8 us per workgroup where its MFMAs need 1.7).  MLP_PF steps are kept in flight in a ring of registers (the hidden
// fragments come from LDS: one step ahead is enough for them).
#define MLP_PF 4
#ifndef MLP_STB_PF
#define MLP_STB_PF 7   // ring depth of the forward's storing product (mlp_gemm_hidden_stb)
#endif
template <int NT, int GT, bool H16>
__device__ __forceinline__ void mlp_gemm_t(f32x16 (&acc)[NT][GT], const unsigned short* __restrict__ Wf, size_t tile_stride,
                                           const unsigned short* Bsrc, int bs, int steps /* a multiple of MLP_PF */, int lane) {
  const int r = lane & 31, kq = (lane >> 5) * 8;
  bf16x8 aq[MLP_PF][NT], bq[2][GT];  // the weight ring; the hidden fragments ping-pong (MLP_PF is even: static parity)
  const unsigned short* wl = Wf + lane * 8;
  const unsigned short* bl = Bsrc + (size_t)r * bs + kq;
  const int last = steps - 1;
#pragma unroll
  for (int p = 0; p < MLP_PF; p++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) aq[p][nt] = *reinterpret_cast<const bf16x8*>(wl + nt * tile_stride + (size_t)p * MLP_FRAG);
#pragma unroll
  for (int gt = 0; gt < GT; gt++) bq[0][gt] = *reinterpret_cast<const bf16x8*>(bl + (size_t)(32 * gt) * bs);
  for (int s0 = 0; s0 < steps; s0 += MLP_PF) {
#pragma unroll
    for (int u = 0; u < MLP_PF; u++) {
      const int s = s0 + u;
      const int sn = min(s + 1, last), sp = min(s + MLP_PF, last);  // (past the end: re-reads of the last step, harmless)
#pragma unroll
      for (int gt = 0; gt < GT; gt++) bq[(u + 1) & 1][gt] = *reinterpret_cast<const bf16x8*>(bl + (size_t)(32 * gt) * bs + 16 * sn);
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int gt = 0; gt < GT; gt++) acc[nt][gt] = mfma16<H16>(aq[u][nt], bq[u & 1][gt], acc[nt][gt]);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) aq[u][nt] = *reinterpret_cast<const bf16x8*>(wl + nt * tile_stride + (size_t)sp * MLP_FRAG);
    }
  }
}
// The product over a hidden layer (K = 256: sixteen steps, the hidden fragments from LDS) with the weight ring OUT OF THE
// COMPILER'S HANDS.  hipcc sinks every global load to just in front of its first use (it minimises live ranges): the ring
// above compiles to load - s_waitcnt vmcnt(0) - MFMA, an L2 round trip (~1000 cycles) in front of every 256 cycles of matrix
// work.  Here the loads are issued by inline asm and waited for with a counted s_waitcnt vmcnt(6) — the six loads of the three
// steps behind the one consumed stay in flight — the CDNA form of cp.async / wait_group.  Straight-line code (STEPS is a
// template argument): no loop-carried copies of registers whose load has not landed; every wait names its two destinations
// as "+v", so the MFMAs that read them cannot be scheduled above it.  The loop holds NO other vector-memory operation (the
// hidden fragments are LDS reads: lgkmcnt), so the count is exact; compiler-issued operations before it are older and only
// make the first waits more conservative.
#define MLP_GLOAD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define MLP_WAIT2(n, a, b) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(a), "+v"(b))
// COPY: the B operand's rows — the layer output that sits in LDS, 128 x 256 values — also leave for HBM here (operand of the
// weight gradients: 1.2 GB per MLP and direction, a quarter of a millisecond of HBM writing that used to sit between two
// barriers with every wave's next product waiting for it: on gfx9 stores and loads share vmcnt, IN ORDER).  The sixteen 16-byte
// stores of a thread are issued BEHIND the ring's first eight loads, so the first four steps' waits leave them in flight
// (vmcnt(6 + 16)); they have to have landed when step 4's fragments — the first loads issued behind them — are waited for:
// four steps of matrix work later.  (One store per K-step, interleaved, LOST: every wait then had a store in front of it.)
template <int STEPS, int GT, bool H16, bool COPY = false>
__device__ __forceinline__ void mlp_gemm_hidden(f32x16 (&acc)[2][GT], const unsigned short* __restrict__ Wf, size_t tile_stride,
                                                const unsigned short* Bsrc, int bs, int lane,
                                                unsigned short* __restrict__ cdst = nullptr, int crows = 0) {
  static_assert(STEPS % MLP_PF == 0 && STEPS >= 2 * MLP_PF && MLP_PF == 4, "the wait counts below are written out for a ring of four");
  const int r = lane & 31, kq = (lane >> 5) * 8;
  bf16x8 aq[MLP_PF][2], bq[2][GT];
  const unsigned short* w0 = Wf + lane * 8;
  const unsigned short* w1 = w0 + tile_stride;
  const unsigned short* bl = Bsrc + (size_t)r * bs + kq;
#pragma unroll
  for (int p = 0; p < MLP_PF; p++) {
    MLP_GLOAD(aq[p][0], w0 + (size_t)p * MLP_FRAG);
    MLP_GLOAD(aq[p][1], w1 + (size_t)p * MLP_FRAG);
  }
#pragma unroll
  for (int gt = 0; gt < GT; gt++) bq[0][gt] = *reinterpret_cast<const bf16x8*>(bl + (size_t)(32 * gt) * bs);
  if constexpr (COPY) {
    // exactly 16 stores per thread, whatever the row count (rows past the end go to the thread's own first piece again: the
    // count in the waits below must not depend on data) — inline asm, so that the order against the ring's loads is the source's
    static_assert(GT == 4, "128 rows: 4096 pieces over 256 threads");
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 2) {  // (two pieces at a time: sixteen LDS reads hoisted in front of the stores spill)
      bf16x8 v[2];
      unsigned short* to[2];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int e = (k0 + k) * 256 + (int)threadIdx.x;
        int cr = e >> 5;
        const int c8 = e & 31;
        cr = cr < crows ? cr : 0;
        v[k] = *reinterpret_cast<const bf16x8*>(Bsrc + (size_t)cr * bs + 8 * c8);
        to[k] = cdst + (size_t)cr * MLP_W + 8 * c8;
      }
#pragma unroll
      for (int k = 0; k < 2; k++) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(to[k]), "v"(v[k]) : "memory");
    }
  }
#pragma unroll
  for (int s = 0; s < STEPS; s++) {
    const int u = s & (MLP_PF - 1);
    // loads younger than this step's: two per step still ahead in the ring
    const int ahead = (STEPS - 1 - s) < (MLP_PF - 1) ? (STEPS - 1 - s) : (MLP_PF - 1);
    if (COPY && s < MLP_PF) MLP_WAIT2(22, aq[u][0], aq[u][1]);  // (+ the sixteen stores issued behind the first eight loads)
    else if (ahead == 3) MLP_WAIT2(6, aq[u][0], aq[u][1]);
    else if (ahead == 2) MLP_WAIT2(4, aq[u][0], aq[u][1]);
    else if (ahead == 1) MLP_WAIT2(2, aq[u][0], aq[u][1]);
    else MLP_WAIT2(0, aq[u][0], aq[u][1]);
    if (s + 1 < STEPS) {
#pragma unroll
      for (int gt = 0; gt < GT; gt++) bq[(s + 1) & 1][gt] = *reinterpret_cast<const bf16x8*>(bl + (size_t)(32 * gt) * bs + 16 * (s + 1));
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int gt = 0; gt < GT; gt++) acc[nt][gt] = mfma16<H16>(aq[u][nt], bq[s & 1][gt], acc[nt][gt]);
    if (s + MLP_PF < STEPS) {  // the slot just consumed takes the step four ahead
      MLP_GLOAD(aq[u][0], w0 + (size_t)(s + MLP_PF) * MLP_FRAG);
      MLP_GLOAD(aq[u][1], w1 + (size_t)(s + MLP_PF) * MLP_FRAG);
    }
  }
}

// The FORWARD's form of "the B operand's rows leave for HBM under the product" (round 6).  The COPY form above issues a layer's
// sixteen stores per thread together, in front of the loop, out of registers of their own: in the forward kernel that took the last
// of the 256 registers and lost what the overlap gained (0.454 -> 0.470 ms).  Here ONE store per thread and K-step: at step s the
// workgroup sends rows 8 s .. 8 s + 7 of the B operand's 128 — 16 bytes per thread, re-read from LDS at the top of the step (the
// MFMAs of the step hide the read), a wave's store = two whole rows of 512 bytes: full lines — through one register quad.
// Stores and loads share vmcnt IN ORDER, so a store issued at step s has to have landed when the fragments loaded behind it are
// waited for: with the ring of four that is one microsecond later, and every wait stood behind a store ("one store per K-step,
// interleaved, LOST" above).  Hence a ring of PF steps (MLP_STB_PF: the weights' loads are waited for PF steps behind their issue,
// and so are the stores in front of them).  The counts, per step s (vector-memory operations issued behind the loads of step s
// when they are waited for): prologue loads L(0..PF-1), then step t issues L(t + PF) (two loads, while they exist) and St(t)
// behind them.
// Measured on the way (profiles/round6_mlp_forward_stores_ab.txt): the same store sourced from the B FRAGMENTS a wave holds anyway
// (wave w sends tile w: no LDS read, no register, but 32-byte pieces of 32 rows per instruction that L2 has to merge) behind a ring
// of six: heads-on iteration 2.65 -> 2.50 ms; as full lines: -> 2.43 (ring 6) / 2.41 (ring 7).  With `nt` on the stores: 3.38 ms.
// The BACKWARD keeps the COPY form: with the fragment-sourced stores behind rings of six and seven its data-gradient pass went
// 0.396 -> 0.445 ms, and with this form behind rings of 6 / 7 / 8 the iteration moved by 2.455 -> 2.449 / 2.457 / 2.440 ms: noise.
template <int N> __device__ __forceinline__ void mlp_wait2(bf16x8& a, bf16x8& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int STEPS, int PF, int S, bool ST> constexpr int mlp_stb_younger() {
  int n = 0;
  if (S < PF) {
    n = 2 * (PF - 1 - S);
    for (int t = 0; t < S; t++) n += (t + PF < STEPS ? 2 : 0) + (ST ? 1 : 0);
  } else {
    n = ST ? 1 : 0;  // St(S - PF): issued behind L(S) in the same step
    for (int t = S - PF + 1; t < S; t++) n += (t + PF < STEPS ? 2 : 0) + (ST ? 1 : 0);
  }
  return n;
}
__device__ unsigned short mlp_store_sink[64 * 8];  // where the store of a row past the end goes (the count of stores must not depend on data)
struct MlpStbRows {  // this thread's 16-byte piece of rows row, row + 8, ..., row + 120: where it lies in LDS and where it goes
  const unsigned short* lrow;
  unsigned short* grow;
  int row, crows, lane;
};
template <int STEPS, int PF, int S, int GT, bool H16, bool ST>
__device__ __forceinline__ void mlp_stb_step(f32x16 (&acc)[2][GT], bf16x8 (&aq)[PF][2], bf16x8 (&bq)[2][GT], const unsigned short* w0,
                                             const unsigned short* w1, const unsigned short* bl, int bs, const MlpStbRows& ro) {
  constexpr int u = S % PF;
  mlp_wait2<mlp_stb_younger<STEPS, PF, S, ST>()>(aq[u][0], aq[u][1]);
  bf16x8 sv;
  if constexpr (ST) sv = *reinterpret_cast<const bf16x8*>(ro.lrow + (size_t)(8 * S) * bs);
  if constexpr (S + 1 < STEPS) {
#pragma unroll
    for (int gt = 0; gt < GT; gt++) bq[(S + 1) & 1][gt] = *reinterpret_cast<const bf16x8*>(bl + (size_t)(32 * gt) * bs + 16 * (S + 1));
  }
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int gt = 0; gt < GT; gt++) acc[nt][gt] = mfma16<H16>(aq[u][nt], bq[S & 1][gt], acc[nt][gt]);
  if constexpr (S + PF < STEPS) {
    MLP_GLOAD(aq[u][0], w0 + (size_t)(S + PF) * MLP_FRAG);
    MLP_GLOAD(aq[u][1], w1 + (size_t)(S + PF) * MLP_FRAG);
  }
  // (exactly one store per thread and step, whatever the row count: rows past the end go to the sink.  s_nop 1: a store of more
  // than 8 bytes needs two wait states before its data registers may be written, and the compiler does not look inside the asm)
  if constexpr (ST) {
    unsigned short* to = (ro.row + 8 * S < ro.crows) ? ro.grow + (size_t)(8 * S) * MLP_W : mlp_store_sink + ro.lane * 8;
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(to), "v"(sv) : "memory");
  }
  if constexpr (S + 1 < STEPS) mlp_stb_step<STEPS, PF, S + 1, GT, H16, ST>(acc, aq, bq, w0, w1, bl, bs, ro);
}
template <int STEPS, int GT, bool H16, int PF, bool ST = true>
__device__ __forceinline__ void mlp_gemm_hidden_stb(f32x16 (&acc)[2][GT], const unsigned short* __restrict__ Wf, size_t tile_stride,
                                                    const unsigned short* Bsrc, int bs, int lane, int wave_u /* uniform */,
                                                    unsigned short* __restrict__ cdst /* row 0 of the workgroup */, int crows) {
  static_assert(GT == 4 && STEPS == 16 && STEPS >= PF, "128 rows = 16 steps x 8 rows; the LDS row stride is MLP_HS");
  const int r = lane & 31, kq = (lane >> 5) * 8;
  bf16x8 aq[PF][2], bq[2][GT];
  const unsigned short* w0 = Wf + lane * 8;
  const unsigned short* w1 = w0 + tile_stride;
  const unsigned short* bl = Bsrc + (size_t)r * bs + kq;
#pragma unroll
  for (int p = 0; p < PF; p++) {
    MLP_GLOAD(aq[p][0], w0 + (size_t)p * MLP_FRAG);
    MLP_GLOAD(aq[p][1], w1 + (size_t)p * MLP_FRAG);
  }
#pragma unroll
  for (int gt = 0; gt < GT; gt++) bq[0][gt] = *reinterpret_cast<const bf16x8*>(bl + (size_t)(32 * gt) * bs);
  const int tid_ = wave_u * 64 + lane;
  MlpStbRows ro;
  ro.row = tid_ >> 5; ro.crows = ST ? crows : 0; ro.lane = lane;
  ro.lrow = Bsrc + (size_t)ro.row * bs + 8 * (tid_ & 31);
  ro.grow = cdst + (size_t)ro.row * MLP_W + 8 * (tid_ & 31);
  mlp_stb_step<STEPS, PF, 0, GT, H16, ST>(acc, aq, bq, w0, w1, bl, bs, ro);
}

// the backward's head product: two K-steps (32 padded outputs), no ring
template <int NT, int GT, bool H16>
__device__ __forceinline__ void mlp_gemm_small(f32x16 (&acc)[NT][GT], const unsigned short* __restrict__ Wf, size_t tile_stride,
                                               const unsigned short* Bsrc, int bs, int steps, int lane) {
  const int r = lane & 31, kq = (lane >> 5) * 8;
  for (int s = 0; s < steps; s++) {
    bf16x8 a[NT], b[GT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) a[nt] = *reinterpret_cast<const bf16x8*>(Wf + lane * 8 + nt * tile_stride + (size_t)s * MLP_FRAG);
#pragma unroll
    for (int gt = 0; gt < GT; gt++) b[gt] = *reinterpret_cast<const bf16x8*>(Bsrc + (size_t)(32 * gt + r) * bs + 16 * s + kq);
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int gt = 0; gt < GT; gt++) acc[nt][gt] = mfma16<H16>(a[nt], b[gt], acc[nt][gt]);
  }
}

// ---- the epilogues' vector work.  A layer's epilogue touches 128 accumulator values per lane; written element by element it
// compiled to ~830 (forward) / ~640 (data gradient) vector instructions per wave and layer — 3 300 / 2 600 issue cycles against the
// 4 096 cycles of the layer's MFMAs.  Packed forms: bias add as v_pk_add_f32, conversion as v_cvt_pk_f16_f32, ReLU as v_pk_max_f16 on
// the converted pair (rounding is monotonic: relu(round(x)) = round(relu(x))), and the ReLU mask shifted in / out through the
// carry: (compare, add-with-carry) appends a bit, (add, select-on-carry) takes one off the top — two instructions per element.
typedef float mlp_f2 __attribute__((ext_vector_type(2)));
typedef _Float16 mlp_h2 __attribute__((ext_vector_type(2)));
// m = (m << 1) | (x > thr)
__device__ __forceinline__ void mlp_mask_push(uint32_t& m, float x, float thr) {
  asm("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(x), "v"(thr) : "vcc");
}
// x = (top bit of m) ? x : 0;  m <<= 1
__device__ __forceinline__ void mlp_mask_pop(uint32_t& m, float& x) {
  asm("v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, 0, %1, vcc" : "+v"(m), "+v"(x) : : "vcc");
}
// two fp32 values -> one dword of two 16-bit values in the operand format, ReLU applied when asked for
template <bool H16, bool RELU> __device__ __forceinline__ uint32_t mlp_pack2(mlp_f2 v) {
  if constexpr (H16) {
    mlp_h2 h = __builtin_convertvector(v, mlp_h2);
    if constexpr (RELU) h = __builtin_elementwise_max(h, (mlp_h2){(_Float16)0.f, (_Float16)0.f});
    return __builtin_bit_cast(uint32_t, h);
  } else {
    if constexpr (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
    return (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
  }
}

// C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int mlp_c_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// one lane's post-activation values of a tile group -> LDS: four CONSECUTIVE neurons of one Gaussian, 8 bytes
__device__ __forceinline__ void mlp_store4(unsigned short* dst, unsigned short v0, unsigned short v1, unsigned short v2, unsigned short v3) {
  *reinterpret_cast<uint2*>(dst) = make_uint2((uint32_t)v0 | ((uint32_t)v1 << 16), (uint32_t)v2 | ((uint32_t)v3 << 16));
}

// STORE: the activations / ReLU masks the backward needs are written (acts, masks non-NULL); a kernel of its own per value — both
// forms of the hidden product inlined behind a run-time test spilled 343 registers
template <int GT, bool H16, bool STORE>  // 32-Gaussian tiles per workgroup (rows per workgroup = 32 GT); every wave owns 64 neurons for all of them
__global__ __launch_bounds__(256, GT == 2 ? 3 : 2) void mlp_forward_kernel(MlpDesc d, const unsigned short* __restrict__ xb,
                                                                            unsigned short* __restrict__ acts /* [depth][N][256] or NULL */,
                                                                            uint4* __restrict__ masks /* [depth][workgroups][256] or NULL */,
                                                                            float* __restrict__ out /* [N][out_ch] */) {
  constexpr int ROWS = 32 * GT;
  __shared__ unsigned short s_h[ROWS * MLP_HS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int row0 = blockIdx.x * ROWS;
  // (a device-side row count — the compacted rows of a row-sparse backward: the grid covers the capacity, the rest leaves here)
  const int rows = d.n_dev ? min(d.n_dev[0], d.N) : d.N;
  if (row0 >= rows) return;
  // the embedding is read from its 16-bit copy in HBM ((rows rounded up to 128) x in_pad, zero padded): it is an operand
  // of two layers only, and keeping it out of LDS lets two 128-row workgroups share a CU
  const unsigned short* xrow = xb + (size_t)row0 * d.in_pad;
  const int n0 = wave * 64;                 // this wave's first neuron
  const int emb_steps = d.in_pad >> 4;
  // every layer's biases once, into LDS: read as global loads inside the epilogue they waited (vmcnt, in order) for the last
  // activation stores of the product in front of them — a store's round trip per layer in the open
  __shared__ float s_bias[10 * MLP_W];
  for (int e = tid; e < d.depth * MLP_W; e += 256) s_bias[e] = d.bias[e >> 8][e & (MLP_W - 1)];
  __syncthreads();
  for (int l = 0; l < d.depth; l++) {
    f32x16 acc[2][GT];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int gt = 0; gt < GT; gt++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[nt][gt][e] = 0.f;
    const int S = mlp_k(d, l) >> 4;         // K-steps of the layer
    const size_t ts = (size_t)S * MLP_FRAG;  // values between neuron tiles
    const unsigned short* Wl = d.Wp[l] + (size_t)(2 * wave) * ts;
    if (l == 0) mlp_gemm_t<2, GT, H16>(acc, Wl, ts, xrow, d.in_pad, emb_steps, lane);
    else {
      const unsigned short* Wh = Wl;
      if (l == d.skip + 1) {
        mlp_gemm_t<2, GT, H16>(acc, Wl, ts, xrow, d.in_pad, emb_steps, lane);
        Wh = Wl + (size_t)emb_steps * MLP_FRAG;
      }
      // (the layer below's activations — this product's B operand — leave for HBM under it, out of the fragment registers:
      // mlp_gemm_hidden_stb; the last layer's go between the barriers below)
      if constexpr (STORE)
        mlp_gemm_hidden_stb<MLP_W / 16, GT, H16, MLP_STB_PF>(acc, Wh, ts, s_h, MLP_HS, lane, wave_u,
                                                              acts + ((size_t)(l - 1) * d.N + row0) * MLP_W, min(ROWS, rows - row0));
      // (the product without stores keeps its ring of four: rings of six and eight measured the same, 2.54 / 2.53 / 2.53 ms)
      else mlp_gemm_hidden<MLP_W / 16, GT, H16, false>(acc, Wh, ts, s_h, MLP_HS, lane);
    }
    __syncthreads();  // every wave is done reading the previous hidden vector
    uint32_t mbits[GT] ;  // ReLU mask of this lane's accumulator elements: word gt, bit nt * 16 + e
#pragma unroll
    for (int gt = 0; gt < GT; gt++) mbits[gt] = 0u;
    const int g_in_tile = lane & 31, nq = 4 * (lane >> 5);
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {          // register group q: neurons n0 + 32 nt + 8 q + nq .. + 3
        const int nb = n0 + 32 * nt + 8 * q + nq;
        const float4 b = *reinterpret_cast<const float4*>(s_bias + l * MLP_W + nb);
        const mlp_f2 b01 = {b.x, b.y}, b23 = {b.z, b.w};
        // (a value is kept when its 16-bit form is non-zero: fp16 rounds everything up to 2^-25 to zero)
        const float thr = H16 ? 0x1p-25f : 0.f;
#pragma unroll
        for (int gt = 0; gt < GT; gt++) {
          const mlp_f2 s01 = (mlp_f2){acc[nt][gt][4 * q], acc[nt][gt][4 * q + 1]} + b01;
          const mlp_f2 s23 = (mlp_f2){acc[nt][gt][4 * q + 2], acc[nt][gt][4 * q + 3]} + b23;
          mlp_mask_push(mbits[gt], s01.x, thr);   // bit 31 - (16 nt + 4 q + j) once all 32 are in
          mlp_mask_push(mbits[gt], s01.y, thr);
          mlp_mask_push(mbits[gt], s23.x, thr);
          mlp_mask_push(mbits[gt], s23.y, thr);
          *reinterpret_cast<uint2*>(s_h + (size_t)(32 * gt + g_in_tile) * MLP_HS + nb) =
              make_uint2(mlp_pack2<H16, true>(s01), mlp_pack2<H16, true>(s23));
        }
      }
    }
    // (the data-gradient kernel uses the same tiling, so the mask travels in the accumulator layout: 16 bytes per lane
    // and layer instead of re-reading the layer's activations)
    if constexpr (STORE) {
      static_assert(GT == 4, "the mask record is one uint4 per lane: four Gaussian tiles");
      masks[((size_t)l * gridDim.x + blockIdx.x) * 256 + tid] = make_uint4(mbits[0], mbits[1], mbits[2], mbits[3]);
    }
    __syncthreads();
    if (STORE && l == d.depth - 1) {  // the last layer's activations (operand of the head's weight gradient): full-line stores
      unsigned short* dst = acts + ((size_t)l * d.N + row0) * MLP_W;
      for (int e = tid; e < ROWS * (MLP_W / 8); e += 256) {
        const int r = e / (MLP_W / 8), c8 = e - r * (MLP_W / 8);
        if (row0 + r < rows)
          *reinterpret_cast<bf16x8*>(dst + (size_t)r * MLP_W + 8 * c8) = *reinterpret_cast<const bf16x8*>(s_h + r * MLP_HS + 8 * c8);
      }
    }
  }
  // output head: 32 (padded) outputs x 32 GT Gaussians, K = 256: wave w < GT takes the Gaussian tile w; the results meet in
  // LDS (the hidden vector's storage, free behind a barrier) and leave as one contiguous run of the workgroup's rows
  f32x16 hacc[1][1];
#pragma unroll
  for (int e = 0; e < 16; e++) hacc[0][0][e] = 0.f;
  if (wave < GT) mlp_gemm_t<1, 1, H16>(hacc, d.Wout, 0, s_h + (size_t)(32 * wave) * MLP_HS, MLP_HS, MLP_W >> 4, lane);
  __syncthreads();
  float* s_out = reinterpret_cast<float*>(s_h);  // [ROWS][33]
  if (wave < GT) {
#pragma unroll
    for (int e = 0; e < 16; e++) s_out[(32 * wave + (lane & 31)) * 33 + mlp_c_row(e, lane)] = hacc[0][0][e];
  }
  __syncthreads();
  const int n_rows = min(ROWS, rows - row0);
  for (int e = tid; e < n_rows * d.out_ch; e += 256) {
    const int r = e / d.out_ch, c = e - r * d.out_ch;
    float v = s_out[r * 33 + c] + d.bout[c];
    if (d.out_sigmoid) v = 1.0f / (1.0f + expf(-v));
    const size_t o = (size_t)row0 * d.out_ch + e;
    out[o] = v;
    if (d.res_out) d.res_out[o] = d.res_base[o] + (d.res_mask ? v * d.res_mask[row0 + r] : v);
  }
}

// Data-gradient pass.  d_post of the last hidden layer comes from the head (g_out W_out, K = 32 padded); then for
// l = depth-1 .. 0:  d_pre_l = d_post_l * [act_l > 0]  (cooperative 16-byte pass: LDS gradient x global activation ->
// LDS, and -> HBM as bf16 for the weight gradients), and for l >= 1  d_post_{l-1} = d_pre_l W_l[:, hidden part]
// with the transposed bf16 copy of the weights as the B operand (row = input feature k, 8 consecutive output neurons
// n contiguous).  No gradient leaves the first layer: both heads' inputs are detached in the reference.
struct MlpBwdDesc {
  int N, out_ch, depth, skip;
  const unsigned short* Wt[10];   // l >= 1: W_l[:, hidden part]^T (rows = units k of layer l - 1, inputs = neurons n), fragment-major: 8 x 16 x 512
  const unsigned short* Wout_t;   // the head transposed (rows = hidden units, inputs = 32 padded outputs), fragment-major: 8 x 2 x 512
  const int32_t* n_dev;           // as MlpDesc::n_dev
};

template <int GT, bool H16>
__global__ __launch_bounds__(256, GT == 2 ? 3 : 2) void mlp_backward_kernel(MlpBwdDesc d, const float* __restrict__ g_out,
                                                           const float* __restrict__ g_scale /* device scalar or NULL */,
                                                           const uint4* __restrict__ masks /* [depth][workgroups][256], from the forward */,
                                                           unsigned short* __restrict__ dpre /* [depth][N][256] */,
                                                           float* __restrict__ db_part /* [workgroups][depth][256] */) {
  constexpr int ROWS = 32 * GT;
  static_assert(GT == 4, "the mask record is one uint4 per lane: four Gaussian tiles");
  __shared__ unsigned short s_d[ROWS * MLP_HS];
  __shared__ unsigned short s_g[ROWS * 40];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * ROWS;
  const int rows = d.n_dev ? min(d.n_dev[0], d.N) : d.N;
  if (row0 >= rows) return;
  const float gs = g_scale ? g_scale[0] : 1.0f;
  for (int e = tid; e < ROWS * 32; e += 256) {
    const int r = e >> 5, c = e & 31;
    float v = 0.f;
    if (row0 + r < rows && c < d.out_ch) v = g_out[(size_t)(row0 + r) * d.out_ch + c] * gs;
    s_g[r * 40 + c] = f2h<H16>(v);
  }
  __syncthreads();
  const int k0 = wave * 64;  // this wave's first hidden unit (of the layer below)
  const int g_in_tile = lane & 31, nq = 4 * (lane >> 5);
  for (int l = d.depth - 1; l >= 0; l--) {
    // ---- d_post_l for this wave's 64 units, transposed like the forward: A = the transposed weights (rows = units of layer
    // l, fragment-major), B = the gradient above (8 consecutive units of one Gaussian, LDS)
    f32x16 acc[2][GT];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int gt = 0; gt < GT; gt++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[nt][gt][e] = 0.f;
    if (l == d.depth - 1) mlp_gemm_small<2, GT, H16>(acc, d.Wout_t + (size_t)(2 * wave) * (2 * MLP_FRAG), 2 * MLP_FRAG, s_g, 40, 2, lane);
    else  // (the gradient of layer l + 1 — this product's B operand — leaves for HBM under it)
      mlp_gemm_hidden<MLP_W / 16, GT, H16, true>(acc, d.Wt[l + 1] + (size_t)(2 * wave) * (16 * MLP_FRAG), 16 * MLP_FRAG, s_d, MLP_HS, lane,
                                                 dpre + ((size_t)(l + 1) * d.N + row0) * MLP_W, min(ROWS, rows - row0));
    __syncthreads();  // every wave is done reading d_pre_{l+1}
    // ---- d_pre_l = d_post_l where the forward's activation was positive (mask bits in this lane's accumulator layout)
    const uint4 mk = masks[((size_t)l * gridDim.x + blockIdx.x) * 256 + tid];
    uint32_t mbits[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int nb = k0 + 32 * nt + 8 * q + nq;
#pragma unroll
        for (int gt = 0; gt < GT; gt++) {
          float x0 = acc[nt][gt][4 * q], x1 = acc[nt][gt][4 * q + 1], x2 = acc[nt][gt][4 * q + 2], x3 = acc[nt][gt][4 * q + 3];
          mlp_mask_pop(mbits[gt], x0);  // (the forward pushed them in this order: the top bit is this element's)
          mlp_mask_pop(mbits[gt], x1);
          mlp_mask_pop(mbits[gt], x2);
          mlp_mask_pop(mbits[gt], x3);
          *reinterpret_cast<uint2*>(s_d + (size_t)(32 * gt + g_in_tile) * MLP_HS + nb) =
              make_uint2(mlp_pack2<H16, false>((mlp_f2){x0, x1}), mlp_pack2<H16, false>((mlp_f2){x2, x3}));
        }
      }
    __syncthreads();
    if (l == 0) {  // (no product follows the first layer's gradient: it leaves here, full lines)
      unsigned short* dl = dpre + (size_t)row0 * MLP_W;
      for (int e = tid; e < ROWS * (MLP_W / 8); e += 256) {
        const int r = e / (MLP_W / 8), c8 = e - r * (MLP_W / 8);
        if (row0 + r < rows)
          *reinterpret_cast<bf16x8*>(dl + (size_t)r * MLP_W + 8 * c8) = *reinterpret_cast<const bf16x8*>(s_d + r * MLP_HS + 8 * c8);
      }
    }
    // bias gradient of the layer: this workgroup's column sums (thread = column; summed over the workgroups in a fixed
    // order by the caller — deterministic, and cheaper than a column reduction of the (N, 256) tensor)
    if (db_part) {  // (NULL: riggs_mlp_wgrad sums the columns of dpre beside its products)
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
// Phase 1: thread = one row of the workgroup's 256.  sin / cos of the base angle once per coordinate (full-precision sincosf),
// every further octave by angle doubling — sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a: two multiply-adds instead of two
// range-reduced transcendentals; the error doubles per octave (<= 2^9 x 6e-8 = 3e-5 at the tenth: a tenth of half a 16-bit ulp at
// 1) — 63 columns cost 3 sincosf + 54 FMAs instead of 60 sinf / cosf calls (the kernel was 55 us of vector work for 38 MB of
// output).  The values meet in LDS (row stride 132 bytes: one column of 32 rows = 32 banks).  Phase 2: the workgroup's rows are
// one contiguous run of the output — 16-byte pieces, consecutive threads = consecutive addresses; a thread's piece sits at the
// same columns in every row it copies, so the tail's values (the same in every row: DeformMLP's pose) are converted once.
#define MLP_EMB_PE 64  // positional-encoding columns staged per row (multires <= 10: 63)
template <bool H16>
__global__ __launch_bounds__(256) void mlp_embed_kernel(int N, int n_rows, int multires, int n_tail, int in_pad,
                                                        const float* __restrict__ x, const float* __restrict__ tail,
                                                        unsigned short* __restrict__ out) {
  __shared__ unsigned short s_pe[256][MLP_EMB_PE + 2];
  const int tid = threadIdx.x, row0 = blockIdx.x * 256;
  const int pe = 3 * (1 + 2 * multires);
  {
    const int n = row0 + tid;
    unsigned short* r = s_pe[tid];
    if (n < N) {
      const float xv[3] = {x[3 * n], x[3 * n + 1], x[3 * n + 2]};
      float sn[3], cs[3];
#pragma unroll
      for (int w = 0; w < 3; w++) { r[w] = f2h<H16>(xv[w]); sincosf(xv[w], &sn[w], &cs[w]); }
      for (int k = 0; k < multires; k++) {
#pragma unroll
        for (int w = 0; w < 3; w++) {
          r[3 + 6 * k + w] = f2h<H16>(sn[w]);
          r[6 + 6 * k + w] = f2h<H16>(cs[w]);
          const float s2 = 2.0f * sn[w] * cs[w], c2 = 1.0f - 2.0f * sn[w] * sn[w];
          sn[w] = s2; cs[w] = c2;
        }
      }
    } else {
      for (int c = 0; c < pe; c++) r[c] = 0;
    }
  }
  __syncthreads();
  const int segs = in_pad >> 3;            // 8 or 16: divides 256, so (e % segs) is the same for every piece of a thread
  const int c8 = (tid % segs) * 8;
  unsigned short tl[8];                    // this thread's columns where they lie in the tail (rows < N)
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int c = c8 + q;
    tl[q] = (c >= pe && c < pe + n_tail) ? f2h<H16>(tail[c - pe]) : (unsigned short)0;
  }
  for (int e = tid; e < 256 * segs; e += 256) {
    const int r = e / segs, n = row0 + r;
    if (n >= n_rows) break;
    bf16x8 v;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int c = c8 + q;
      v[q] = (short)((c < pe) ? s_pe[r][c < MLP_EMB_PE ? c : 0] : (n < N ? tl[q] : (unsigned short)0));
    }
    *reinterpret_cast<bf16x8*>(out + (size_t)n * in_pad + c8) = v;
  }
}


// ---- the rows that carry a gradient (row-sparse backward).  The WeightMLP's cotangent is zero in every row whose Gaussian got
// no gradient from the render (its regulariser is dead code in the reference: train_rig.py:433-444; the skinning backward writes
// exact zeros there), typically 70-90 % of the rows; a zero row contributes zero to every data gradient, bias sum and weight
// product, so the backward runs on the COMPACTED live rows: their embedding rows and cotangent rows are gathered here (ascending
// order, no atomics: the result does not depend on scheduling), the forward is repeated for them alone (the first forward then
// stores no activations at all) and the data-gradient / weight-gradient launches read the live count from the device.
// Launch 1: per 256-row block the live flags (as four 64-bit ballots), their number and max|cotangent| of the block.
// sig (optional): the head's output went through a sigmoid (MlpDesc::out_sigmoid) — the cotangent of the pre-activation is
// g * s (1 - s), formed here and again in the gather (what was a sigmoid_backward launch over the whole tensor in front).
__device__ __forceinline__ float mlp_sig_cot(float g, float sg) { return g * (sg * (1.0f - sg)); }
__global__ __launch_bounds__(256) void mlp_live_flags_kernel(int N, int out_ch, const float* __restrict__ g_out,
                                                             const float* __restrict__ sig,
                                                             unsigned long long* __restrict__ bits /* [blocks][4] */,
                                                             int32_t* __restrict__ counts /* [blocks] */,
                                                             uint32_t* __restrict__ bmax /* [blocks] or NULL */) {
  __shared__ int s_live[256];
  __shared__ uint32_t s_m[4];
  const int tid = threadIdx.x, row0 = blockIdx.x * 256;
  s_live[tid] = 0;
  __syncthreads();
  const int n_rows = min(256, N - row0);
  const float* g = g_out + (size_t)row0 * out_ch;
  const float* sg = sig ? sig + (size_t)row0 * out_ch : nullptr;
  uint32_t m = 0u;
  for (int e = tid; e < n_rows * out_ch; e += 256) {      // coalesced over the block's contiguous run
    const float v = sg ? mlp_sig_cot(g[e], sg[e]) : g[e];
    m = max(m, __float_as_uint(v) & 0x7FFFFFFFu);
    if (v != 0.0f) s_live[e / out_ch] = 1;                // (benign race: every writer stores 1; NaN != 0 counts as live)
  }
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((tid & 63) == 0) s_m[tid >> 6] = m;
  __syncthreads();
  const unsigned long long b = __ballot(s_live[tid] != 0);
  __shared__ int s_cnt[4];
  if ((tid & 63) == 0) {
    bits[(size_t)blockIdx.x * 4 + (tid >> 6)] = b;
    s_cnt[tid >> 6] = __popcll(b);
  }
  __syncthreads();
  if (tid == 0) {
    counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (bmax) bmax[blockIdx.x] = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
  }
}
// Launch 2: block b sums the counts in front of it (<= a few thousand words), ranks its live rows and copies them; with `scale`
// every block also takes the maximum over ALL blocks' maxima (the same few thousand words) — block 0 turns it into the fp16
// gradient scale of riggs_mlp_grad_scale, which then costs no launch of its own (and no atomic).
__global__ __launch_bounds__(256) void mlp_live_gather_kernel(int N, int out_ch, int in_pad, const float* __restrict__ g_out,
                                                              const float* __restrict__ sig,
                                                              const unsigned short* __restrict__ xb,
                                                              const unsigned long long* __restrict__ bits,
                                                              const int32_t* __restrict__ counts, const uint32_t* __restrict__ bmax,
                                                              int32_t* __restrict__ idx,
                                                              int32_t* __restrict__ count, unsigned short* __restrict__ x_live,
                                                              float* __restrict__ g_live, float* __restrict__ scale) {
  __shared__ int s_red[4];
  __shared__ uint32_t s_mx[4];
  __shared__ int s_rows[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row0 = blockIdx.x * 256;
  int part = 0;
  for (int i = tid; i < (int)blockIdx.x; i += 256) part += counts[i];
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) s_red[wave] = part;
  if (scale && blockIdx.x == 0) {
    uint32_t m = 0u;
    for (int i = tid; i < (int)gridDim.x; i += 256) m = max(m, bmax[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if (lane == 0) s_mx[wave] = m;
  }
  const unsigned long long b0 = bits[(size_t)blockIdx.x * 4], b1 = bits[(size_t)blockIdx.x * 4 + 1],
                           b2 = bits[(size_t)blockIdx.x * 4 + 2], b3 = bits[(size_t)blockIdx.x * 4 + 3];
  __syncthreads();
  if (scale && blockIdx.x == 0 && tid == 0) {
    const float amax = fmaxf(__uint_as_float(max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]))), 1e-30f);
    scale[0] = exp2f(floorf(log2f(1024.0f / amax)));
  }
  const int base = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  const int c0 = __popcll(b0), c1 = __popcll(b1), c2 = __popcll(b2), c3 = __popcll(b3);
  const unsigned long long mine = wave == 0 ? b0 : (wave == 1 ? b1 : (wave == 2 ? b2 : b3));
  const int before = wave == 0 ? 0 : (wave == 1 ? c0 : (wave == 2 ? c0 + c1 : c0 + c1 + c2));
  if ((mine >> lane) & 1ull) {
    const int rk = before + __popcll(mine & ((1ull << lane) - 1ull));
    s_rows[rk] = row0 + tid;
    idx[base + rk] = row0 + tid;
  }
  const int n_live = c0 + c1 + c2 + c3;
  __syncthreads();
  const int segs = in_pad >> 3;  // 16-byte pieces per embedding row
  for (int e = tid; e < n_live * segs; e += 256) {
    const int r = e / segs, c = e - r * segs;
    *reinterpret_cast<bf16x8*>(x_live + (size_t)(base + r) * in_pad + 8 * c) =
        *reinterpret_cast<const bf16x8*>(xb + (size_t)s_rows[r] * in_pad + 8 * c);
  }
  for (int e = tid; e < n_live * out_ch; e += 256) {
    const int r = e / out_ch, c = e - r * out_ch;
    const size_t src = (size_t)s_rows[r] * out_ch + c;
    g_live[(size_t)(base + r) * out_ch + c] = sig ? mlp_sig_cot(g_out[src], sig[src]) : g_out[src];
  }
  if (blockIdx.x == gridDim.x - 1) {  // the last block knows the total: it publishes it and zero-fills the rows up to the next
    const int total = base + n_live;  // multiple of 128 (the forward kernel's workgroups read whole 128-row tiles of x_live)
    if (tid == 0) count[0] = total;
    const int pad_rows = ((total + 127) & ~127) - total;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = tid; e < pad_rows * segs; e += 256)
      *reinterpret_cast<bf16x8*>(x_live + (size_t)total * in_pad + 8 * (size_t)e) = z;
  }
}

// fp32 master weights -> the bf16 operand layouts of both kernels, in ONE launch (the masters change every optimizer step,
// and ~60 small conversion / padding / transposition launches per MLP were a tenth of a heads-on training iteration)
struct MlpPackDesc {
  int in_ch, in_pad, out_ch, depth, skip;
  int tail_ch;                   // columns of the masters' input part that are NOT packed (riggs_mlp_pack_tail): they multiply a
                                 // vector that is the same for every row and enter the layer through its bias (riggs_mlp_tail_bias)
  const float* W[10];            // (256, K_true): K_true = in_ch + tail_ch, in_ch + tail_ch + 256 (layer skip + 1) or 256
  const float* Wout;             // (out_ch, 256)
  unsigned short* Wp[10];        // 256 x K_pad values, fragment-major
  unsigned short* Wt[10];        // l >= 1: the hidden part transposed, 256 x 256 values, fragment-major
  unsigned short* Wout_p;        // 32 x 256 values, fragment-major
  unsigned short* Wout_t;        // 256 x 32 values, fragment-major
};
// value j of lane `lane` of the fragment block (tile T, step st) of a matrix with rows r and inputs k: row 32 T + (lane & 31),
// input 16 st + 8 (lane >> 5) + j
template <bool H16>
__global__ __launch_bounds__(256) void mlp_pack_kernel(MlpPackDesc d) {
  const int l = blockIdx.y;  // depth = the head
  const int tid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  if (l == d.depth) {
    for (int e = tid; e < 16 * MLP_FRAG; e += stride) {           // Wout: 1 tile x 16 steps
      const int st = e / MLP_FRAG, w = e - st * MLP_FRAG, lane = w >> 3, j = w & 7;
      const int c = lane & 31, k = 16 * st + 8 * (lane >> 5) + j;
      d.Wout_p[e] = f2h<H16>(c < d.out_ch ? d.Wout[(size_t)c * MLP_W + k] : 0.f);
    }
    for (int e = tid; e < 8 * 2 * MLP_FRAG; e += stride) {         // Wout^T: 8 tiles (hidden units) x 2 steps (32 padded outputs)
      const int T = e / (2 * MLP_FRAG), rem = e - T * (2 * MLP_FRAG), st = rem / MLP_FRAG, w = rem - st * MLP_FRAG, lane = w >> 3, j = w & 7;
      const int k = 32 * T + (lane & 31), c = 16 * st + 8 * (lane >> 5) + j;
      d.Wout_t[e] = f2h<H16>(c < d.out_ch ? d.Wout[(size_t)c * MLP_W + k] : 0.f);
    }
    return;
  }
  const bool first = (l == 0), sk = (l == d.skip + 1);
  const int in_true = d.in_ch + d.tail_ch;
  const int k_true = first ? in_true : (sk ? in_true + MLP_W : MLP_W);
  const int k_pad = first ? d.in_pad : (sk ? d.in_pad + MLP_W : MLP_W);
  const int hoff_true = sk ? in_true : 0, hoff_pad = sk ? d.in_pad : 0;  // where the hidden part starts
  const float* W = d.W[l];
  const int S = k_pad >> 4;
  for (int e = tid; e < MLP_W * k_pad; e += stride) {              // W_l: 8 tiles x S steps
    const int T = e / (S * MLP_FRAG), rem = e - T * (S * MLP_FRAG), st = rem / MLP_FRAG, w = rem - st * MLP_FRAG, lane = w >> 3, j = w & 7;
    const int n = 32 * T + (lane & 31), k = 16 * st + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (first || (sk && k < hoff_pad)) { if (k < d.in_ch) v = W[(size_t)n * k_true + k]; }
    else v = W[(size_t)n * k_true + hoff_true + (k - hoff_pad)];
    d.Wp[l][e] = f2h<H16>(v);
  }
  if (!first)
    for (int e = tid; e < MLP_W * MLP_W; e += stride) {            // (hidden part of W_l)^T: 8 tiles (units k) x 16 steps (neurons n)
      const int T = e / (16 * MLP_FRAG), rem = e - T * (16 * MLP_FRAG), st = rem / MLP_FRAG, w = rem - st * MLP_FRAG, lane = w >> 3, j = w & 7;
      const int k = 32 * T + (lane & 31), n = 16 * st + 8 * (lane >> 5) + j;
      d.Wt[l][e] = f2h<H16>(W[(size_t)n * k_true + hoff_true + k]);
    }
}

// The constant tail of the input (DeformMLP: every row's input ends in the SAME pose vector — skeleton_warp.py:152 expands
// local_rot over the Gaussians, detached): W[:, in_ch : in_ch + tail_ch] · tail is one vector per layer that reads the input (the
// first and layer skip + 1), so it joins the bias — in fp32, where the packed operands would have rounded the pose to 16 bits —
// and the layers' products run over the positional embedding alone (K 128 -> 64).  bias_eff[y][n] = b_y[n] + sum_k W_y[n][in_ch + k] tail[k];
// a wave per row.
__global__ __launch_bounds__(256) void mlp_tail_bias_kernel(int in_ch, int tail_ch, const float* __restrict__ W0, const float* __restrict__ b0,
                                                            const float* __restrict__ Ws, const float* __restrict__ bs,
                                                            const float* __restrict__ tail, float* __restrict__ bias_eff) {
  const int y = blockIdx.y, n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ld = in_ch + tail_ch + (y ? MLP_W : 0);
  const float* row = (y ? Ws : W0) + (size_t)n * ld + in_ch;
  float acc = 0.f;
  for (int k = lane; k < tail_ch; k += 64) acc = fmaf(row[k], tail[k], acc);
  acc = wave_sum(acc);
  if (lane == 63) bias_eff[y * MLP_W + n] = (y ? bs : b0)[n] + acc;
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

// 128-row workgroups: 2 (neuron) x 4 (Gaussian) tiles of 32 x 32 per wave (64-row workgroups cost 0.84 / 1.01 ms against
// 0.66 / 0.63 ms: twice the weight loads per MFMA)
#define MLP_RT 4

static int mlp_fill(MlpDesc& d, int32_t N, int32_t in_ch, int32_t out_ch, int32_t depth, int32_t skip, const void* const* Wp,
                    const float* const* bias, const void* Wout, const float* bout, const int32_t* n_dev,
                    const riggs_mlp_epilogue* epi = nullptr) {
  d.n_dev = n_dev;
  d.out_sigmoid = 0; d.res_base = nullptr; d.res_mask = nullptr; d.res_out = nullptr;
  if (epi) {
    d.out_sigmoid = epi->sigmoid ? 1 : 0;
    d.res_base = epi->res_base; d.res_mask = epi->res_mask; d.res_out = epi->res_out;
    RIGGS_REQUIRE((epi->res_out == nullptr) == (epi->res_base == nullptr), "riggs_mlp_epilogue: res_base and res_out come together");
    RIGGS_REQUIRE(epi->res_mask == nullptr || epi->res_out != nullptr, "riggs_mlp_epilogue: res_mask without res_out");
  }
  RIGGS_REQUIRE(N >= 0 && depth >= 1 && depth <= 10, "MLP depth out of range");
  RIGGS_REQUIRE(in_ch >= 1 && in_ch <= MLP_MAX_IN, "MLP input width must be <= 128");
  RIGGS_REQUIRE(out_ch >= 1 && out_ch <= 32, "MLP output width must be <= 32");
  RIGGS_REQUIRE(skip >= 0 && skip < depth - 1, "MLP skip layer out of range");
  d.N = N; d.in_ch = in_ch; d.in_pad = (in_ch + 63) & ~63; d.out_ch = out_ch; d.depth = depth; d.skip = skip;
  for (int l = 0; l < depth; l++) { d.Wp[l] = (const unsigned short*)Wp[l]; d.bias[l] = bias[l]; RIGGS_REQUIRE(Wp[l] && bias[l], "MLP layer pointers"); }
  d.Wout = (const unsigned short*)Wout; d.bout = bout;
  RIGGS_REQUIRE(Wout && bout, "MLP head pointers");
  return 0;
}

int riggs_mlp_forward(int32_t N, int32_t in_ch, int32_t out_ch, int32_t depth, int32_t skip, const void* const* weights_bf16,
                      const float* const* biases, const void* w_out_bf16, const float* b_out, const void* x_emb_bf16,
                      void* acts_bf16, void* relu_masks, float* out, const int32_t* n_rows_dev,
                      const struct riggs_mlp_epilogue* epilogue, int32_t fp16, riggs_stream stream) {
  MlpDesc d;
  int rc = mlp_fill(d, N, in_ch, out_ch, depth, skip, weights_bf16, biases, w_out_bf16, b_out, n_rows_dev, epilogue);
  if (rc) return rc;
  if (N == 0) return 0;
  RIGGS_REQUIRE(x_emb_bf16 && out, "MLP input / output pointers");
  RIGGS_REQUIRE((acts_bf16 == nullptr) == (relu_masks == nullptr), "acts_bf16 and relu_masks: both (training) or neither (inference)");
  const dim3 grid((N + 127) / 128), block(256);
  hipStream_t s = (hipStream_t)stream;
  const unsigned short* xb = (const unsigned short*)x_emb_bf16;
  unsigned short* ab = (unsigned short*)acts_bf16;
  uint4* mk = (uint4*)relu_masks;
  if (ab) {
    if (fp16) hipLaunchKernelGGL((mlp_forward_kernel<MLP_RT, true, true>), grid, block, 0, s, d, xb, ab, mk, out);
    else hipLaunchKernelGGL((mlp_forward_kernel<MLP_RT, false, true>), grid, block, 0, s, d, xb, ab, mk, out);
  } else {
    if (fp16) hipLaunchKernelGGL((mlp_forward_kernel<MLP_RT, true, false>), grid, block, 0, s, d, xb, ab, mk, out);
    else hipLaunchKernelGGL((mlp_forward_kernel<MLP_RT, false, false>), grid, block, 0, s, d, xb, ab, mk, out);
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_mlp_backward(int32_t N, int32_t out_ch, int32_t depth, int32_t skip, const void* const* weights_t_bf16,
                       const void* w_out_t_bf16, const float* g_out, const float* g_scale, const void* relu_masks, void* dpre_bf16,
                       float* db_partial, const int32_t* n_rows_dev, int32_t fp16, riggs_stream stream) {
  RIGGS_REQUIRE(N >= 0 && depth >= 1 && depth <= 10, "MLP depth out of range");
  RIGGS_REQUIRE(out_ch >= 1 && out_ch <= 32, "MLP output width must be <= 32");
  if (N == 0) return 0;
  MlpBwdDesc d;
  d.N = N; d.out_ch = out_ch; d.depth = depth; d.skip = skip; d.n_dev = n_rows_dev;
  for (int l = 0; l < depth; l++) { d.Wt[l] = (const unsigned short*)weights_t_bf16[l]; RIGGS_REQUIRE(l == 0 || d.Wt[l], "MLP transposed weights"); }
  d.Wout_t = (const unsigned short*)w_out_t_bf16;
  RIGGS_REQUIRE(d.Wout_t && g_out && relu_masks && dpre_bf16, "MLP backward pointers");
  if (fp16) hipLaunchKernelGGL((mlp_backward_kernel<MLP_RT, true>), dim3((N + 127) / 128), dim3(256), 0, (hipStream_t)stream, d, g_out,
                               g_scale, (const uint4*)relu_masks, (unsigned short*)dpre_bf16, db_partial);
  else hipLaunchKernelGGL((mlp_backward_kernel<MLP_RT, false>), dim3((N + 127) / 128), dim3(256), 0, (hipStream_t)stream, d, g_out,
                          g_scale, (const uint4*)relu_masks, (unsigned short*)dpre_bf16, db_partial);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}


size_t riggs_mlp_live_rows_workspace_bytes(int32_t N) {
  const size_t blocks = N > 0 ? ((size_t)N + 255) / 256 : 0;
  return blocks * (4 * sizeof(unsigned long long) + 2 * sizeof(int32_t)) + 64;
}

int riggs_mlp_live_rows(int32_t N, int32_t out_ch, int32_t in_ch, const float* g_out, const float* sigmoid_out, const void* x_emb_bf16,
                        void* workspace, int32_t* live_idx, int32_t* live_count, void* x_live_bf16, float* g_live, float* scale,
                        riggs_stream stream) {
  RIGGS_REQUIRE(N >= 0 && out_ch >= 1 && out_ch <= 32 && in_ch >= 1 && in_ch <= MLP_MAX_IN, "riggs_mlp_live_rows: shape out of range");
  RIGGS_REQUIRE(live_count, "riggs_mlp_live_rows: live_count");
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) {
    RIGGS_HIP_CHECK(hipMemsetAsync(live_count, 0, sizeof(int32_t), s));
    if (scale) { static const float one = 1.0f; RIGGS_HIP_CHECK(hipMemcpyAsync(scale, &one, sizeof(float), hipMemcpyHostToDevice, s)); }
    return 0;
  }
  RIGGS_REQUIRE(g_out && x_emb_bf16 && workspace && live_idx && x_live_bf16 && g_live, "riggs_mlp_live_rows: pointers");
  RIGGS_REQUIRE(((uintptr_t)workspace & 7) == 0, "riggs_mlp_live_rows: the workspace must be 8-byte aligned");
  const int blocks = (N + 255) / 256;
  unsigned long long* bits = (unsigned long long*)workspace;
  int32_t* counts = (int32_t*)(bits + (size_t)blocks * 4);
  uint32_t* bmax = (uint32_t*)(counts + blocks);
  hipLaunchKernelGGL(mlp_live_flags_kernel, dim3(blocks), dim3(256), 0, s, N, out_ch, g_out, sigmoid_out, bits, counts, scale ? bmax : nullptr);
  RIGGS_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(mlp_live_gather_kernel, dim3(blocks), dim3(256), 0, s, N, out_ch, (in_ch + 63) & ~63, g_out, sigmoid_out,
                     (const unsigned short*)x_emb_bf16, bits, counts, bmax, live_idx, live_count, (unsigned short*)x_live_bf16, g_live,
                     scale);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_mlp_embed(int32_t N, int32_t multires, int32_t n_tail, const float* x, const float* tail, void* out_bf16,
                    int32_t fp16, riggs_stream stream) {
  RIGGS_REQUIRE(N >= 0 && multires >= 0 && n_tail >= 0, "MLP embedding arguments");
  const int in_ch = 3 * (1 + 2 * multires) + n_tail, in_pad = (in_ch + 63) & ~63;
  RIGGS_REQUIRE(in_ch <= MLP_MAX_IN, "MLP input width must be <= 128");
  const int n_rows = (N + 127) / 128 * 128;
  if (n_rows == 0) return 0;
  RIGGS_REQUIRE(x && out_bf16 && (n_tail == 0 || tail), "MLP embedding pointers");
  RIGGS_REQUIRE(3 * (1 + 2 * multires) <= MLP_EMB_PE, "MLP embedding: multires must be <= 10");
  const size_t n_thr = (size_t)n_rows;
  if (fp16) hipLaunchKernelGGL(mlp_embed_kernel<true>, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, n_rows,
                               multires, n_tail, in_pad, x, tail, (unsigned short*)out_bf16);
  else hipLaunchKernelGGL(mlp_embed_kernel<false>, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, n_rows,
                          multires, n_tail, in_pad, x, tail, (unsigned short*)out_bf16);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_mlp_pack_tail(int32_t in_ch, int32_t tail_ch, int32_t out_ch, int32_t depth, int32_t skip, const float* const* weights,
                        const float* w_out, void* const* weights_bf16, void* const* weights_t_bf16, void* w_out_bf16,
                        void* w_out_t_bf16, int32_t fp16, riggs_stream stream) {
  RIGGS_REQUIRE(depth >= 1 && depth <= 10 && in_ch >= 1 && in_ch <= MLP_MAX_IN && out_ch >= 1 && out_ch <= 32 && skip >= 0 &&
                skip < depth - 1 && tail_ch >= 0 && tail_ch <= 4096, "MLP shape out of range");
  MlpPackDesc d;
  d.in_ch = in_ch; d.in_pad = (in_ch + 63) & ~63; d.out_ch = out_ch; d.depth = depth; d.skip = skip; d.tail_ch = tail_ch;
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

int riggs_mlp_pack(int32_t in_ch, int32_t out_ch, int32_t depth, int32_t skip, const float* const* weights, const float* w_out,
                   void* const* weights_bf16, void* const* weights_t_bf16, void* w_out_bf16, void* w_out_t_bf16,
                   int32_t fp16, riggs_stream stream) {
  return riggs_mlp_pack_tail(in_ch, 0, out_ch, depth, skip, weights, w_out, weights_bf16, weights_t_bf16, w_out_bf16, w_out_t_bf16, fp16,
                             stream);
}

int riggs_mlp_tail_bias(int32_t in_ch, int32_t tail_ch, const float* w_first, const float* b_first, const float* w_skip,
                        const float* b_skip, const float* tail, float* bias_eff, riggs_stream stream) {
  RIGGS_REQUIRE(in_ch >= 1 && in_ch <= MLP_MAX_IN && tail_ch >= 1 && tail_ch <= 4096, "riggs_mlp_tail_bias: widths out of range");
  RIGGS_REQUIRE(w_first && b_first && w_skip && b_skip && tail && bias_eff, "riggs_mlp_tail_bias: pointers");
  hipLaunchKernelGGL(mlp_tail_bias_kernel, dim3(MLP_W / 4, 2), dim3(256), 0, (hipStream_t)stream, in_ch, tail_ch, w_first, b_first, w_skip,
                     b_skip, tail, bias_eff);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

/* rows per workgroup of the MLP kernels = rows that one slice of riggs_mlp_backward's db_partial covers */
int32_t riggs_mlp_rows_per_workgroup(void) { return 32 * MLP_RT; }

}  // extern "C"
