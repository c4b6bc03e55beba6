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
#include "dq_math.h"

namespace riggs {

#define DQ_MAX_K_SHARED 1024
#define DQ_MAX_K_ROWS 8

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

// ---- ROWS: every row its own K <= 8 transforms, a thread per row (dq_math.h: dq_row) ---------------------------------------
template <int KK, bool BWD>
__global__ __launch_bounds__(256) void dqb_rows_kernel(DqArgs a) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < a.N) dq_row<KK, BWD>(a, n);
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
    static unsigned long long attr = 0ull;  // (per device)
    if (once_per_device(attr)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dqb_shared_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
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
    static unsigned long long attr = 0ull;
    if (once_per_device(attr)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dqb_shared_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dqb_shared_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dqb_shared_bwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
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
