// PoseMLP (time -> per-joint quaternions + root translation) as three launches instead of
// ~60 torch ops: /root/reference/skeleton_utils/network_utils.py:115-150 with the positional
// embedding of utils/time_utils.py:208-256 (input_dims 1, [t, sin(2^k t), cos(2^k t)]).
//
// The network is ONE row wide (batch 1): every layer is a GEMV over a <= 273 x 256 weight
// matrix that lives in L2 after the first iteration, so the job is pure latency.  One
// 1024-thread workgroup (16 wave64) walks the layers: a wave owns output rows, its lanes
// stride the row (coalesced 256-B reads), DPP reduces the 64 partial sums.  The backward chain
// (dz_l, dh_l) runs the same way with thread-per-column accumulation (coalesced along rows);
// the 0.5 M weight gradients are outer products written by a wide second kernel.
#include "common.h"

namespace riggs {

#define PM_MAX_LAYERS 12
#define PM_MAX_W 256
#define PM_MAX_IN 320
#define PM_MAX_EMB 64
#define PM_RPW 8       // rows per wave per pass (forward GEMV)
#define PM_NG 16       // row groups (= waves) of the transposed GEMV in the backward chain
#define PM_CPL 5       // columns per lane: ceil(PM_MAX_IN / 64)

struct PoseMlpDesc {
  int depth;        // hidden layers (8)
  int width;        // hidden width (<= 256)
  int multires;     // embedding frequencies (8) -> emb = 1 + 2*multires
  int skip;         // after layer `skip` the embedding is concatenated IN FRONT of h (network_utils.py:144-145)
  int n_rot;        // rotation head outputs (4 J)
  const float* W[PM_MAX_LAYERS];  // hidden layers, row-major (out, in)
  const float* b[PM_MAX_LAYERS];
  const float *W_rot, *b_rot, *W_tr, *b_tr;
};

__device__ __forceinline__ int pm_in_dim(const PoseMlpDesc& d, int l, int emb) {
  if (l == 0) return emb;
  return (l - 1 == d.skip) ? d.width + emb : d.width;
}

// acts layout (floats): [0, emb) embedding, then per layer l: width post-ReLU activations.
// A single workgroup can pull only ~25 GB/s through its CU, and the 2 MB of weights are evicted
// from L2 by the streaming kernels of every iteration, so each layer is spread over 64+
// workgroups (one wave per output row) and the layers are separate graph nodes.

// input element i of layer l (embedding / previous activations / [emb, h] after the skip layer)
__device__ __forceinline__ float pm_input(const PoseMlpDesc& d, const float* __restrict__ acts, int l, int emb, int i) {
  if (l == 0) return acts[i];
  if (l - 1 == d.skip) return (i < emb) ? acts[i] : acts[emb + (size_t)(l - 1) * d.width + (i - emb)];
  return acts[emb + (size_t)(l - 1) * d.width + i];
}

__global__ __launch_bounds__(64) void pm_embed_kernel(PoseMlpDesc d, const float* __restrict__ t, float* __restrict__ acts) {
  const int tid = threadIdx.x, emb = 1 + 2 * d.multires;
  if (tid < emb) {
    const float tv = t[0];
    float v = tv;
    if (tid > 0) {
      const int k = (tid - 1) >> 1;
      const float f = (float)(1 << k);
      v = ((tid - 1) & 1) ? cosf(tv * f) : sinf(tv * f);
    }
    acts[tid] = v;
  }
}

// layer l (l == depth: the two heads): one wave64 per output row
__global__ __launch_bounds__(256) void pm_layer_kernel(PoseMlpDesc d, int l, float* __restrict__ acts,
                                                       const float* __restrict__ rot_bias4,
                                                       float* __restrict__ rotation, float* __restrict__ translation) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int emb = 1 + 2 * d.multires;
  const bool heads = (l == d.depth);
  const int n_out = heads ? d.n_rot + 3 : d.width;
  const int r = blockIdx.x * 4 + wave;
  if (r >= n_out) return;
  const int in_dim = pm_in_dim(d, l, emb);  // for l == depth this is the heads' input width
  const float* row;
  float bias;
  if (!heads) { row = d.W[l] + (size_t)r * in_dim; bias = d.b[l][r]; }
  else if (r < d.n_rot) { row = d.W_rot + (size_t)r * in_dim; bias = d.b_rot[r]; }
  else { row = d.W_tr + (size_t)(r - d.n_rot) * in_dim; bias = d.b_tr[r - d.n_rot]; }
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < PM_CPL; k++) {
    const int i = lane + 64 * k;
    if (i < in_dim) acc += row[i] * pm_input(d, acts, l, emb, i);
  }
  acc = wave_sum(acc);
  if (lane == 63) {
    const float v = acc + bias;
    if (!heads) acts[emb + (size_t)l * d.width + r] = fmaxf(v, 0.f);
    else if (r < d.n_rot) rotation[r] = rot_bias4 ? v + rot_bias4[r & 3] : v;
    else translation[r - d.n_rot] = v;
  }
}

// Backward step for the consumer `l` (l == depth: heads, else hidden layer l):
//   v = dz of the consumer (for the heads: the incoming output gradients),
//   dh_in[c] = sum_r M[r][c] v[r]   for the consumer's input columns c.
// Each block owns 64 columns; its 4 waves split the rows (coalesced 256-B row segments), LDS combines.
// For hidden layers v = dz_l = dh_l * relu'(h_l) is computed by every block (block 0 stores it for the
// weight-gradient kernel).  dh buffers: dh[l] holds the gradient w.r.t. the INPUT of consumer l.
#define PM_RG 8  // row groups of the transposed GEMV (grid.y)
__global__ __launch_bounds__(256) void pm_backward_step_kernel(PoseMlpDesc d, int l, const float* __restrict__ acts,
                                                               const float* __restrict__ g_rot,
                                                               const float* __restrict__ g_tr,
                                                               const float* __restrict__ dh_out /* input-grad of consumer l+1 */,
                                                               float* __restrict__ dh_in /* zeroed */, float* __restrict__ dzs) {
  __shared__ float s_part[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int emb = 1 + 2 * d.multires;
  const bool heads = (l == d.depth);
  const int n_rows = heads ? d.n_rot + 3 : d.width;
  const int in_dim = pm_in_dim(d, l, emb);
  // dh of h_l lives in the consumer (l+1)'s input gradient, behind the embedding when that input was [emb, h]
  const int off = (l == d.skip) ? emb : 0;
  const int c = blockIdx.x * 64 + lane;
  const int rstep = 4 * PM_RG, r0 = blockIdx.y * 4 + wave;
  float acc = 0.f;
#pragma unroll 4
  for (int r = r0; r < n_rows; r += rstep) {
    float v;
    const float* row;
    if (heads) {
      v = (r < d.n_rot) ? g_rot[r] : g_tr[r - d.n_rot];
      row = (r < d.n_rot) ? d.W_rot + (size_t)r * in_dim : d.W_tr + (size_t)(r - d.n_rot) * in_dim;
    } else {
      const float h = acts[emb + (size_t)l * d.width + r];
      v = (h > 0.f) ? dh_out[off + r] : 0.f;  // dz_l
      row = d.W[l] + (size_t)r * in_dim;
      if (blockIdx.x == 0 && lane == 0) dzs[(size_t)l * d.width + r] = v;
    }
    if (l > 0 && c < in_dim) acc += row[c] * v;
  }
  if (l == 0) return;  // the embedding has no trainable input
  s_part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < in_dim) atomicAdd(&dh_in[c], s_part[0][lane] + s_part[1][lane] + s_part[2][lane] + s_part[3][lane]);
}

// Weight / bias gradients: one block per (parameter matrix, row): dW[r][:] = v[r] * input[:], db[r] = v[r].
// Flat layout: [W_0, b_0, ..., W_{depth-1}, b_{depth-1}, W_rot, b_rot, W_tr, b_tr]
struct PoseMlpGradDesc {
  int64_t w_off[PM_MAX_LAYERS + 2];  // offset of W block (hidden layers, rot head, tr head)
  int64_t b_off[PM_MAX_LAYERS + 2];
  int row_start[PM_MAX_LAYERS + 3];  // prefix of rows over the depth + 2 matrices
};
__global__ __launch_bounds__(64) void pm_backward_weights_kernel(PoseMlpDesc d, PoseMlpGradDesc g,
                                                                 const float* __restrict__ acts,
                                                                 const float* __restrict__ dzs,
                                                                 const float* __restrict__ g_rot,
                                                                 const float* __restrict__ g_tr,
                                                                 float* __restrict__ flat) {
  const int emb = 1 + 2 * d.multires;
  const int nmat = d.depth + 2;
  int m = 0;
  while (m + 1 < nmat && (int)blockIdx.x >= g.row_start[m + 1]) m++;
  const int r = blockIdx.x - g.row_start[m];
  const int l = (m < d.depth) ? m : d.depth;  // input of the heads = input index `depth`
  const int in_dim = pm_in_dim(d, l, emb);
  const float v = (m < d.depth) ? dzs[(size_t)m * d.width + r] : (m == d.depth ? g_rot[r] : g_tr[r]);
  float* out = flat + g.w_off[m] + (size_t)r * in_dim;
  for (int c = threadIdx.x; c < in_dim; c += 64) out[c] = v * pm_input(d, acts, l, emb, c);
  if (threadIdx.x == 0) flat[g.b_off[m] + r] = v;
}

}  // namespace riggs

using namespace riggs;

extern "C" {

static int pm_fill(PoseMlpDesc& d, int32_t depth, int32_t width, int32_t multires, int32_t skip, int32_t n_rot,
                   const float* const* weights, const float* const* biases, const float* W_rot, const float* b_rot,
                   const float* W_tr, const float* b_tr) {
  RIGGS_REQUIRE(depth >= 1 && depth <= PM_MAX_LAYERS, "PoseMLP depth out of range");
  RIGGS_REQUIRE(width >= 1 && width <= PM_MAX_W, "PoseMLP width must be <= 256");
  RIGGS_REQUIRE(multires >= 0 && 1 + 2 * multires <= PM_MAX_EMB, "PoseMLP multires out of range");
  RIGGS_REQUIRE(n_rot >= 1 && n_rot <= 4 * 64, "PoseMLP rotation head too wide");
  d.depth = depth; d.width = width; d.multires = multires; d.skip = skip; d.n_rot = n_rot;
  for (int l = 0; l < depth; l++) { d.W[l] = weights[l]; d.b[l] = biases[l]; }
  d.W_rot = W_rot; d.b_rot = b_rot; d.W_tr = W_tr; d.b_tr = b_tr;
  return 0;
}

size_t riggs_pose_mlp_acts_floats(int32_t depth, int32_t width, int32_t multires) {
  return (size_t)(1 + 2 * multires) + (size_t)depth * width;
}

int riggs_pose_mlp_forward(int32_t depth, int32_t width, int32_t multires, int32_t skip, int32_t n_rot,
                           const float* const* weights, const float* const* biases, const float* W_rot,
                           const float* b_rot, const float* W_tr, const float* b_tr, const float* t,
                           const float* rot_bias4, float* acts, float* rotation, float* translation,
                           riggs_stream stream) {
  PoseMlpDesc d;
  int rc = pm_fill(d, depth, width, multires, skip, n_rot, weights, biases, W_rot, b_rot, W_tr, b_tr);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(PROF_POSE_FWD, s);
  hipLaunchKernelGGL(pm_embed_kernel, dim3(1), dim3(64), 0, s, d, t, acts);
  for (int l = 0; l <= depth; l++) {
    const int n_out = (l == depth) ? n_rot + 3 : width;
    hipLaunchKernelGGL(pm_layer_kernel, dim3((n_out + 3) / 4), dim3(256), 0, s, d, l, acts, rot_bias4, rotation, translation);
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

size_t riggs_pose_mlp_backward_workspace_floats(int32_t depth, int32_t width, int32_t multires) {
  return (size_t)depth * width + (size_t)(depth + 1) * PM_MAX_IN;
}

int riggs_pose_mlp_backward(int32_t depth, int32_t width, int32_t multires, int32_t skip, int32_t n_rot,
                            const float* const* weights, const float* const* biases, const float* W_rot,
                            const float* b_rot, const float* W_tr, const float* b_tr, const float* acts,
                            const float* g_rotation, const float* g_translation, float* workspace,
                            float* flat_grads, riggs_stream stream) {
  PoseMlpDesc d;
  int rc = pm_fill(d, depth, width, multires, skip, n_rot, weights, biases, W_rot, b_rot, W_tr, b_tr);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int emb = 1 + 2 * multires;
  float* dzs = workspace;
  float* dh = workspace + (size_t)depth * width;  // (depth + 1) vectors of PM_MAX_IN: input-gradient of consumer l
  PoseMlpGradDesc g;
  int64_t o = 0;
  int rows = 0;
  for (int l = 0; l < depth; l++) {
    const int in_l = (l == 0) ? emb : ((l - 1 == skip) ? width + emb : width);
    g.w_off[l] = o; o += (int64_t)width * in_l;
    g.b_off[l] = o; o += width;
    g.row_start[l] = rows; rows += width;
  }
  const int in_h = (depth - 1 == skip) ? width + emb : width;
  g.w_off[depth] = o; o += (int64_t)n_rot * in_h;
  g.b_off[depth] = o; o += n_rot;
  g.row_start[depth] = rows; rows += n_rot;
  g.w_off[depth + 1] = o; o += (int64_t)3 * in_h;
  g.b_off[depth + 1] = o; o += 3;
  g.row_start[depth + 1] = rows; rows += 3;
  g.row_start[depth + 2] = rows;
  ProfScope ps(PROF_POSE_BWD, s);
  RIGGS_HIP_CHECK(hipMemsetAsync(dh, 0, (size_t)(depth + 1) * PM_MAX_IN * sizeof(float), s));
  for (int l = depth; l >= 0; l--) {
    const int in_l = (l == 0) ? emb : ((l - 1 == skip) ? width + emb : width);
    const float* dh_out = (l == depth) ? nullptr : dh + (size_t)(l + 1) * PM_MAX_IN;
    hipLaunchKernelGGL(pm_backward_step_kernel, dim3(l == 0 ? 1 : (in_l + 63) / 64, PM_RG), dim3(256), 0, s, d, l, acts,
                       g_rotation, g_translation, dh_out, dh + (size_t)l * PM_MAX_IN, dzs);
  }
  hipLaunchKernelGGL(pm_backward_weights_kernel, dim3(rows), dim3(64), 0, s, d, g, acts, dzs, g_rotation,
                     g_translation, flat_grads);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
