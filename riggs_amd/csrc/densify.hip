// Device-side densification / pruning of the Gaussian cloud — scene/gaussian_model.py of the reference: densify_and_prune
// :500-514 (densify_and_clone :475-498, densify_and_split :440-473, prune_points :373-392 with the optimizer surgery
// _prune_optimizer :355-371 / cat_tensors_to_optimizer :394-417).  The reference builds boolean masks and lets torch index,
// concatenate and re-wrap every parameter and both Adam moments one tensor at a time (~40 launches and as many allocations per
// call); here: ONE selection kernel (the three predicates of a call), a stream compaction per predicate (count per block ->
// scan of the block counts -> ranked scatter) and ONE gather launch that writes every row of every tensor of the new cloud —
// the seven parameter tensors and their fourteen moment tensors — from an index plan, plus a fix-up of the split children.
// HBM-bound row copies: a row is 59 (+ feature) floats and its two moments.
#include "common.h"

namespace riggs {

#define DN_MAX_TENSORS 32

// ---- selection ---------------------------------------------------------------------------------------------------------
// flags[0][n]: the old row survives          = not split and not pruned
// flags[1][n]: a clone of n is made AND kept = clone and not pruned        (a clone has its source's opacity and scale)
// flags[2][n]: the children of n are kept    = split and their own prune test (their scale is the parent's / (0.8 N))
__global__ __launch_bounds__(256) void densify_select_kernel(int N, int S /* columns of _scaling: 3, or 1 isotropic */,
                                                             const float* __restrict__ accum, const float* __restrict__ denom,
                                                             const float* __restrict__ scaling, const float* __restrict__ opacity,
                                                             float grad_threshold, float dense_limit /* percent_dense * extent */,
                                                             float min_opacity, float world_limit /* 0.1 * extent, or < 0: off */,
                                                             float child_div /* 0.8 * N */, unsigned char* __restrict__ flags) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float g = accum[n] / denom[n];
  if (g != g) g = 0.0f;  // grads[grads.isnan()] = 0.0 (:502)
  float smax = expf(scaling[(size_t)n * S]);
  for (int k = 1; k < S; k++) smax = fmaxf(smax, expf(scaling[(size_t)n * S + k]));
  const bool clone = fabsf(g) >= grad_threshold && smax <= dense_limit;   // :478-481 (torch.norm of a 1-vector)
  const bool split = g >= grad_threshold && smax > dense_limit;           // :447-450
  const float op = 1.0f / (1.0f + expf(-opacity[n]));
  const bool prune_self = op < min_opacity || (world_limit >= 0.0f && smax > world_limit);   // :507-512 (max_radii2D is all
  // zeros by then — densification_postfix has reset it — so the screen-size test of :510 never fires)
  const float child = smax / child_div;
  const bool prune_child = op < min_opacity || (world_limit >= 0.0f && child > world_limit);
  flags[n] = (!split && !prune_self) ? 1 : 0;
  flags[(size_t)N + n] = (clone && !prune_self) ? 1 : 0;
  flags[2 * (size_t)N + n] = (split && !prune_child) ? 1 : 0;
}

// ---- stream compaction: indices of the set flags, ascending ------------------------------------------------------------
__global__ __launch_bounds__(256) void compact_count_kernel(int N, const unsigned char* __restrict__ flags, uint32_t* __restrict__ block_count) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const bool f = n < N && flags[n] != 0;
  const uint64_t m = __builtin_amdgcn_ballot_w64(f);
  __shared__ uint32_t s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = (uint32_t)__builtin_popcountll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
// one workgroup: exclusive scan of the block counts in place, the total into *count
__global__ __launch_bounds__(1024) void compact_scan_kernel(int n_blocks, uint32_t* __restrict__ block_count, int32_t* __restrict__ count) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0u;
  __syncthreads();
  for (int base = 0; base < n_blocks; base += 1024) {
    const int i = base + tid;
    const uint32_t c = i < n_blocks ? block_count[i] : 0u;
    uint32_t v = c;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)v, o); if (lane >= o) v += u; }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    uint32_t off = s_carry;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    if (i < n_blocks) block_count[i] = off + v - c;
    __syncthreads();
    if (tid == 1023) s_carry = off + v;
    __syncthreads();
  }
  if (tid == 0) *count = (int32_t)s_carry;
}
__global__ __launch_bounds__(256) void compact_scatter_kernel(int N, const unsigned char* __restrict__ flags, const uint32_t* __restrict__ block_start,
                                                              int32_t* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const bool f = n < N && flags[n] != 0;
  const uint64_t m = __builtin_amdgcn_ballot_w64(f);
  __shared__ uint32_t s[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s[wave] = (uint32_t)__builtin_popcountll(m);
  __syncthreads();
  uint32_t off = block_start[blockIdx.x];
  for (int w = 0; w < wave; w++) off += s[w];
  if (f) out[off + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = n;
}

// ---- the gather: every row of every tensor of the new cloud ------------------------------------------------------------
struct GatherArgs {
  const float* src[DN_MAX_TENSORS];
  float* dst[DN_MAX_TENSORS];
  int width[DN_MAX_TENSORS];               // floats per row
  unsigned char zero_new[DN_MAX_TENSORS];  // 1: a NEW row (plan entry < 0: ~source) gets zeros (the Adam moments), 0: its source's row
  int n_out;
  const int32_t* plan;
};
__global__ __launch_bounds__(256) void rows_gather_kernel(GatherArgs a) {
  const int t = blockIdx.y;
  const int w = a.width[t];
  const float* __restrict__ src = a.src[t];
  float* __restrict__ dst = a.dst[t];
  const bool zn = a.zero_new[t] != 0;
  const size_t total = (size_t)a.n_out * w;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int m = (int)(e / w), c = (int)(e - (size_t)m * w);
    const int32_t p = a.plan[m];
    const int s = p >= 0 ? p : ~p;
    dst[e] = (p < 0 && zn) ? 0.0f : src[(size_t)s * w + c];
  }
}

// ---- split children (densify_and_split :452-460): position = R(q_parent) (z * scale_parent) + xyz_parent, scale / (0.8 N) ----
__global__ __launch_bounds__(256) void split_children_kernel(int n_children /* parents kept x copies */, int n_parents, int S,
                                                             const int32_t* __restrict__ parents, const float* __restrict__ z /* (n_children, 3) unit normals */,
                                                             const float* __restrict__ xyz, const float* __restrict__ scaling,
                                                             const float* __restrict__ rotation, float child_div,
                                                             float* __restrict__ new_xyz /* the children's rows */, float* __restrict__ new_scaling) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_children) return;
  const int p = parents[j % n_parents];
  float sc[3];
  for (int k = 0; k < 3; k++) sc[k] = expf(scaling[(size_t)p * S + (S == 1 ? 0 : k)]);   // get_scaling (isotropic: column 0 repeated)
  const float q0 = rotation[4 * (size_t)p], q1 = rotation[4 * (size_t)p + 1], q2 = rotation[4 * (size_t)p + 2], q3 = rotation[4 * (size_t)p + 3];
  const float nq = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);   // build_rotation (utils/general_utils.py:137-158)
  const float r = q0 / nq, x = q1 / nq, y = q2 / nq, zq = q3 / nq;
  const float s0 = z[3 * (size_t)j] * sc[0], s1 = z[3 * (size_t)j + 1] * sc[1], s2 = z[3 * (size_t)j + 2] * sc[2];
  new_xyz[3 * (size_t)j] = (1 - 2 * (y * y + zq * zq)) * s0 + 2 * (x * y - r * zq) * s1 + 2 * (x * zq + r * y) * s2 + xyz[3 * (size_t)p];
  new_xyz[3 * (size_t)j + 1] = 2 * (x * y + r * zq) * s0 + (1 - 2 * (x * x + zq * zq)) * s1 + 2 * (y * zq - r * x) * s2 + xyz[3 * (size_t)p + 1];
  new_xyz[3 * (size_t)j + 2] = 2 * (x * zq - r * y) * s0 + 2 * (y * zq + r * x) * s1 + (1 - 2 * (x * x + y * y)) * s2 + xyz[3 * (size_t)p + 2];
  for (int k = 0; k < S; k++) new_scaling[(size_t)j * S + k] = logf(sc[k] / child_div);
}

}  // namespace riggs

using namespace riggs;

extern "C" {

int riggs_densify_select(int32_t N, int32_t scaling_columns, const float* xyz_gradient_accum, const float* denom, const float* scaling,
                         const float* opacity, float grad_threshold, float dense_limit, float min_opacity, float world_limit,
                         float child_div, uint8_t* flags /* (3, N) */, riggs_stream stream) {
  RIGGS_REQUIRE(N >= 0 && (scaling_columns == 1 || scaling_columns == 3), "riggs_densify_select: bad sizes");
  if (N == 0) return 0;
  RIGGS_REQUIRE(xyz_gradient_accum && denom && scaling && opacity && flags, "riggs_densify_select: NULL argument");
  hipLaunchKernelGGL(densify_select_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, scaling_columns, xyz_gradient_accum,
                     denom, scaling, opacity, grad_threshold, dense_limit, min_opacity, world_limit, child_div, flags);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

size_t riggs_compact_workspace_bytes(int32_t N) { return align_up((size_t)((N > 0 ? N : 1) + 255) / 256 * 4); }

int riggs_compact_indices(int32_t N, const uint8_t* flags, int32_t* out_indices /* capacity N */, int32_t* count /* device */,
                          void* workspace, riggs_stream stream) {
  RIGGS_REQUIRE(N >= 0 && count && workspace, "riggs_compact_indices: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) { RIGGS_HIP_CHECK(hipMemsetAsync(count, 0, 4, s)); return 0; }
  RIGGS_REQUIRE(flags && out_indices, "riggs_compact_indices: NULL argument");
  const int nb = (N + 255) / 256;
  uint32_t* bc = (uint32_t*)workspace;
  hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(256), 0, s, N, flags, bc);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, nb, bc, count);
  hipLaunchKernelGGL(compact_scatter_kernel, dim3(nb), dim3(256), 0, s, N, flags, bc, out_indices);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_rows_gather(int32_t n_out, const int32_t* plan, int32_t n_tensors, const float* const* src, float* const* dst,
                      const int32_t* row_floats, const uint8_t* zero_new, riggs_stream stream) {
  RIGGS_REQUIRE(n_out >= 0 && n_tensors >= 1 && n_tensors <= DN_MAX_TENSORS, "riggs_rows_gather: 1..32 tensors");
  if (n_out == 0) return 0;
  RIGGS_REQUIRE(plan && src && dst && row_floats && zero_new, "riggs_rows_gather: NULL argument");
  GatherArgs a{};
  int wmax = 1;
  for (int t = 0; t < n_tensors; t++) {
    RIGGS_REQUIRE(src[t] && dst[t] && row_floats[t] >= 1, "riggs_rows_gather: bad tensor");
    a.src[t] = src[t]; a.dst[t] = dst[t]; a.width[t] = row_floats[t]; a.zero_new[t] = zero_new[t];
    wmax = row_floats[t] > wmax ? row_floats[t] : wmax;
  }
  a.n_out = n_out; a.plan = plan;
  size_t blocks = ((size_t)n_out * wmax + 255) / 256;
  if (blocks > 4096) blocks = 4096;  // grid-stride: ~16 workgroups per CU per tensor row of the grid
  hipLaunchKernelGGL(rows_gather_kernel, dim3((unsigned)blocks, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, a);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_split_children(int32_t n_children, int32_t n_parents, int32_t scaling_columns, const int32_t* parents, const float* unit_normals,
                         const float* xyz, const float* scaling, const float* rotation, float child_div, float* new_xyz,
                         float* new_scaling, riggs_stream stream) {
  RIGGS_REQUIRE(n_children >= 0 && (scaling_columns == 1 || scaling_columns == 3), "riggs_split_children: bad sizes");
  if (n_children == 0) return 0;
  RIGGS_REQUIRE(n_parents > 0 && parents && unit_normals && xyz && scaling && rotation && new_xyz && new_scaling, "riggs_split_children: NULL argument");
  hipLaunchKernelGGL(split_children_kernel, dim3((n_children + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_children, n_parents,
                     scaling_columns, parents, unit_normals, xyz, scaling, rotation, child_div, new_xyz, new_scaling);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
