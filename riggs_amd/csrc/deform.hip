// Skeleton-driven deformation of the Gaussian cloud:
//   fk_*   : forward kinematics over <= 64 joints on ONE wave64 (replaces ~3*J torch launches
//            of SkeletonWarp.chain_product_transform, skeleton_warp.py:242-273)
//   lbs_*  : bone-distance skinning weights + linear blend skinning fused per Gaussian, with the
//            <= 63 bone records (segment, radius, 3x4 transform, quaternion) staged in LDS.
// Backward reduces over the N Gaussians inside the kernel: wave64 DPP sums -> LDS -> one atomic
// per workgroup per output.
#include "fk_device.h"

namespace riggs {

#define LOG2E 1.4426950408889634f
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }

__global__ __launch_bounds__(64) void fk_forward_kernel(int J, const float* __restrict__ local_rot,
                                                        const float* __restrict__ joints,
                                                        const int32_t* __restrict__ parents,
                                                        const float* __restrict__ global_trans,
                                                        float* __restrict__ transforms, float* __restrict__ node_rot,
                                                        float* __restrict__ d_nodes) {
  FkIn in;
  fk_load(J, local_rot, joints, parents, nullptr, nullptr, in);
  FkLane f;
  fk_wave_forward(J, in, f);
  const int j = threadIdx.x;
  if (j < J) {
    const float x = in.x[0], y = in.x[1], z = in.x[2];
#pragma unroll
    for (int e = 0; e < 12; e++) transforms[12 * j + e] = f.G[e];
#pragma unroll
    for (int r = 0; r < 3; r++)
      d_nodes[3 * j + r] = (f.G[4 * r] * x + f.G[4 * r + 1] * y + f.G[4 * r + 2] * z + f.G[4 * r + 3]) + global_trans[r];
    float q[4];
    R_to_quat(f.G, q);
#pragma unroll
    for (int e = 0; e < 4; e++) node_rot[4 * j + e] = q[e];
  }
}

__global__ __launch_bounds__(64) void fk_backward_kernel(int J, const float* __restrict__ local_rot,
                                                         const float* __restrict__ joints,
                                                         const int32_t* __restrict__ parents,
                                                         const float* __restrict__ dL_dG_in,
                                                         const float* __restrict__ dL_dnodes,
                                                         float* __restrict__ dL_dlocal_rot,
                                                         float* __restrict__ dL_dglobal_trans) {
  FkIn in;
  fk_load(J, local_rot, joints, parents, dL_dG_in, dL_dnodes, in);
  FkLane f;
  fk_wave_forward(J, in, f);
  const int j = threadIdx.x;
  if (dL_dnodes && j < 3) {  // d_nodes = posed + global_trans
    float s = 0.f;
    for (int k = 0; k < J; k++) s += dL_dnodes[3 * k + j];
    dL_dglobal_trans[j] += s;
  }
  float dq[4];
  fk_wave_backward(J, in, f, dq);
  if (j < J) {
#pragma unroll
    for (int e = 0; e < 4; e++) dL_dlocal_rot[4 * j + e] = dq[e];
  }
}

// ------------------------------------------------------------------------------------ LBS
struct Bone {        // 24 floats, LDS resident
  float a[3];        // parent joint (segment start)  skeleton_warp.py:208-209
  float len2c;       // max(|b-a|^2, 1e-6)            :226
  float ba[3];       // b - a
  float inv2r2;      // 1 / (2 exp(rho)^2)            :65-66
  float G[12];       // global transform of the CHILD joint
  float q[4];        // node_rot of the child joint (detached)
  float rl2;         // 1 / len2c (the all-bones path: no top-K ties to reproduce, so no IEEE division per bone and Gaussian)
  float pad_[3];     // (records stay multiples of 16 bytes: whole-record ds_read_b128)
};

struct LbsArgs {
  int N, J, K;
  const float *x, *joints, *node_radius_log, *transforms, *node_rot, *global_trans, *motion_mask;
  const float* weight_mod;  // (N, J-1) or NULL: sigmoid(WeightMLP(x)) multiplying the kernel weights (skeleton_warp.py:56-69)
  float* dmod;              // backward: dL/dweight_mod (N, J-1) or NULL
  const int32_t* parents;
  float *d_xyz, *d_rot, *nn_weight;
  int64_t* nn_idx;
  // backward
  const float *g_xyz, *g_rot;
  float *dG, *drho, *dgt, *dmask;
  float* partial;  // [workgroups][(J-1)*13 + 3] per-workgroup sums (deterministic two-stage reduction)
  // forward with the kinematic chain inside (riggs_lbs_forward_fk): the pose, and where workgroup 0 leaves the chain's results
  const float* local_rot;
  float *fk_transforms, *fk_node_rot, *fk_d_nodes;
  int mod_lds;  // forward: weight_mod is staged through LDS (the launch reserved the tile)
};

// (transforms / node_rot: global memory, or the LDS arrays of a kinematic chain that this workgroup ran itself)
__device__ __forceinline__ void stage_bones(const LbsArgs& a, Bone* bones, const float* transforms, const float* node_rot);
__device__ __forceinline__ void stage_bones(const LbsArgs& a, Bone* bones) { stage_bones(a, bones, a.transforms, a.node_rot); }
__device__ __forceinline__ void stage_bones(const LbsArgs& a, Bone* bones, const float* transforms, const float* node_rot) {
  for (int k = threadIdx.x; k < a.J - 1; k += blockDim.x) {
    const int child = k + 1, par = a.parents[child];
    Bone b;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      b.a[c] = a.joints[3 * par + c];
      b.ba[c] = a.joints[3 * child + c] - b.a[c];
    }
    const float l2 = __fadd_rn(__fadd_rn(__fmul_rn(b.ba[0], b.ba[0]), __fmul_rn(b.ba[1], b.ba[1])), __fmul_rn(b.ba[2], b.ba[2]));
    b.len2c = fmaxf(l2, 1e-6f);
    b.rl2 = 1.0f / b.len2c;
    b.pad_[0] = 0.f; b.pad_[1] = 0.f; b.pad_[2] = 0.f;
    const float rad = expf(a.node_radius_log[child]);
    b.inv2r2 = 1.0f / (2.0f * rad * rad);
#pragma unroll
    for (int e = 0; e < 12; e++) b.G[e] = transforms[12 * child + e];
#pragma unroll
    for (int e = 0; e < 4; e++) b.q[e] = node_rot[4 * child + e];
    bones[k] = b;
  }
  __syncthreads();
}

// line_segment_distance (skeleton_warp.py:215-238), squared.  Written in the reference's own
// operation order with FP contraction off: bones that share a joint give near-tied distances
// and the top-K selection (K > 0) must break those ties the way the reference's arithmetic does.
__device__ __forceinline__ float bone_d2(const Bone& b, float px, float py, float pz) {
#pragma clang fp contract(off)
  const float ex = px - b.a[0], ey = py - b.a[1], ez = pz - b.a[2];
  float t = ((ex * b.ba[0] + ey * b.ba[1]) + ez * b.ba[2]) / b.len2c;
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  const float sx = (b.a[0] + t * b.ba[0]) - px, sy = (b.a[1] + t * b.ba[1]) - py, sz = (b.a[2] + t * b.ba[2]) - pz;
  return (sx * sx + sy * sy) + sz * sz;
}

// The same distance for the all-bones configuration (K = -1): nothing is selected by comparing distances there, so the
// division becomes a multiplication by the staged reciprocal and contraction is allowed (an IEEE division is ~10 of this
// function's ~30 instructions, and it runs once per bone and Gaussian).
__device__ __forceinline__ float bone_d2_fast(const Bone& b, float px, float py, float pz) {
  const float ex = px - b.a[0], ey = py - b.a[1], ez = pz - b.a[2];
  const float t = __builtin_amdgcn_fmed3f((ex * b.ba[0] + ey * b.ba[1] + ez * b.ba[2]) * b.rl2, 0.0f, 1.0f);
  const float sx = t * b.ba[0] - ex, sy = t * b.ba[1] - ey, sz = t * b.ba[2] - ez;
  return sx * sx + sy * sy + sz * sz;
}

// Top-K selection (K > 0): bitmask of the K bones with the smallest (d2, index), found by K
// selection passes over the <= 63 bones.  Already-picked bones are excluded through the mask, so
// no floating-point value is ever compared for equality across call sites.
__device__ __forceinline__ uint64_t topk_mask(const Bone* bones, int B, int K, float px, float py, float pz) {
  uint64_t mask = 0;
  for (int s = 0; s < K; s++) {
    float best = INFINITY;
    int bi = -1;
    for (int k = 0; k < B; k++) {
      if ((mask >> k) & 1ull) continue;
      const float d2 = bone_d2(bones[k], px, py, pz);
      if (d2 < best || bi < 0) { best = d2; bi = k; }
    }
    mask |= 1ull << bi;
  }
  return mask;
}

#define LBS_PTS2_MIN_N 1000000
#define LBS_PTS2_MIN_J 16
// FK: the workgroup runs the kinematic chain itself (fk_device.h: ~1 us on its first wave, against a launch of its own in front
// of this one — every workgroup repeats it, workgroup 0 keeps the results)
template <bool TOPK, bool FK, int PTS>
__global__ __launch_bounds__(256) void lbs_forward_kernel(LbsArgs a) {
  __shared__ Bone bones[MAX_J - 1];
  // (the Gaussians' positions first: their round trip runs under the chain and the staging — the compiler keeps loads behind
  // the barriers where it finds them)
  static_assert(!TOPK || PTS == 1, "the top-K path takes one Gaussian per thread");
  const int n0 = blockIdx.x * (256 * PTS) + threadIdx.x;
  float px[PTS], py[PTS], pz[PTS];
  bool on[PTS];
#pragma unroll
  for (int p = 0; p < PTS; p++) {
    const int n = n0 + 256 * p;
    on[p] = n < a.N;
    const int nn = min(on[p] ? n : n0, a.N - 1);
    px[p] = a.x[3 * nn]; py[p] = a.x[3 * nn + 1]; pz[p] = a.x[3 * nn + 2];
  }
  // weight_mod (WeightMLP head on): a thread reads its own row of B floats, bone by bone — as global loads that is B
  // instructions of 64 addresses 4 B bytes apart each (46 lines per instruction, the L1 thrashes: +20 us on this kernel).  The
  // wave's 64 rows are one contiguous run: copied once, coalesced, into a wave-private LDS tile (row stride odd: a column
  // read is conflict-free), where the bone loop reads them.  (a.mod_lds: the launch reserved 256 (B | 1) floats)
  extern __shared__ float s_mod_dyn[];
  const int BP = (a.J - 1) | 1;
  float* s_mod = s_mod_dyn + (threadIdx.x >> 6) * 64 * BP;
  if (PTS == 1 && a.mod_lds) {
    const int Bm = a.J - 1, lane_ = threadIdx.x & 63;
    const int ng = blockIdx.x * 256 + (threadIdx.x & ~63);
    const int n_el = max(0, min(64, a.N - ng)) * Bm;
    const float* run = a.weight_mod + (size_t)ng * Bm;
    const float invB = 1.0f / (float)Bm;
    for (int e = lane_; e < n_el; e += 64) {
      const int r = (int)(((float)e + 0.5f) * invB);
      s_mod[r * BP + (e - r * Bm)] = run[e];
    }
  }
  if constexpr (FK) {
    __shared__ float G[MAX_J][12];
    __shared__ float Q[MAX_J][4];
    if (threadIdx.x < 64) {  // (wave 0: one joint per lane)
      FkIn in;
      fk_load(a.J, a.local_rot, a.joints, a.parents, nullptr, nullptr, in);
      FkLane f;
      fk_wave_forward(a.J, in, f);
      const int j = threadIdx.x;
      if (j < a.J) {
        float q[4];
        R_to_quat(f.G, q);
#pragma unroll
        for (int e = 0; e < 12; e++) G[j][e] = f.G[e];
#pragma unroll
        for (int e = 0; e < 4; e++) Q[j][e] = q[e];
        if (blockIdx.x == 0) {
          const float x = in.x[0], y = in.x[1], z = in.x[2];
#pragma unroll
          for (int e = 0; e < 12; e++) a.fk_transforms[12 * j + e] = f.G[e];
#pragma unroll
          for (int r = 0; r < 3; r++)
            a.fk_d_nodes[3 * j + r] = (f.G[4 * r] * x + f.G[4 * r + 1] * y + f.G[4 * r + 2] * z + f.G[4 * r + 3]) + a.global_trans[r];
#pragma unroll
          for (int e = 0; e < 4; e++) a.fk_node_rot[4 * j + e] = q[e];
        }
      }
    }
    __syncthreads();
    stage_bones(a, bones, &G[0][0], &Q[0][0]);
  } else {
    stage_bones(a, bones);
  }
  // PTS Gaussians per thread (all-bones configuration): what bounds this loop at 64 joints is the LDS handing every wave the
  // bone records (24 floats per bone, broadcast: 8 cycles of the CU's 128 B / clk return path per ds_read_b128 and wave,
  // ~48 per bone and wave against ~26 VALU instructions per SIMD) — two Gaussians per lane read each record once for both
  // (2 M x 63: 148 -> 133 us; the kernel is then bound by vector issue: ~32 instructions per Gaussian and bone, 16 of them
  // the blend.  The blend on the matrix pipe instead — v_mfma_f32_16x16x4_f32, lane = (Gaussian i of a tile of 16, bone
  // 4 s + j): the same k-ordered fmaf chain, 19 vector instructions per pair — measured 128 us: its four dependent-free MFMAs
  // per step still cost the wave their issue, plus a transposing epilogue through LDS; not kept for 4 %).
  // (one block of 256 * PTS Gaussians per workgroup: fewer, looping workgroups — the chain and the staging above once per
  // workgroup instead of once per block — are slower: 17.5 -> 19.7 -> 24.6 us at 1172 / 586 / 293 workgroups)
  const int B = a.J - 1;
  if (n0 >= a.N) return;
  const uint64_t selmask = TOPK ? topk_mask(bones, B, a.K, px[0], py[0], pz[0]) : ~0ull;
  float M[PTS][12], qa[PTS][4], sum[PTS];
#pragma unroll
  for (int p = 0; p < PTS; p++) {
    sum[p] = 0.f;
#pragma unroll
    for (int e = 0; e < 12; e++) M[p][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) qa[p][e] = 0.f;
  }
  for (int k = 0; k < B; k++) {
    const Bone& b = bones[k];
    if (TOPK && !((selmask >> k) & 1ull)) continue;
#pragma unroll
    for (int p = 0; p < PTS; p++) {
      const float d2 = TOPK ? bone_d2(b, px[p], py[p], pz[p]) : bone_d2_fast(b, px[p], py[p], pz[p]);
      float u = fast_exp(-d2 * b.inv2r2);                // skeleton_warp.py:66
      if (a.weight_mod) u *= (PTS == 1 && a.mod_lds) ? s_mod[(threadIdx.x & 63) * BP + k]
                                                     : a.weight_mod[(size_t)(on[p] ? n0 + 256 * p : n0) * B + k];  // :68-69
      const float v = u + 1e-7f;                          // :71
      sum[p] += v;
#pragma unroll
      for (int e = 0; e < 12; e++) M[p][e] += v * b.G[e];
#pragma unroll
      for (int e = 0; e < 4; e++) qa[p][e] += v * b.q[e];
    }
  }
  const float gx = a.global_trans[0], gy = a.global_trans[1], gz = a.global_trans[2];
#pragma unroll
  for (int p = 0; p < PTS; p++) {
    if (!on[p]) continue;
    const int n = n0 + 256 * p;
    const float inv = 1.0f / sum[p];
    const float m = a.motion_mask ? a.motion_mask[n] : 1.0f;
    const float ax = (M[p][0] * px[p] + M[p][1] * py[p] + M[p][2] * pz[p] + M[p][3]) * inv + gx;
    const float ay = (M[p][4] * px[p] + M[p][5] * py[p] + M[p][6] * pz[p] + M[p][7]) * inv + gy;
    const float az = (M[p][8] * px[p] + M[p][9] * py[p] + M[p][10] * pz[p] + M[p][11]) * inv + gz;
    a.d_xyz[3 * n] = (ax - px[p]) * m; a.d_xyz[3 * n + 1] = (ay - py[p]) * m; a.d_xyz[3 * n + 2] = (az - pz[p]) * m;
    reinterpret_cast<float4*>(a.d_rot)[n] = make_float4(qa[p][0] * inv * m, qa[p][1] * inv * m, qa[p][2] * inv * m, qa[p][3] * inv * m);
    if (a.nn_weight || a.nn_idx) {
      if (TOPK) {
        // ascending-d2 order like torch.topk(largest=False): K selection passes inside the mask
        uint64_t left = selmask;
        for (int s = 0; s < a.K; s++) {
          float best = INFINITY; int bi = -1;
          for (int k = 0; k < B; k++) {
            if (!((left >> k) & 1ull)) continue;
            const float d2 = bone_d2(bones[k], px[p], py[p], pz[p]);
            if (d2 < best || bi < 0) { best = d2; bi = k; }
          }
          left &= ~(1ull << bi);
          if (a.nn_weight) a.nn_weight[(size_t)n * a.K + s] = (fast_exp(-best * bones[bi].inv2r2) + 1e-7f) * inv;
          if (a.nn_idx) a.nn_idx[(size_t)n * a.K + s] = bi + 1;
        }
      } else {
        for (int k = 0; k < B; k++) {
          if (a.nn_weight) a.nn_weight[(size_t)n * B + k] = (fast_exp(-bone_d2_fast(bones[k], px[p], py[p], pz[p]) * bones[k].inv2r2) *
                                                                (a.weight_mod ? a.weight_mod[(size_t)n * B + k] : 1.0f) + 1e-7f) * inv;
          if (a.nn_idx) a.nn_idx[(size_t)n * B + k] = k + 1;
        }
      }
    }
  }
}

// ---- the all-bones forward with the bone records in SGPRs (large scenes; riggs_set_option("lbs_scalar")).  A bone
// record is the same for every lane: here it comes from a table in global memory through the scalar cache (the constant address
// space makes the uniform loads s_load_dwordx8 / x4) and feeds the vector instructions as scalar operands — no LDS image, no
// barrier, no ds_read_b128 broadcast per bone and wave; the results are bit-identical to the LDS form's (same operations, same
// order).  The table (the staged Bone records) is written by a one-workgroup launch in front: chosen by lbs_use_scalar.
__global__ __launch_bounds__(64) void lbs_bone_table_kernel(LbsArgs a, Bone* __restrict__ table) {
  __shared__ Bone bones[MAX_J - 1];
  stage_bones(a, bones);
  for (int k = threadIdx.x; k < a.J - 1; k += 64) table[k] = bones[k];
}
// ... and with the kinematic chain in front of it (riggs_lbs_forward_fk's large-scene form): one wave runs the chain, leaves its
// results where the backward reads them, and the table
__global__ __launch_bounds__(64) void lbs_fk_table_kernel(LbsArgs a, Bone* __restrict__ table) {
  __shared__ Bone bones[MAX_J - 1];
  __shared__ float G[MAX_J][12];
  __shared__ float Q[MAX_J][4];
  FkIn in;
  fk_load(a.J, a.local_rot, a.joints, a.parents, nullptr, nullptr, in);
  FkLane f;
  fk_wave_forward(a.J, in, f);
  const int j = threadIdx.x;
  if (j < a.J) {
    float q[4];
    R_to_quat(f.G, q);
    const float x = in.x[0], y = in.x[1], z = in.x[2];
#pragma unroll
    for (int e = 0; e < 12; e++) { G[j][e] = f.G[e]; a.fk_transforms[12 * j + e] = f.G[e]; }
#pragma unroll
    for (int r = 0; r < 3; r++)
      a.fk_d_nodes[3 * j + r] = (f.G[4 * r] * x + f.G[4 * r + 1] * y + f.G[4 * r + 2] * z + f.G[4 * r + 3]) + a.global_trans[r];
#pragma unroll
    for (int e = 0; e < 4; e++) { Q[j][e] = q[e]; a.fk_node_rot[4 * j + e] = q[e]; }
  }
  __syncthreads();
  stage_bones(a, bones, &G[0][0], &Q[0][0]);
  for (int k = threadIdx.x; k < a.J - 1; k += 64) table[k] = bones[k];
}
typedef __attribute__((address_space(4))) const Bone* LbsConstBones;
template <int PTS>
__global__ __launch_bounds__(256) void lbs_forward_scalar_kernel(LbsArgs a, const Bone* __restrict__ table_g) {
  const LbsConstBones table = (LbsConstBones)table_g;
  const int n0 = blockIdx.x * (256 * PTS) + threadIdx.x;
  float px[PTS], py[PTS], pz[PTS];
  bool on[PTS];
#pragma unroll
  for (int p = 0; p < PTS; p++) {
    const int n = n0 + 256 * p;
    on[p] = n < a.N;
    const int nn = min(on[p] ? n : n0, a.N - 1);
    px[p] = a.x[3 * nn]; py[p] = a.x[3 * nn + 1]; pz[p] = a.x[3 * nn + 2];
  }
  const int B = a.J - 1;
  if (n0 >= a.N) return;
  float M[PTS][12], qa[PTS][4], sum[PTS];
#pragma unroll
  for (int p = 0; p < PTS; p++) {
    sum[p] = 0.f;
#pragma unroll
    for (int e = 0; e < 12; e++) M[p][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) qa[p][e] = 0.f;
  }
  for (int k = 0; k < B; k++) {
    const float a0 = table[k].a[0], a1 = table[k].a[1], a2 = table[k].a[2];
    const float b0 = table[k].ba[0], b1 = table[k].ba[1], b2 = table[k].ba[2];
    const float rl2 = table[k].rl2, i2r = table[k].inv2r2;
    float G[12], q[4];
#pragma unroll
    for (int e = 0; e < 12; e++) G[e] = table[k].G[e];
#pragma unroll
    for (int e = 0; e < 4; e++) q[e] = table[k].q[e];
#pragma unroll
    for (int p = 0; p < PTS; p++) {
      const float ex = px[p] - a0, ey = py[p] - a1, ez = pz[p] - a2;
      const float t = __builtin_amdgcn_fmed3f((ex * b0 + ey * b1 + ez * b2) * rl2, 0.0f, 1.0f);
      const float sx = t * b0 - ex, sy = t * b1 - ey, sz = t * b2 - ez;
      const float d2 = sx * sx + sy * sy + sz * sz;
      float u = fast_exp(-d2 * i2r);
      if (a.weight_mod) u *= a.weight_mod[(size_t)(on[p] ? n0 + 256 * p : n0) * B + k];
      const float v = u + 1e-7f;
      sum[p] += v;
#pragma unroll
      for (int e = 0; e < 12; e++) M[p][e] += v * G[e];
#pragma unroll
      for (int e = 0; e < 4; e++) qa[p][e] += v * q[e];
    }
  }
  const float gx = a.global_trans[0], gy = a.global_trans[1], gz = a.global_trans[2];
#pragma unroll
  for (int p = 0; p < PTS; p++) {
    if (!on[p]) continue;
    const int n = n0 + 256 * p;
    const float inv = 1.0f / sum[p];
    const float m = a.motion_mask ? a.motion_mask[n] : 1.0f;
    const float ax = (M[p][0] * px[p] + M[p][1] * py[p] + M[p][2] * pz[p] + M[p][3]) * inv + gx;
    const float ay = (M[p][4] * px[p] + M[p][5] * py[p] + M[p][6] * pz[p] + M[p][7]) * inv + gy;
    const float az = (M[p][8] * px[p] + M[p][9] * py[p] + M[p][10] * pz[p] + M[p][11]) * inv + gz;
    a.d_xyz[3 * n] = (ax - px[p]) * m; a.d_xyz[3 * n + 1] = (ay - py[p]) * m; a.d_xyz[3 * n + 2] = (az - pz[p]) * m;
    reinterpret_cast<float4*>(a.d_rot)[n] = make_float4(qa[p][0] * inv * m, qa[p][1] * inv * m, qa[p][2] * inv * m, qa[p][3] * inv * m);
  }
}

// Backward.  Per bone 13 sums over the Gaussians: dG_k (12) = sum_n w_nk * (ghat_n (x) [x_n;1]) and
// drho_k = sum_n dL/dv_nk * u_nk * d2_nk * exp(-2 rho_k).
// This thread-per-Gaussian kernel serves the top-K configuration (K > 0, skeleton_warp.py:46-49: every Gaussian has its
// own bone subset); the all-bones default (K = -1) runs the bone-lane kernel below.
__global__ __launch_bounds__(256) void lbs_backward_kernel(LbsArgs a) {
  __shared__ Bone bones[MAX_J - 1];
  __shared__ float s_acc[MAX_J - 1][13];
  __shared__ float s_gt[3];
  stage_bones(a, bones);
  const int B = a.J - 1;
  for (int e = threadIdx.x; e < B * 13; e += 256) (&s_acc[0][0])[e] = 0.f;
  if (threadIdx.x < 3) s_gt[threadIdx.x] = 0.f;
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  const bool valid = n < a.N;
  const int lane = threadIdx.x & 63;
  float px = 0.f, py = 0.f, pz = 0.f, m = 0.f;
  float g[3] = {0.f, 0.f, 0.f}, h[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
    px = a.x[3 * n]; py = a.x[3 * n + 1]; pz = a.x[3 * n + 2];
    m = a.motion_mask ? a.motion_mask[n] : 1.0f;
    g[0] = a.g_xyz[3 * n]; g[1] = a.g_xyz[3 * n + 1]; g[2] = a.g_xyz[3 * n + 2];
    const float4 hh = reinterpret_cast<const float4*>(a.g_rot)[n];
    h[0] = hh.x; h[1] = hh.y; h[2] = hh.z; h[3] = hh.w;
  }
  const float gh[3] = {g[0] * m, g[1] * m, g[2] * m};
  const float hh4[4] = {h[0] * m, h[1] * m, h[2] * m, h[3] * m};
  // pass 1: normaliser and blended outputs
  const uint64_t selmask = a.K > 0 ? topk_mask(bones, B, a.K, px, py, pz) : ~0ull;
  float M[12], qa[4] = {0.f, 0.f, 0.f, 0.f}, sum = 0.f;
#pragma unroll
  for (int e = 0; e < 12; e++) M[e] = 0.f;
  for (int k = 0; k < B; k++) {
    const Bone& b = bones[k];
    if (!((selmask >> k) & 1ull)) continue;
    const float d2 = bone_d2(b, px, py, pz);
    const float v = fast_exp(-d2 * b.inv2r2) + 1e-7f;
    sum += v;
#pragma unroll
    for (int e = 0; e < 12; e++) M[e] += v * b.G[e];
#pragma unroll
    for (int e = 0; e < 4; e++) qa[e] += v * b.q[e];
  }
  const float inv = valid ? 1.0f / sum : 0.f;
  const float ax = (M[0] * px + M[1] * py + M[2] * pz + M[3]) * inv;
  const float ay = (M[4] * px + M[5] * py + M[6] * pz + M[7]) * inv;
  const float az = (M[8] * px + M[9] * py + M[10] * pz + M[11]) * inv;
  const float qv[4] = {qa[0] * inv, qa[1] * inv, qa[2] * inv, qa[3] * inv};
  // S = sum_j w_j dL/dw_j
  const float S = gh[0] * ax + gh[1] * ay + gh[2] * az + hh4[0] * qv[0] + hh4[1] * qv[1] + hh4[2] * qv[2] + hh4[3] * qv[3];
  if (valid && a.dmask) {
    const float gxs = a.global_trans[0], gys = a.global_trans[1], gzs = a.global_trans[2];
    a.dmask[n] = g[0] * (ax + gxs - px) + g[1] * (ay + gys - py) + g[2] * (az + gzs - pz) + h[0] * qv[0] + h[1] * qv[1] +
                 h[2] * qv[2] + h[3] * qv[3];
  }
  // pass 2: per-bone contributions, reduced over the wave
  const float P[12] = {gh[0] * px, gh[0] * py, gh[0] * pz, gh[0], gh[1] * px, gh[1] * py, gh[1] * pz, gh[1],
                       gh[2] * px, gh[2] * py, gh[2] * pz, gh[2]};
  for (int k = 0; k < B; k++) {
    const Bone& b = bones[k];
    const float d2 = bone_d2(b, px, py, pz);
    float w = 0.f, r = 0.f;
    const bool sel = valid && ((selmask >> k) & 1ull);
    if (sel) {
      const float u = fast_exp(-d2 * b.inv2r2);
      w = (u + 1e-7f) * inv;
      const float Ax = b.G[0] * px + b.G[1] * py + b.G[2] * pz + b.G[3];
      const float Ay = b.G[4] * px + b.G[5] * py + b.G[6] * pz + b.G[7];
      const float Az = b.G[8] * px + b.G[9] * py + b.G[10] * pz + b.G[11];
      const float dLdw = gh[0] * Ax + gh[1] * Ay + gh[2] * Az + hh4[0] * b.q[0] + hh4[1] * b.q[1] + hh4[2] * b.q[2] + hh4[3] * b.q[3];
      const float dLdv = (dLdw - S) * inv;
      r = dLdv * u * d2 * (2.0f * b.inv2r2);
    }
    float red[13];
#pragma unroll
    for (int e = 0; e < 12; e++) red[e] = wave_sum(w * P[e]);
    red[12] = wave_sum(r);
    if (lane == 63) {
#pragma unroll
      for (int e = 0; e < 13; e++) atomicAdd(&s_acc[k][e], red[e]);
    }
  }
  const float t0 = wave_sum(gh[0]), t1 = wave_sum(gh[1]), t2 = wave_sum(gh[2]);
  if (lane == 63) { atomicAdd(&s_gt[0], t0); atomicAdd(&s_gt[1], t1); atomicAdd(&s_gt[2], t2); }
  __syncthreads();
  for (int e = threadIdx.x; e < B * 13; e += 256) {
    const int k = e / 13, c = e % 13;
    const float v = s_acc[k][c];
    if (c < 12) atomicAdd(&a.dG[12 * (k + 1) + c], v);
    else atomicAdd(&a.drho[k + 1], v);
  }
  if (threadIdx.x < 3) atomicAdd(&a.dgt[threadIdx.x], s_gt[threadIdx.x]);
}

// ---- backward, K <= 0 (all bones): lane <-> (Gaussian slot, bone) ---------------------------------
// A wave64 is 8 Gaussian slots x 8 bone lanes; bones are covered in blocks of 8.  Each lane owns one
// bone per block, so the 13 per-bone sums over the Gaussians are plain register accumulations (no
// cross-lane reduction per Gaussian); only the two per-Gaussian normalisers (sum_k v_k and
// sum_k v_k dL/dw_k) are reduced over the 8 bone lanes (3 DPP steps).  The 8 slot rows are folded once
// per wave at the end.  (The pixel-major variant above spent ~30 DPP ops per (Gaussian, bone).)
#define LB_BONES 8                      // bone lanes per Gaussian slot
#define LB_MAXBLK ((MAX_J - 1 + LB_BONES - 1) / LB_BONES)
#ifndef LB_GPB
#define LB_GPB 1024                     // Gaussians per workgroup: every wave looks at a quarter of them, 64 at a time
#endif

__device__ __forceinline__ float row8_sum(float v) {
  // sum over the 8 lanes sharing (lane >> 3): xor 1, 2, 4 via DPP quad_perm / row_half_mirror patterns
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  return v;
}

template <int NBLK, bool MOD, int GPB>
__global__ __launch_bounds__(256) void lbs_backward_bonelane_kernel(LbsArgs a) {
  __shared__ Bone bones[MAX_J - 1 + LB_BONES];
  __shared__ float s_acc[MAX_J - 1 + LB_BONES][13];
  __shared__ float s_gt[3];
  const int B = a.J - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane >> 3, bl = lane & 7;
  const int wave_first = blockIdx.x * GPB + wave * (GPB / 4);
  const int wave_end = min(a.N, wave_first + GPB / 4);
  // Gaussians without an incoming gradient (behind saturated pixels, never at alpha >= 1/255, invisible: most of a
  // deep scene) contribute exact zeros to every sum: the wave looks at its Gaussians once, writes the zeros of
  // the per-Gaussian outputs, and walks only the others, eight at a time.
  // The wave's Gaussians are looked at in batches of four groups of 64 — their loads in flight together — and the ones with a
  // gradient are listed across the groups: walking group by group paid two dependent round trips (test, then the listed
  // Gaussians' operands) per group, which was most of this kernel's time in a sparse frame.  The FIRST batch is requested
  // before the bones are staged: its round trip runs under theirs.
  // (a large scene's workgroups take four or eight times the Gaussians — GPB = 4096 / 8192: the bones' staging, the fold of
  // the sums and the partials cost a workgroup ~15 us whatever it walks)
  constexpr int GROUPS = GPB / 4 / 64, GBATCH = 4;
  float4 hq[GBATCH];
  float gq[GBATCH][3];
  auto request = [&](const int gb) {
#pragma unroll
    for (int g4 = 0; g4 < GBATCH; g4++) {
      const int n = wave_first + 64 * (gb + g4) + lane;
      hq[g4] = make_float4(0.f, 0.f, 0.f, 0.f); gq[g4][0] = 0.f; gq[g4][1] = 0.f; gq[g4][2] = 0.f;
      if (n < wave_end) {
        hq[g4] = reinterpret_cast<const float4*>(a.g_rot)[n];
        gq[g4][0] = a.g_xyz[3 * n]; gq[g4][1] = a.g_xyz[3 * n + 1]; gq[g4][2] = a.g_xyz[3 * n + 2];
      }
    }
  };
  request(0);
  stage_bones(a, bones);
  for (int k = B + threadIdx.x; k < NBLK * LB_BONES; k += 256) {  // padding bones: never selected
    Bone z;
    memset(&z, 0, sizeof(z));
    z.len2c = 1.f;
    bones[k] = z;
  }
  for (int e = threadIdx.x; e < NBLK * LB_BONES * 13; e += 256) (&s_acc[0][0])[e] = 0.f;
  if (threadIdx.x < 3) s_gt[threadIdx.x] = 0.f;
  __syncthreads();
  float acc[NBLK][13];
#pragma unroll
  for (int bb = 0; bb < NBLK; bb++)
#pragma unroll
    for (int e = 0; e < 13; e++) acc[bb][e] = 0.f;
  float gt0 = 0.f, gt1 = 0.f, gt2 = 0.f;
  const float gx = a.global_trans[0], gy = a.global_trans[1], gz = a.global_trans[2];
  __shared__ unsigned short s_list[4][GPB / 4];
  int n_work = 0;
#pragma unroll 1
  for (int gb = 0; gb < GROUPS; gb += GBATCH) {
    if (gb > 0) request(gb);
#pragma unroll
    for (int g4 = 0; g4 < GBATCH; g4++) {
      const int n = wave_first + 64 * (gb + g4) + lane;
      const bool touched = (gq[g4][0] != 0.f) || (gq[g4][1] != 0.f) || (gq[g4][2] != 0.f) ||
                           (hq[g4].x != 0.f) || (hq[g4].y != 0.f) || (hq[g4].z != 0.f) || (hq[g4].w != 0.f);
      if (n < wave_end && !touched && a.dmask) a.dmask[n] = 0.f;
      const uint64_t tm = __builtin_amdgcn_ballot_w64(touched);
      if constexpr (MOD) {
        // the zeros of dL/dweight_mod for the group's untouched rows: the group's 64 rows are ONE contiguous run of 64 B floats —
        // 16-byte pieces, consecutive lanes = consecutive addresses (a row of B floats per lane was B stores of 64 scattered
        // dwords each: 46 partially written lines per instruction, +29 us on this kernel with the WeightMLP on); a piece that
        // overlaps a touched row (the walk below writes those) is written element by element
        const int ng = wave_first + 64 * (gb + g4);
        const int n_el = max(0, min(64, wave_end - ng)) * B;
        float* run = a.dmod + (size_t)ng * B;
        const float invB = 1.0f / (float)B;
        for (int p4 = lane * 4; p4 < n_el; p4 += 256) {
          const int r0 = (int)(((float)p4 + 0.5f) * invB), r1 = (int)(((float)min(p4 + 3, n_el - 1) + 0.5f) * invB);
          if (p4 + 3 < n_el && !((tm >> r0) & 1ull) && !((tm >> r1) & 1ull)) {
            *reinterpret_cast<float4*>(run + p4) = make_float4(0.f, 0.f, 0.f, 0.f);
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int e = p4 + q;
              if (e < n_el && !((tm >> (int)(((float)e + 0.5f) * invB)) & 1ull)) run[e] = 0.f;
            }
          }
        }
      }
      if (touched) s_list[wave][n_work + __builtin_popcountll(tm & ((1ull << lane) - 1ull))] = (unsigned short)(64 * (gb + g4) + lane);
      n_work += __builtin_popcountll(tm);
    }
  }
  {
  const int sub_first = wave_first;
  // operands of a step's eight Gaussians, requested one step ahead of their use
  struct StepOps { int n; bool valid; float px, py, pz, m, g0, g1, g2; float4 h; };
  auto fetch = [&](int w0) {
    StepOps o;
    o.valid = w0 + slot < n_work;
    o.n = o.valid ? sub_first + (int)s_list[wave][w0 + slot] : 0;  // (a wave's own LDS writes are ordered)
    o.px = 0.f; o.py = 0.f; o.pz = 0.f; o.m = 0.f; o.g0 = 0.f; o.g1 = 0.f; o.g2 = 0.f;
    o.h = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o.valid) {
      o.px = a.x[3 * o.n]; o.py = a.x[3 * o.n + 1]; o.pz = a.x[3 * o.n + 2];
      o.m = a.motion_mask ? a.motion_mask[o.n] : 1.0f;
      o.g0 = a.g_xyz[3 * o.n]; o.g1 = a.g_xyz[3 * o.n + 1]; o.g2 = a.g_xyz[3 * o.n + 2];
      o.h = reinterpret_cast<const float4*>(a.g_rot)[o.n];
    }
    return o;
  };
  StepOps nxt = fetch(0);
  for (int w0 = 0; w0 < n_work; w0 += 8) {
    const StepOps cu = nxt;
    if (w0 + 8 < n_work) nxt = fetch(w0 + 8);
    const bool valid = cu.valid;
    const int n = cu.n;
    const float px = cu.px, py = cu.py, pz = cu.pz, m = cu.m, g0 = cu.g0, g1 = cu.g1, g2 = cu.g2;
    const float4 h = cu.h;
    const float gh0 = g0 * m, gh1 = g1 * m, gh2 = g2 * m;
    const float hh0 = h.x * m, hh1 = h.y * m, hh2 = h.z * m, hh3 = h.w * m;
    // pass 1: this lane's bones
    float v[NBLK], u[NBLK], d2[NBLK], dw[NBLK], du[NBLK], md[NBLK];
    float sum = 0.f, sv = 0.f, su = 0.f;
#pragma unroll
    for (int bb = 0; bb < NBLK; bb++) {
      const int k = bb * LB_BONES + bl;
      const Bone& b = bones[k];
      d2[bb] = bone_d2_fast(b, px, py, pz);
      const bool on = valid && (k < B);
      u[bb] = on ? fast_exp(-d2[bb] * b.inv2r2) : 0.f;
      if constexpr (MOD) {
        md[bb] = on ? a.weight_mod[(size_t)n * B + k] : 1.0f;
        v[bb] = on ? u[bb] * md[bb] + 1e-7f : 0.f;
      } else {
        md[bb] = 1.0f;
        v[bb] = on ? u[bb] + 1e-7f : 0.f;
      }
      const float Ax = b.G[0] * px + b.G[1] * py + b.G[2] * pz + b.G[3];
      const float Ay = b.G[4] * px + b.G[5] * py + b.G[6] * pz + b.G[7];
      const float Az = b.G[8] * px + b.G[9] * py + b.G[10] * pz + b.G[11];
      dw[bb] = gh0 * Ax + gh1 * Ay + gh2 * Az + hh0 * b.q[0] + hh1 * b.q[1] + hh2 * b.q[2] + hh3 * b.q[3];
      du[bb] = g0 * Ax + g1 * Ay + g2 * Az + h.x * b.q[0] + h.y * b.q[1] + h.z * b.q[2] + h.w * b.q[3];  // un-masked (dmask)
      sum += v[bb];
      sv += v[bb] * dw[bb];
      su += v[bb] * du[bb];
    }
    sum = row8_sum(sum);
    sv = row8_sum(sv);
    const float inv = valid ? 1.0f / sum : 0.f;
    const float S = sv * inv;
    if (a.dmask) {
      su = row8_sum(su);
      if (valid && bl == 0) a.dmask[n] = su * inv + g0 * (gx - px) + g1 * (gy - py) + g2 * (gz - pz);
    }
    // pass 2: accumulate this lane's bones
    const float P[12] = {gh0 * px, gh0 * py, gh0 * pz, gh0, gh1 * px, gh1 * py, gh1 * pz, gh1,
                         gh2 * px, gh2 * py, gh2 * pz, gh2};
#pragma unroll
    for (int bb = 0; bb < NBLK; bb++) {
      const float w = v[bb] * inv;
      const float dLdv = (dw[bb] - S) * inv;
      float r;
      if constexpr (MOD) r = dLdv * md[bb] * u[bb] * d2[bb] * (2.0f * bones[bb * LB_BONES + bl].inv2r2);
      else r = dLdv * u[bb] * d2[bb] * (2.0f * bones[bb * LB_BONES + bl].inv2r2);
      if (MOD && valid && bb * LB_BONES + bl < B) a.dmod[(size_t)n * B + bb * LB_BONES + bl] = dLdv * u[bb];  // v = u * mod + 1e-7
#pragma unroll
      for (int e = 0; e < 12; e++) acc[bb][e] += w * P[e];
      acc[bb][12] += r;
    }
    if (bl == 0) { gt0 += gh0; gt1 += gh1; gt2 += gh2; }
  }
  }
  // fold the 8 slot rows (lanes l, l^8, l^16, l^32) and push to the workgroup accumulators
#pragma unroll
  for (int bb = 0; bb < NBLK; bb++)
#pragma unroll
    for (int e = 0; e < 13; e++) {
      float t = acc[bb][e];
      t += __shfl_xor(t, 8); t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
      if (slot == 0) atomicAdd(&s_acc[bb * LB_BONES + bl][e], t);
    }
  gt0 = wave_sum(gt0); gt1 = wave_sum(gt1); gt2 = wave_sum(gt2);
  if (lane == 63) { atomicAdd(&s_gt[0], gt0); atomicAdd(&s_gt[1], gt1); atomicAdd(&s_gt[2], gt2); }
  __syncthreads();
  // per-workgroup partial sums; lbs_backward_finish_kernel adds them up in a fixed order (deterministic)
  float* part = a.partial + (size_t)blockIdx.x * (B * 13 + 3);
  for (int e = threadIdx.x; e < B * 13; e += 256) part[e] = (&s_acc[0][0])[e];
  if (threadIdx.x < 3) part[B * 13 + threadIdx.x] = s_gt[threadIdx.x];
}

// one workgroup per output value: sum over the workgroups' partials
__global__ __launch_bounds__(256) void lbs_backward_finish_kernel(LbsArgs a, int n_parts) {
  __shared__ float s_red[4];
  const int B = a.J - 1, stride = B * 13 + 3, e = blockIdx.x;
  float t = 0.f;
  for (int p = threadIdx.x; p < n_parts; p += 256) t += a.partial[(size_t)p * stride + e];
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 63) s_red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    if (e < B * 13) {
      const int k = e / 13, c = e % 13;
      if (c < 12) a.dG[12 * (k + 1) + c] = v; else a.drho[k + 1] = v;
    } else a.dgt[e - B * 13] = v;
  }
  // joint 0 carries no bone: its transform / radius receive no gradient from the skinning
  if (e == 0 && threadIdx.x < 12) a.dG[threadIdx.x] = 0.f;
  if (e == 0 && threadIdx.x == 12) a.drho[0] = 0.f;
}

template <int NBLK, int GPB>
static void launch_lbs_bwd_bonelane_g(const LbsArgs& a, hipStream_t s) {
  const int blocks = (a.N + GPB - 1) / GPB;
  // (the weight-modulated variant — WeightMLP head on — is a separate instantiation: the LBS-only kernel keeps its registers)
  if (a.weight_mod) hipLaunchKernelGGL((lbs_backward_bonelane_kernel<NBLK, true, GPB>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((lbs_backward_bonelane_kernel<NBLK, false, GPB>), dim3(blocks), dim3(256), 0, s, a);
  hipLaunchKernelGGL(lbs_backward_finish_kernel, dim3((a.J - 1) * 13 + 3), dim3(256), 0, s, a, blocks);
}
template <int NBLK>
static void launch_lbs_bwd_bonelane(const LbsArgs& a, hipStream_t s) {
  // 1024 Gaussians per workgroup fill the chip at the bench size (293 workgroups).  At 64 joints the kernel holds 13 sums for
  // 8 blocks of bones per lane — 256 VGPRs, ONE workgroup per CU — and a workgroup costs ~15 us whatever it walks (staging
  // the bones, the fold of the sums, the partials): 2 M Gaussians were 1954 workgroups = 7.6 rounds = 154 us.  So a large
  // scene's workgroups take as many Gaussians as leave about one workgroup per CU.
  if (a.N >= 8 * 240 * 1024) launch_lbs_bwd_bonelane_g<NBLK, 8192>(a, s);
  else if (a.N >= 4 * 240 * 1024) launch_lbs_bwd_bonelane_g<NBLK, 4096>(a, s);
  else launch_lbs_bwd_bonelane_g<NBLK, LB_GPB>(a, s);
}

}  // namespace riggs

using namespace riggs;

extern "C" {

int riggs_fk_forward(int32_t J, const float* local_rot, const float* joints, const int32_t* parents,
                     const float* global_trans, float* transforms, float* node_rot, float* d_nodes,
                     riggs_stream stream) {
  RIGGS_REQUIRE(J >= 1 && J <= MAX_J, "num_joints must be in [1, 64]");
  {
    ProfScope ps(PROF_FK_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(fk_forward_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, J, local_rot, joints, parents,
                       global_trans, transforms, node_rot, d_nodes);
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_fk_backward(int32_t J, const float* local_rot, const float* joints, const int32_t* parents,
                      const float* dL_dtransforms, const float* dL_dd_nodes, float* dL_dlocal_rot,
                      float* dL_dglobal_trans, riggs_stream stream) {
  RIGGS_REQUIRE(J >= 1 && J <= MAX_J, "num_joints must be in [1, 64]");
  {
    ProfScope ps(PROF_FK_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(fk_backward_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, J, local_rot, joints, parents,
                       dL_dtransforms, dL_dd_nodes, dL_dlocal_rot, dL_dglobal_trans);
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

static int fill_lbs(LbsArgs& a, int32_t N, int32_t J, int32_t K, const float* x, const float* joints,
                    const int32_t* parents, const float* rho, const float* transforms, const float* node_rot,
                    const float* gt, const float* mask) {
  RIGGS_REQUIRE(J >= 2 && J <= MAX_J, "num_joints must be in [2, 64]");
  RIGGS_REQUIRE(N >= 0, "num_points < 0");
  RIGGS_REQUIRE(K < J, "K must be < num_joints");
  memset(&a, 0, sizeof(a));
  a.N = N; a.J = J; a.K = K; a.x = x; a.joints = joints; a.parents = parents; a.node_radius_log = rho;
  a.transforms = transforms; a.node_rot = node_rot; a.global_trans = gt; a.motion_mask = mask;
  return 0;
}

// two Gaussians per thread in the all-bones forward (see lbs_forward_kernel) where the launch still fills the chip
static bool lbs_two_per_thread(int N, int J) { return N >= LBS_PTS2_MIN_N && J >= LBS_PTS2_MIN_J; }

static unsigned lbs_grid(int N, int pts) { return (unsigned)((N + 256 * pts - 1) / (256 * pts)); }
// The bone records through the scalar cache (lbs_forward_scalar_kernel) where that wins: the large, many-joint scenes whose loop is
// bound by the LDS handing every wave the records (2 M x 64: 129 -> 104 us; at 300 k x 24 the extra one-workgroup launch in front
// costs what the loop gains: 13.0 against 15.9 us — tools/lbs_scalar_ab.py); the caller must have handed over a table
// (riggs_lbs_bone_table_bytes).  riggs_set_option("lbs_scalar", 1) forces it at every size (the A/B), -1 forbids it.
static bool lbs_use_scalar(const LbsArgs& a, const void* bone_table) {
  if (!bone_table || a.K > 0 || a.weight_mod) return false;
  const int o = option(OPT_LBS_SCALAR);
  return o > 0 || (o == 0 && lbs_two_per_thread(a.N, a.J));
}
// the forward's weight_mod tile (one Gaussian per thread, up to 47 bones: 48 KB; beyond, the rows are read from global memory)
static size_t lbs_mod_lds(LbsArgs& a, int pts) {
  a.mod_lds = (a.weight_mod && pts == 1 && a.J - 1 <= 47) ? 1 : 0;
  return a.mod_lds ? (size_t)256 * ((a.J - 1) | 1) * sizeof(float) : 0;
}

int riggs_lbs_forward(int32_t N, int32_t J, int32_t K, const float* x, const float* joints, const int32_t* parents,
                      const float* node_radius_log, const float* transforms, const float* node_rot,
                      const float* global_trans, const float* motion_mask, const float* weight_mod, float* d_xyz,
                      float* d_rotation, float* nn_weight, int64_t* nn_idx, void* bone_table, riggs_stream stream) {
  LbsArgs a;
  int rc = fill_lbs(a, N, J, K, x, joints, parents, node_radius_log, transforms, node_rot, global_trans, motion_mask);
  if (rc) return rc;
  a.d_xyz = d_xyz; a.d_rot = d_rotation; a.nn_weight = nn_weight; a.nn_idx = nn_idx;
  a.weight_mod = weight_mod;
  RIGGS_REQUIRE(weight_mod == nullptr || K <= 0, "weight_mod is supported with K = -1 (all bones) only");
  if (N == 0) return 0;
  if (lbs_use_scalar(a, bone_table) && !nn_weight && !nn_idx) {
    Bone* table = (Bone*)bone_table;
    hipLaunchKernelGGL(lbs_bone_table_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, table);
    {
      ProfScope ps(PROF_LBS_FWD, (hipStream_t)stream);
      if (lbs_two_per_thread(N, J)) hipLaunchKernelGGL((lbs_forward_scalar_kernel<2>), dim3(lbs_grid(N, 2)), dim3(256), 0, (hipStream_t)stream, a, table);
      else hipLaunchKernelGGL((lbs_forward_scalar_kernel<1>), dim3(lbs_grid(N, 1)), dim3(256), 0, (hipStream_t)stream, a, table);
    }
    RIGGS_HIP_CHECK(hipGetLastError());
    return 0;
  }
  {
    ProfScope ps(PROF_LBS_FWD, (hipStream_t)stream);
    if (a.K > 0) hipLaunchKernelGGL((lbs_forward_kernel<true, false, 1>), dim3(lbs_grid(N, 1)), dim3(256), 0, (hipStream_t)stream, a);
    else if (lbs_two_per_thread(N, J)) hipLaunchKernelGGL((lbs_forward_kernel<false, false, 2>), dim3(lbs_grid(N, 2)), dim3(256), 0, (hipStream_t)stream, a);
    else { const size_t lds = lbs_mod_lds(a, 1); hipLaunchKernelGGL((lbs_forward_kernel<false, false, 1>), dim3(lbs_grid(N, 1)), dim3(256), lds, (hipStream_t)stream, a); }
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_lbs_forward_fk(int32_t N, int32_t J, int32_t K, const float* x, const float* joints, const int32_t* parents,
                         const float* node_radius_log, const float* local_rot, const float* global_trans,
                         const float* motion_mask, const float* weight_mod, float* transforms, float* node_rot, float* d_nodes,
                         float* d_xyz, float* d_rotation, void* bone_table, riggs_stream stream) {
  LbsArgs a;
  int rc = fill_lbs(a, N, J, K, x, joints, parents, node_radius_log, transforms, node_rot, global_trans, motion_mask);
  if (rc) return rc;
  a.d_xyz = d_xyz; a.d_rot = d_rotation;
  a.weight_mod = weight_mod;
  a.local_rot = local_rot; a.fk_transforms = transforms; a.fk_node_rot = node_rot; a.fk_d_nodes = d_nodes;
  RIGGS_REQUIRE(weight_mod == nullptr || K <= 0, "weight_mod is supported with K = -1 (all bones) only");
  RIGGS_REQUIRE(local_rot && transforms && node_rot && d_nodes, "riggs_lbs_forward_fk needs the pose and the three chain outputs");
  if (N == 0) return riggs_fk_forward(J, local_rot, joints, parents, global_trans, transforms, node_rot, d_nodes, stream);
  if (lbs_use_scalar(a, bone_table)) {
    Bone* table = (Bone*)bone_table;
    hipLaunchKernelGGL(lbs_fk_table_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, table);
    {
      ProfScope ps(PROF_LBS_FWD, (hipStream_t)stream);
      if (lbs_two_per_thread(N, J)) hipLaunchKernelGGL((lbs_forward_scalar_kernel<2>), dim3(lbs_grid(N, 2)), dim3(256), 0, (hipStream_t)stream, a, table);
      else hipLaunchKernelGGL((lbs_forward_scalar_kernel<1>), dim3(lbs_grid(N, 1)), dim3(256), 0, (hipStream_t)stream, a, table);
    }
    RIGGS_HIP_CHECK(hipGetLastError());
    return 0;
  }
  {
    ProfScope ps(PROF_LBS_FWD, (hipStream_t)stream);
    if (a.K > 0) hipLaunchKernelGGL((lbs_forward_kernel<true, true, 1>), dim3(lbs_grid(N, 1)), dim3(256), 0, (hipStream_t)stream, a);
    else if (lbs_two_per_thread(N, J)) hipLaunchKernelGGL((lbs_forward_kernel<false, true, 2>), dim3(lbs_grid(N, 2)), dim3(256), 0, (hipStream_t)stream, a);
    else { const size_t lds = lbs_mod_lds(a, 1); hipLaunchKernelGGL((lbs_forward_kernel<false, true, 1>), dim3(lbs_grid(N, 1)), dim3(256), lds, (hipStream_t)stream, a); }
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

size_t riggs_lbs_bone_table_bytes(void) { return sizeof(Bone) * MAX_J; }

size_t riggs_lbs_backward_workspace_bytes(int32_t N, int32_t J) {
  const size_t blocks = (size_t)(N > 0 ? (N + LB_GPB - 1) / LB_GPB : 1);
  return align_up(blocks * ((size_t)(J - 1) * 13 + 3) * sizeof(float));
}

int riggs_lbs_backward(int32_t N, int32_t J, int32_t K, const float* x, const float* joints, const int32_t* parents,
                       const float* node_radius_log, const float* transforms, const float* node_rot,
                       const float* global_trans, const float* motion_mask, const float* weight_mod,
                       const float* g_xyz, const float* g_rot, float* dL_dtransforms, float* dL_dnode_radius_log,
                       float* dL_dglobal_trans, float* dL_dmotion_mask, float* dL_dweight_mod, void* workspace,
                       riggs_stream stream) {
  LbsArgs a;
  int rc = fill_lbs(a, N, J, K, x, joints, parents, node_radius_log, transforms, node_rot, global_trans, motion_mask);
  if (rc) return rc;
  a.g_xyz = g_xyz; a.g_rot = g_rot; a.dG = dL_dtransforms; a.drho = dL_dnode_radius_log; a.dgt = dL_dglobal_trans;
  a.dmask = dL_dmotion_mask;
  a.weight_mod = weight_mod; a.dmod = dL_dweight_mod;
  RIGGS_REQUIRE(weight_mod == nullptr || (K <= 0 && dL_dweight_mod != nullptr), "weight_mod needs K = -1 and dL_dweight_mod");
  a.partial = (float*)workspace;
  RIGGS_REQUIRE(workspace != nullptr, "riggs_lbs_backward needs its workspace");
  hipStream_t s = (hipStream_t)stream;
  if (N == 0 || K > 0) {  // the top-K (thread-per-Gaussian) path accumulates into zeroed outputs
    RIGGS_HIP_CHECK(hipMemsetAsync(dL_dtransforms, 0, (size_t)J * 48, s));
    RIGGS_HIP_CHECK(hipMemsetAsync(dL_dnode_radius_log, 0, (size_t)J * 4, s));
    RIGGS_HIP_CHECK(hipMemsetAsync(dL_dglobal_trans, 0, 12, s));
  }
  if (N == 0) return 0;
  {
    ProfScope ps(PROF_LBS_BWD, s);
    const int nblk = (J - 1 + LB_BONES - 1) / LB_BONES;
    if (K > 0) hipLaunchKernelGGL(lbs_backward_kernel, dim3((N + 255) / 256), dim3(256), 0, s, a);
    else switch (nblk) {
      case 1: launch_lbs_bwd_bonelane<1>(a, s); break;
      case 2: launch_lbs_bwd_bonelane<2>(a, s); break;
      case 3: launch_lbs_bwd_bonelane<3>(a, s); break;
      case 4: launch_lbs_bwd_bonelane<4>(a, s); break;
      case 5: launch_lbs_bwd_bonelane<5>(a, s); break;
      case 6: launch_lbs_bwd_bonelane<6>(a, s); break;
      case 7: launch_lbs_bwd_bonelane<7>(a, s); break;
      default: launch_lbs_bwd_bonelane<8>(a, s); break;
    }
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
