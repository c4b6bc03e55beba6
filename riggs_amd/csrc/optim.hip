// Gaussian optimizer step and densification statistics.
//
// The reference steps torch.optim.Adam(l, lr=0.0, eps=1e-15) over six parameter tensors with per-group
// learning rates (/root/reference/scene/gaussian_model.py:205-217, train_rig.py:527): ~10 elementwise passes
// per tensor.  Here ALL groups are updated by ONE launch that touches every byte once: read p, g, m, v
// (16 B/element), write p, m, v (12 B/element) — 28 B/element is the algorithmic traffic, and the kernel is
// a pure HBM stream (59 floats per Gaussian: 1.65 KB, 0.5 GB per step at 300k Gaussians).  16-byte accesses,
// grid-stride over the concatenated index space of the groups; the arithmetic follows torch's single-tensor
// Adam operation for operation (lerp_, mul_/addcmul_, sqrt/div/add_, addcdiv_) with FP contraction off.
#include "common.h"

namespace riggs {

#define ADAM_MAX_GROUPS 32  // 32 x 80 B of kernel arguments (the 35 tensors of the skeleton optimizer: two launches)

struct AdamArgs {
  int n_groups;
  float* p[ADAM_MAX_GROUPS];
  const float* g[ADAM_MAX_GROUPS];
  float* m[ADAM_MAX_GROUPS];
  float* v[ADAM_MAX_GROUPS];
  int64_t vec_start[ADAM_MAX_GROUPS + 1];  // prefix of ceil(numel / 4) over the groups
  int64_t numel[ADAM_MAX_GROUPS];
  float neg_step_size[ADAM_MAX_GROUPS];    // -lr / (1 - beta1^t)
  float bc2_sqrt[ADAM_MAX_GROUPS];         // sqrt(1 - beta2^t)
  float w1, beta2, w2, eps;                // 1 - beta1, beta2, 1 - beta2, eps
  // capturable mode (hipGraph replay): the step count and, optionally, the learning rate live in DEVICE memory, as
  // torch.optim.Adam(capturable=True) keeps them; the coefficients above are then derived inside the kernel
  const float* step_dev[ADAM_MAX_GROUPS];  // 0-dim float tensors holding the count AFTER this update, or NULL
  const float* lr_dev[ADAM_MAX_GROUPS];    // 0-dim float tensors, or NULL (= lr_host)
  double lr_host[ADAM_MAX_GROUPS];
  double beta1_d, beta2_d;
  GateArg gate;                            // the frame's "valid" words (include/riggs_hip.h: riggs_gate); n == 0: never gated
  uint32_t* nonfinite;                     // riggs_adam_step_guarded: elements whose gradient is NaN / Inf are left alone and counted here
  const float* coef[ADAM_MAX_GROUPS];      // capturable mode: (1 - beta1^t, sqrt(1 - beta2^t)) from adam_advance_kernel, or NULL (computed here)
};

// The step counts of a gated update: advanced by ONE small launch in front of it (every workgroup of the update reads them, so
// none of them may be the writer) — or, when the frame is invalid, left alone and `skipped` counted up instead.
#define ADAM_ADVANCE_MAX 128
struct AdvanceArgs {
  int n;
  float* step[ADAM_ADVANCE_MAX];
  GateArg gate;
  // the bias corrections of the new counts, evaluated HERE once (double precision, as torch does with Python floats) instead of
  // in the prologue of every one of the update's 16 384 workgroups — two double-precision pow() per workgroup were a tenth of a
  // bandwidth-bound kernel: coef[2 k] = 1 - beta1^t, coef[2 k + 1] = sqrt(1 - beta2^t)   (NULL: not wanted)
  float* coef;
  double beta1, beta2;
};
__global__ __launch_bounds__(ADAM_ADVANCE_MAX) void adam_advance_kernel(AdvanceArgs a, uint32_t* skipped) {
  const bool closed = gate_is_set(a.gate);
  if (closed) {
    if (threadIdx.x == 0 && skipped) atomicAdd(skipped, 1u);
    return;
  }
  if ((int)threadIdx.x < a.n) {
    const float t = a.step[threadIdx.x][0] + 1.0f;
    a.step[threadIdx.x][0] = t;
    if (a.coef) {
      a.coef[2 * threadIdx.x] = (float)(1.0 - pow(a.beta1, (double)t));
      a.coef[2 * threadIdx.x + 1] = (float)sqrt(1.0 - pow(a.beta2, (double)t));
    }
  }
}

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float w1, float beta2, float w2, float eps,
                                            float neg_step, float bc2s) {
#pragma clang fp contract(off)
  m = m + w1 * (g - m);
  v = v * beta2 + (w2 * g) * g;
  const float denom = sqrtf(v) / bc2s + eps;
  p = p + neg_step * (m / denom);
}

// riggs_adam_step_guarded: an element whose gradient is not finite keeps p, m, v (a frame poisoned by a lost PoseMLP hand-off must
// not reach the parameters of an EAGER trainer either — it has no gate: its step counts live on the host); returns 1 for it
__device__ __forceinline__ uint32_t adam_update_guarded(float& p, float g, float& m, float& v, float w1, float beta2, float w2, float eps,
                                                        float neg_step, float bc2s) {
  if (!__builtin_isfinite(g)) return 1u;
  adam_update(p, g, m, v, w1, beta2, w2, eps, neg_step, bc2s);
  return 0u;
}

template <bool GUARD>
__global__ __launch_bounds__(256) void adam_step_kernel(AdamArgs a) {
  __shared__ float s_ns[ADAM_MAX_GROUPS], s_bs[ADAM_MAX_GROUPS];
  // an invalid frame (NaN pose after a lost PoseMLP hand-off, truncated lists, an exchange that unpacked nothing) is a
  // SKIPPED step: p, m, v stay bit for bit what they were (every workgroup reads the same words: a uniform decision)
  if (a.gate.n > 0 && gate_is_set(a.gate)) return;
  if (threadIdx.x < ADAM_MAX_GROUPS) {
    const int k = threadIdx.x;
    float ns = a.neg_step_size[k], bs = a.bc2_sqrt[k];
    if (k < a.n_groups && a.step_dev[k]) {  // same double-precision bias corrections as the host path, from device state
      const double lr = a.lr_dev[k] ? (double)a.lr_dev[k][0] : a.lr_host[k];
      if (a.coef[k]) {  // (evaluated by adam_advance_kernel)
        ns = (float)(-(lr / (double)a.coef[k][0]));
        bs = a.coef[k][1];
      } else {
        const double t = (double)a.step_dev[k][0];
        ns = (float)(-(lr / (1.0 - pow(a.beta1_d, t))));
        bs = (float)sqrt(1.0 - pow(a.beta2_d, t));
      }
    }
    s_ns[k] = ns; s_bs[k] = bs;
  }
  __syncthreads();
  const int64_t total = a.vec_start[a.n_groups];
  uint32_t bad = 0u;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int gi = 0;
#pragma unroll
    for (int k = 1; k < ADAM_MAX_GROUPS; k++)
      if (k < a.n_groups && i >= a.vec_start[k]) gi = k;
    const int64_t e = (i - a.vec_start[gi]) * 4;
    const int64_t n = a.numel[gi];
    float* __restrict__ P = a.p[gi];
    const float* __restrict__ G = a.g[gi];
    float* __restrict__ M = a.m[gi];
    float* __restrict__ V = a.v[gi];
    const float ns = s_ns[gi], bs = s_bs[gi];
    if (e + 4 <= n) {
      float4 p = *reinterpret_cast<float4*>(P + e);
      const float4 g = *reinterpret_cast<const float4*>(G + e);
      float4 m = *reinterpret_cast<float4*>(M + e), v = *reinterpret_cast<float4*>(V + e);
      if constexpr (GUARD) {
        bad += adam_update_guarded(p.x, g.x, m.x, v.x, a.w1, a.beta2, a.w2, a.eps, ns, bs);
        bad += adam_update_guarded(p.y, g.y, m.y, v.y, a.w1, a.beta2, a.w2, a.eps, ns, bs);
        bad += adam_update_guarded(p.z, g.z, m.z, v.z, a.w1, a.beta2, a.w2, a.eps, ns, bs);
        bad += adam_update_guarded(p.w, g.w, m.w, v.w, a.w1, a.beta2, a.w2, a.eps, ns, bs);
      } else {
        adam_update(p.x, g.x, m.x, v.x, a.w1, a.beta2, a.w2, a.eps, ns, bs);
        adam_update(p.y, g.y, m.y, v.y, a.w1, a.beta2, a.w2, a.eps, ns, bs);
        adam_update(p.z, g.z, m.z, v.z, a.w1, a.beta2, a.w2, a.eps, ns, bs);
        adam_update(p.w, g.w, m.w, v.w, a.w1, a.beta2, a.w2, a.eps, ns, bs);
      }
      *reinterpret_cast<float4*>(P + e) = p;
      *reinterpret_cast<float4*>(M + e) = m;
      *reinterpret_cast<float4*>(V + e) = v;
    } else {
      for (int64_t j = e; j < n; j++) {
        float p = P[j], m = M[j], v = V[j];
        if constexpr (GUARD) bad += adam_update_guarded(p, G[j], m, v, a.w1, a.beta2, a.w2, a.eps, ns, bs);
        else adam_update(p, G[j], m, v, a.w1, a.beta2, a.w2, a.eps, ns, bs);
        P[j] = p; M[j] = m; V[j] = v;
      }
    }
  }
  if constexpr (GUARD) {
    if (bad) atomicAdd(a.nonfinite, bad);  // (the rare path: nothing is added on a healthy step)
  }
}

// add_densification_stats (scene/gaussian_model.py:516-518) + the max_radii2D update of train_rig.py:333-335
__global__ __launch_bounds__(256) void densify_stats_kernel(int N, const float* __restrict__ vgrad, const uint8_t* __restrict__ filt,
                                                            const int32_t* __restrict__ radii, float* __restrict__ accum,
                                                            float* __restrict__ denom, float* __restrict__ max_radii) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N || !filt[i]) return;
  const float gx = vgrad[3 * i], gy = vgrad[3 * i + 1];
  accum[i] += sqrtf(gx * gx + gy * gy);
  denom[i] += 1.0f;
  if (max_radii && radii) max_radii[i] = fmaxf(max_radii[i], (float)radii[i]);
}

}  // namespace riggs

using namespace riggs;

extern "C" {

static int adam_launch(int32_t n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* numel, const double* lr, const int64_t* step,
                       const float* const* step_dev, const float* const* lr_dev, double beta1, double beta2, double eps,
                       riggs_stream stream, const riggs_gate* gate = nullptr, uint32_t* nonfinite = nullptr,
                       const float* coef = nullptr) {
  RIGGS_REQUIRE(n_groups >= 0 && n_groups <= ADAM_MAX_GROUPS, "at most 32 parameter tensors per launch");
  AdamArgs a;
  memset(&a, 0, sizeof(a));
  RIGGS_REQUIRE(gate_arg(a.gate, gate) == 0, "riggs_gate: 0..4 non-NULL words");
  a.n_groups = n_groups;
  a.nonfinite = nonfinite;
  int64_t vs = 0;
  for (int k = 0; k < n_groups; k++) {
    RIGGS_REQUIRE(params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k], "NULL tensor in an Adam group");
    RIGGS_REQUIRE(numel[k] >= 0 && (step_dev || step[k] >= 1), "numel >= 0 and step >= 1 (the count AFTER this update) are required");
    RIGGS_REQUIRE((((uintptr_t)params[k] | (uintptr_t)grads[k] | (uintptr_t)exp_avg[k] | (uintptr_t)exp_avg_sq[k]) & 15) == 0,
                  "Adam tensors must be 16-byte aligned");
    a.p[k] = params[k]; a.g[k] = grads[k]; a.m[k] = exp_avg[k]; a.v[k] = exp_avg_sq[k];
    a.numel[k] = numel[k];
    a.vec_start[k] = vs;
    vs += (numel[k] + 3) / 4;
    // bias corrections in double, as torch does with Python floats (torch/optim/adam.py _single_tensor_adam)
    if (step_dev) {
      RIGGS_REQUIRE(step_dev[k] != nullptr, "capturable mode needs a device step tensor per group");
      a.step_dev[k] = step_dev[k];
      a.coef[k] = coef ? coef + 2 * k : nullptr;
      a.lr_dev[k] = lr_dev ? lr_dev[k] : nullptr;
      a.lr_host[k] = lr[k];
    } else {
      const double bc1 = 1.0 - pow(beta1, (double)step[k]), bc2 = 1.0 - pow(beta2, (double)step[k]);
      a.neg_step_size[k] = (float)(-(lr[k] / bc1));
      a.bc2_sqrt[k] = (float)sqrt(bc2);
    }
  }
  a.vec_start[n_groups] = vs;
  for (int k = n_groups + 1; k <= ADAM_MAX_GROUPS; k++) a.vec_start[k] = vs;
  a.w1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.w2 = (float)(1.0 - beta2); a.eps = (float)eps;
  a.beta1_d = beta1; a.beta2_d = beta2;
  if (vs == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope ps(PROF_ADAM, s);
    const int64_t want = (vs + 255) / 256;
    // grid-stride beyond 64 workgroups per CU (measured: 16 -> 64 = -7 %); with the counts on the device every workgroup starts with
    // a round trip for its coefficients, so half as many of them: 32 per CU (106.5 -> 96.1 us ungated, 97.8 -> 94.2 us gated)
    const int64_t cap_wgs = 256 * (int64_t)(step_dev ? 32 : 64);
    const unsigned blocks = (unsigned)(want < cap_wgs ? want : cap_wgs);
    if (nonfinite) hipLaunchKernelGGL(adam_step_kernel<true>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(adam_step_kernel<false>, dim3(blocks), dim3(256), 0, s, a);
  }
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

int riggs_adam_step(int32_t n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, const double* lr, const int64_t* step, double beta1,
                    double beta2, double eps, riggs_stream stream) {
  return adam_launch(n_groups, params, grads, exp_avg, exp_avg_sq, numel, lr, step, nullptr, nullptr, beta1, beta2, eps, stream);
}

int riggs_adam_step_guarded(int32_t n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                            float* const* exp_avg_sq, const int64_t* numel, const double* lr, const int64_t* step, double beta1,
                            double beta2, double eps, uint32_t* nonfinite_count, riggs_stream stream) {
  RIGGS_REQUIRE(nonfinite_count != nullptr, "riggs_adam_step_guarded: the counter is NULL");
  return adam_launch(n_groups, params, grads, exp_avg, exp_avg_sq, numel, lr, step, nullptr, nullptr, beta1, beta2, eps, stream, nullptr,
                     nonfinite_count);
}

int riggs_adam_step_capturable(int32_t n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, const double* lr,
                               const float* const* step_dev, const float* const* lr_dev, double beta1, double beta2,
                               double eps, riggs_stream stream) {
  RIGGS_REQUIRE(step_dev != nullptr, "step_dev is NULL");
  return adam_launch(n_groups, params, grads, exp_avg, exp_avg_sq, numel, lr, nullptr, step_dev, lr_dev, beta1, beta2, eps, stream);
}

static int adam_advance(int32_t n_steps, float* const* step_dev, const riggs_gate* gate, uint32_t* skipped, double beta1, double beta2,
                        float* coef, riggs_stream stream) {
  RIGGS_REQUIRE(n_steps >= 0 && (n_steps == 0 || step_dev != nullptr), "riggs_adam_steps_advance_gated: bad arguments");
  for (int at = 0; at < n_steps; at += ADAM_ADVANCE_MAX) {
    AdvanceArgs adv;
    memset(&adv, 0, sizeof(adv));
    adv.coef = coef ? coef + 2 * at : nullptr; adv.beta1 = beta1; adv.beta2 = beta2;
    RIGGS_REQUIRE(gate_arg(adv.gate, gate) == 0, "riggs_gate: 0..4 non-NULL words");
    adv.n = n_steps - at < ADAM_ADVANCE_MAX ? n_steps - at : ADAM_ADVANCE_MAX;
    for (int k = 0; k < adv.n; k++) {
      RIGGS_REQUIRE(step_dev[at + k] != nullptr, "NULL step tensor");
      adv.step[k] = step_dev[at + k];
    }
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(ADAM_ADVANCE_MAX), 0, (hipStream_t)stream, adv, at == 0 ? skipped : nullptr);
    RIGGS_HIP_CHECK(hipGetLastError());
  }
  return 0;
}
int riggs_adam_steps_advance_gated(int32_t n_steps, float* const* step_dev, const riggs_gate* gate, uint32_t* skipped,
                                   riggs_stream stream) {
  return adam_advance(n_steps, step_dev, gate, skipped, 0.0, 0.0, nullptr, stream);
}
int riggs_adam_steps_advance_coef(int32_t n_steps, float* const* step_dev, const riggs_gate* gate, uint32_t* skipped, double beta1,
                                  double beta2, float* coef, riggs_stream stream) {
  RIGGS_REQUIRE(coef != nullptr, "riggs_adam_steps_advance_coef: coef is NULL");
  return adam_advance(n_steps, step_dev, gate, skipped, beta1, beta2, coef, stream);
}
int riggs_adam_step_gated_coef(int32_t n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, const double* lr, float* const* step_dev,
                               const float* const* lr_dev, double beta1, double beta2, double eps, const riggs_gate* gate,
                               const float* coef, riggs_stream stream) {
  RIGGS_REQUIRE(step_dev != nullptr && coef != nullptr, "riggs_adam_step_gated_coef: step_dev / coef is NULL");
  RIGGS_REQUIRE(n_groups >= 0 && n_groups <= ADAM_MAX_GROUPS, "at most 32 parameter tensors per launch");
  return adam_launch(n_groups, params, grads, exp_avg, exp_avg_sq, numel, lr, nullptr, step_dev, lr_dev, beta1, beta2, eps, stream, gate,
                     nullptr, coef);
}

int riggs_adam_step_gated(int32_t n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                          float* const* exp_avg_sq, const int64_t* numel, const double* lr, float* const* step_dev,
                          const float* const* lr_dev, double beta1, double beta2, double eps, const riggs_gate* gate,
                          uint32_t* skipped, int32_t advance_steps, riggs_stream stream) {
  RIGGS_REQUIRE(step_dev != nullptr, "step_dev is NULL");
  RIGGS_REQUIRE(n_groups >= 0 && n_groups <= ADAM_MAX_GROUPS, "at most 32 parameter tensors per launch");
  if (advance_steps && n_groups > 0) {
    int rc = riggs_adam_steps_advance_gated(n_groups, step_dev, gate, skipped, stream);
    if (rc) return rc;
  }
  return adam_launch(n_groups, params, grads, exp_avg, exp_avg_sq, numel, lr, nullptr, step_dev, lr_dev, beta1, beta2, eps, stream, gate);
}

int riggs_densify_stats(int32_t N, const float* viewspace_grad, const uint8_t* update_filter, const int32_t* radii,
                        float* xyz_gradient_accum, float* denom, float* max_radii2D, riggs_stream stream) {
  RIGGS_REQUIRE(N >= 0, "num_points < 0");
  if (N == 0) return 0;
  RIGGS_REQUIRE(viewspace_grad && update_filter && xyz_gradient_accum && denom, "missing buffers");
  hipLaunchKernelGGL(densify_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, viewspace_grad,
                     update_filter, radii, xyz_gradient_accum, denom, max_radii2D);
  RIGGS_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
