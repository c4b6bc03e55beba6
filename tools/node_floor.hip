// What a dependent stage costs on this chip, as a graph node and as an in-launch grid barrier (the
// depth sort's six launches as "one sweep").  The stages do next to nothing — every workgroup reads 8 KB that ANOTHER
// workgroup (another XCD) wrote in the stage before and writes 8 KB — so the times are the floors any fused form of
// count -> scan -> scatter has to beat:
//   (a) S stages as S kernel nodes of one hipGraph (what the sort is today),
//   (b) the same S stages in ONE launch with S - 1 grid barriers (arrive counter + bounded spin, data through
//       agent-scope stores / loads: the eight XCDs' L2s are not coherent for plain accesses inside a launch),
//   (c) an empty kernel node (nothing read or written), for the launch + end-of-kernel floor alone.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/node_floor tools/node_floor.hip ; run: /tmp/node_floor [workgroups] [stages]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define WORDS 2048  // per workgroup and stage
typedef __attribute__((address_space(1))) uint32_t gu32;

__global__ __launch_bounds__(256) void empty_kernel() {}

__global__ __launch_bounds__(256) void stage_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
  const int src = (blockIdx.x * 37 + 11) % gridDim.x;  // (another workgroup's output of the stage before)
  for (int i = threadIdx.x; i < WORDS; i += 256) out[blockIdx.x * WORDS + i] = in[src * WORDS + i] + 1u;
}

__global__ __launch_bounds__(256) void fused_kernel(uint32_t* a, uint32_t* b, uint32_t* counters, int stages, uint32_t* fail) {
  __shared__ int s_fail;
  uint32_t* in = a;
  uint32_t* out = b;
  if (threadIdx.x == 0) s_fail = 0;
  for (int s = 0; s < stages; s++) {
    const int src = (blockIdx.x * 37 + 11) % gridDim.x;
    for (int i = threadIdx.x; i < WORDS; i += 256) {
      const uint32_t v = __hip_atomic_load((gu32*)(in + src * WORDS + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((gu32*)(out + blockIdx.x * WORDS + i), v + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (s + 1 < stages) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add((gu32*)(counters + s), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        while (__hip_atomic_load((gu32*)(counters + s), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 22)) { s_fail = 1; break; }
        }
      }
      __syncthreads();
      if (s_fail) { if (threadIdx.x == 0) *fail = 1u; return; }
    }
    uint32_t* t = in; in = out; out = t;
  }
}

__global__ void clear_kernel(uint32_t* counters, int n) { if ((int)threadIdx.x < n) counters[threadIdx.x] = 0u; }

static float time_graph(hipGraphExec_t g, hipStream_t s, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 20; i++) CHECK(hipGraphLaunch(g, s));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; i++) CHECK(hipGraphLaunch(g, s));
  CHECK(hipEventRecord(e1, s));
  CHECK(hipStreamSynchronize(s));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const int wg = argc > 1 ? atoi(argv[1]) : 147;
  const int stages = argc > 2 ? atoi(argv[2]) : 6;
  const int chains = 8;  // (the graph holds `chains` repetitions of the sequence: the replay's own cost is shared)
  uint32_t *a, *b, *counters, *fail;
  CHECK(hipMalloc(&a, (size_t)wg * WORDS * 4)); CHECK(hipMalloc(&b, (size_t)wg * WORDS * 4));
  CHECK(hipMalloc(&counters, 64 * 4)); CHECK(hipMalloc(&fail, 4));
  CHECK(hipMemset(a, 0, (size_t)wg * WORDS * 4)); CHECK(hipMemset(fail, 0, 4));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  auto capture = [&](auto body) {
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int c = 0; c < chains; c++) body();
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    return ge;
  };
  hipGraphExec_t g_nodes = capture([&] {
    for (int st = 0; st < stages; st++) hipLaunchKernelGGL(stage_kernel, dim3(wg), dim3(256), 0, s, (st & 1) ? b : a, (st & 1) ? a : b);
  });
  hipGraphExec_t g_empty = capture([&] {
    for (int st = 0; st < stages; st++) hipLaunchKernelGGL(empty_kernel, dim3(wg), dim3(256), 0, s);
  });
  // (the fused form needs its arrive counters cleared: that launch is part of its price, as counters[] of the sort are
  // cleared by the first kernel of the sort — here a one-wave launch in front, so (b) = 2 nodes)
  hipGraphExec_t g_fused = capture([&] {
    hipLaunchKernelGGL(clear_kernel, dim3(1), dim3(64), 0, s, counters, stages);
    hipLaunchKernelGGL(fused_kernel, dim3(wg), dim3(256), 0, s, a, b, counters, stages, fail);
  });
  hipGraphExec_t g_clear = capture([&] { hipLaunchKernelGGL(clear_kernel, dim3(1), dim3(64), 0, s, counters, stages); });
  const int reps = 200;
  for (int round = 0; round < 3; round++) {
    const float t_nodes = time_graph(g_nodes, s, reps) / chains;
    const float t_empty = time_graph(g_empty, s, reps) / chains;
    const float t_fused = time_graph(g_fused, s, reps) / chains;
    const float t_clear = time_graph(g_clear, s, reps) / chains;
    uint32_t f = 0;
    CHECK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
    printf("workgroups %d stages %d | %d kernel nodes: %.2f us (%.2f per stage) | %d empty nodes: %.2f us (%.2f each) | one launch with %d grid "
           "barriers (+ its clear node %.2f us): %.2f us -> %.2f per barrier-separated stage%s\n",
           wg, stages, stages, t_nodes, t_nodes / stages, stages, t_empty, t_empty / stages, stages - 1, t_clear, t_fused,
           (t_fused - t_clear) / stages, f ? "  [A BARRIER TIMED OUT]" : "");
  }
  return 0;
}
