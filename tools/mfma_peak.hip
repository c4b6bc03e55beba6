// What the matrix pipe delivers with NOTHING else in the loop (the yardstick of DESIGN.md §4d: the fused heads' K-loops run at
// 0.9 PFLOP/s): v_mfma_f32_32x32x16_f16 on register operands, eight independent accumulators per wave (the heads' 2 x 4 tiling),
// W waves per SIMD.  build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
  h8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f - threadIdx.x * 0.002f); }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int wg_per_cu, int cus, float* d) {
  const int iters = 4000, blocks = cus * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  spin<NACC><<<blocks, 256>>>(d, 100);
  hipEventRecord(e0);
  spin<NACC><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * iters * NACC * 32768.0;
  printf("%d accumulators, %d workgroups of 4 waves per CU (%d waves per SIMD): %.3f ms, %.0f TFLOP/s\n", NACC, wg_per_cu, wg_per_cu,
         ms, flop / ms / 1e9);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s, %d CUs, %d MHz\n", p.name, cus, p.clockRate / 1000);
  float* d; hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
  run<8>(1, cus, d); run<8>(2, cus, d); run<4>(2, cus, d); run<2>(4, cus, d); run<8>(1, cus, d);
  return 0;
}
