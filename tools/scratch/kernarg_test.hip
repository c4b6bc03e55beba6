#include <hip/hip_runtime.h>
#include <stdio.h>
struct Args { int n; float* out; const float* in; int pad[20]; };
typedef const Args __attribute__((address_space(4))) CA;
__device__ __attribute__((noinline)) float callee(int i) {
  const Args& a = *(const Args*)(const void*)__builtin_amdgcn_kernarg_segment_ptr();
  return a.in[i] * 2.f + a.n;
}
__global__ void k(Args a) {
  CA* p = (CA*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  int i = threadIdx.x;
  if (i < p->n) a.out[i] = callee(i) + p->in[i];
}
int main() {
  float *in, *out; hipMalloc(&in, 256); hipMalloc(&out, 256);
  float h[64]; for (int i = 0; i < 64; i++) h[i] = i;
  hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
  Args a{}; a.n = 64; a.out = out; a.in = in;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
  printf("err %d: %f %f (expect 64 and 3*63+64=253)\n", (int)e, h[0], h[63]);
}
