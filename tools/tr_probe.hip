// What gfx950's transposing LDS read returns (csrc/mlp_wgrad.hip relies on it): every lane gets ONE column's four consecutive rows of the
// [4 rows][16 columns] block its 16-lane group addresses (four lanes per row, 8 bytes each; the row stride is free).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  const unsigned short* p;
  if (mode == 0) p = lds + l * 4;                                   // lane-linear
  else {                                                             // rows of 256 elements: group g: lane p -> row (p>>2), cols 16*(g&1)+4*(p&3), +8 rows for g>=2
    const int g = l >> 4, q = l & 15;
    p = lds + ((g >> 1) * 8 + (q >> 2)) * 256 + 16 * (g & 1) + 4 * (q & 3);
  }
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
  for (int j = 0; j < 4; j++) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int mode = 0; mode < 2; mode++) {
    probe<<<1, 64>>>(d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; l++) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
