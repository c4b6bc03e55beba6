// Microbenchmark: latency of a flag hand-off between two workgroups, on the same XCD and on different ones, for several
// cache-policy combinations of the store and the polling load.   hipcc --offload-arch=gfx950 -O3 pingpong.hip -o pp && ./pp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 2000
#define SPIN_MAX 2000000

template <int ST, int LD>
__device__ __forceinline__ void st(uint32_t* p, uint32_t v) {
  if constexpr (ST == 0) __hip_atomic_store((__attribute__((address_space(1))) uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if constexpr (ST == 1) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else if constexpr (ST == 2) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
template <int LD>
__device__ __forceinline__ uint32_t ld(const uint32_t* p) {
  uint32_t v;
  if constexpr (LD == 0) v = __hip_atomic_load((__attribute__((address_space(1))) uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if constexpr (LD == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (LD == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (LD == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// workgroups `a` and `b` play; everyone records its XCC id
template <int ST, int LD>
__global__ void pingpong(uint32_t* flags, int a, int b, uint32_t base, uint64_t* out, uint32_t* xcc) {
  const int w = blockIdx.x;
  if (threadIdx.x == 0) xcc[w] = __builtin_amdgcn_s_getreg(63508) & 0xF;  // HW_REG_XCC_ID
  if (threadIdx.x != 0 || (w != a && w != b)) return;
  uint32_t* mine = flags + (w == a ? 0 : 64);   // separate cache lines
  uint32_t* other = flags + (w == a ? 64 : 0);
  const uint64_t t0 = wall_clock64();
  uint32_t fail = 0;
  for (uint32_t i = 1; i <= ITERS && !fail; i++) {
    if (w == a) {
      st<ST, LD>(mine, base + i);
      uint32_t spins = 0;
      while (ld<LD>(other) != base + i) if (++spins > SPIN_MAX) { fail = 1; break; }
    } else {
      uint32_t spins = 0;
      while (ld<LD>(other) != base + i) if (++spins > SPIN_MAX) { fail = 1; break; }
      st<ST, LD>(mine, base + i);
    }
  }
  const uint64_t t1 = wall_clock64();
  if (w == a) { out[0] = t1 - t0; out[1] = fail; }
}

template <int ST, int LD>
void run(const char* name, uint32_t* flags, uint64_t* out, uint32_t* xcc, int a, int b, uint32_t& base) {
  hipMemset(out, 0, 16);
  hipLaunchKernelGGL((pingpong<ST, LD>), dim3(64), dim3(64), 0, 0, flags, a, b, base, out, xcc);
  hipDeviceSynchronize();
  base += ITERS + 7;
  uint64_t h[2];
  uint32_t hx[64];
  hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
  hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
  printf("%-44s wg %2d (xcc %u) <-> wg %2d (xcc %u): %s  round trip %.2f us\n", name, a, hx[a], b, hx[b], h[1] ? "TIMED OUT" : "ok",
         (double)h[0] / 100.0 / ITERS);
}

int main() {
  uint32_t *flags, *xcc;
  uint64_t* out;
  hipMalloc(&flags, 4096); hipMemset(flags, 0, 4096);
  hipMalloc(&xcc, 256); hipMalloc(&out, 64);
  uint32_t base = 100;
  for (int rep = 0; rep < 2; rep++) {
    const int pairs[3][2] = {{0, 8}, {0, 1}, {8, 40}};
    for (auto& p : pairs) {
      run<0, 0>("agent atomic store / agent atomic load", flags, out, xcc, p[0], p[1], base);
      run<1, 1>("plain store / load sc0", flags, out, xcc, p[0], p[1], base);
      run<1, 2>("plain store / load sc1", flags, out, xcc, p[0], p[1], base);
      run<1, 3>("plain store / load sc0 sc1", flags, out, xcc, p[0], p[1], base);
      run<2, 1>("store sc0 / load sc0", flags, out, xcc, p[0], p[1], base);
      run<3, 2>("store sc1 / load sc1", flags, out, xcc, p[0], p[1], base);
      run<1, 4>("plain store / buffer_inv sc1 + plain load", flags, out, xcc, p[0], p[1], base);
    }
  }
  return 0;
}
