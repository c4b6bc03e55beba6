// SH colour evaluation (computeColorFromSH) as a BLOCK JOB: 256 Gaussians per workgroup, their
// coefficient rows staged through LDS, one Gaussian per thread.
//
// Two hosts run it.  preprocess_fwd_kernel evaluates the colours of its own 256 Gaussians (the general path: any SH layout,
// the grouped tile sort).  In the usual configuration the geometry kernel leaves the colours out — no 46 KB LDS image, no
// 180-byte row per Gaussian on its critical path — and the blocks run as EXTRA WORKGROUPS of the tile sort's scatter launch
// (csrc/binning.hip, bin_scatter_kernel: its own workgroups are bound by the instruction issue of their ordered walk and
// leave the memory system idle for 30 us; the colours are first read by the compositing, the launch after).
#pragma once
#include "gauss_math.h"
#include "raster_internal.h"

namespace riggs {

// ---- coalesced SH staging -----------------------------------------------------------------
// A workgroup's 256 Gaussians own one contiguous run of coefficients in HBM ((N,16,3) records of
// 192 B, or the reference's split parameters _features_dc (N,1,3) / _features_rest (N,15,3) of
// 12 B + 180 B, scene/gaussian_model.py:177-195).  Per-thread record reads would touch every
// 128-B line from 8+ different load instructions; instead the run is copied with full-line float4
// accesses into LDS (row stride padded to an odd number of dwords -> conflict-free b32 reads) and
// each thread then picks its own record.  The same path, reversed, writes dL/dsh.
__device__ __forceinline__ int sh_lds_stride(int per) { return per | 1; }

// (every thread of the workgroup: nthr of them, four floats each per step)
__device__ __forceinline__ void sh_stage_in(const float* __restrict__ src, int per, int count, float* lds, int nthr = 256) {
  const int total = count * per;
  const int stride = sh_lds_stride(per);
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  const int step = nthr * 4;
  const int q1024 = step / per, r1024 = step % per;
  int e = threadIdx.x * 4;
  int g = e / per, k = e % per;
  for (; e < total; e += step) {
    float v[4];
    if (aligned && e + 3 < total) {
      const float4 t = *reinterpret_cast<const float4*>(src + e);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = (e + q < total) ? src[e + q] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int kk = k + q, gg = g;
      if (kk >= per) { kk -= per; gg++; }
      if (e + q < total) lds[gg * stride + kk] = v[q];
    }
    k += r1024; g += q1024;
    if (k >= per) { k -= per; g++; }
  }
}

// The same copy with NO registers and no LDS-write pass: direct-to-LDS loads (global_load_lds_dwordx4: a wave instruction moves
// 64 x 16 bytes to M0-base + lane x 16).  The LDS image is then the run itself, unpadded — which is the conflict-free layout
// whenever the record length is odd (45 floats: the reference's _features_rest rows; stride 45 = 13 mod 32), so no swizzle is
// needed; even record lengths (48: one (N,16,3) tensor) keep the register path with its padded rows.  The tail of the last
// workgroup's run (< 1 KB) goes lane-masked, its last < 16 bytes as scalars.  __syncthreads() behind it carries the vmcnt(0).
// Every wave of the workgroup takes part (1 KB slices, round robin).
typedef __attribute__((address_space(1))) const void* sh_gptr;
typedef __attribute__((address_space(3))) void* sh_lptr;
__device__ __forceinline__ void sh_stage_dma(const float* __restrict__ src, int total /* floats */, float* lds) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bytes = total * 4, pitch = (int)blockDim.x * 16;
  const char* g = reinterpret_cast<const char*>(src);
  char* l = reinterpret_cast<char*>(lds);
  for (int c = wave * 1024; c < bytes; c += pitch) {
    if (c + lane * 16 + 16 <= bytes)
      __builtin_amdgcn_global_load_lds((sh_gptr)(g + c + lane * 16), (sh_lptr)(l + c), 16, 0, 0);
  }
  const int done = bytes & ~15;
  if ((int)threadIdx.x < (bytes - done) / 4) lds[done / 4 + threadIdx.x] = src[done / 4 + threadIdx.x];
}

// colour of one Gaussian from its staged row: basis of the normalised view direction, + 0.5, clamped at 0 (bit k of the
// return value: channel k was clamped — the backward passes no gradient through it)
__device__ __forceinline__ uint8_t sh_color(int deg, const float p[3], const float* __restrict__ campos, bool split,
                                            const float dc0[3], const float* mine, float rgbv[3]) {
  float dx = p[0] - campos[0], dy = p[1] - campos[1], dz = p[2] - campos[2];
  float len = sqrtf(dx * dx + dy * dy + dz * dz);
  dx = dx / len; dy = dy / len; dz = dz / len;
  float B[16];
  sh_basis(deg, dx, dy, dz, B);
  const int nb = (deg + 1) * (deg + 1);
  const int koff = split ? 3 : 0;  // split layout: coefficient 0 comes from _features_dc
  float r0 = 0.f, r1 = 0.f, r2 = 0.f;
  if (split) { r0 = B[0] * dc0[0]; r1 = B[0] * dc0[1]; r2 = B[0] * dc0[2]; }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    if (k < nb && 3 * k >= koff) {
      r0 += B[k] * mine[3 * k - koff]; r1 += B[k] * mine[3 * k + 1 - koff]; r2 += B[k] * mine[3 * k + 2 - koff];
    }
  }
  r0 += 0.5f; r1 += 0.5f; r2 += 0.5f;
  rgbv[0] = fmaxf(r0, 0.f); rgbv[1] = fmaxf(r1, 0.f); rgbv[2] = fmaxf(r2, 0.f);
  return (uint8_t)((r0 < 0.f ? 1 : 0) | (r1 < 0.f ? 2 : 0) | (r2 < 0.f ? 4 : 0));
}

// one block of gpb Gaussians (a multiple of 64, <= blockDim.x: the threads beyond only help with the staging);
// lds: gpb x sh_lds_stride(per) floats.  The Gaussian's position is the one preprocess_fwd projected: means3D (+ d_xyz).
struct __attribute__((packed, aligned(4))) ColorF3 { float x, y, z; };
__device__ __forceinline__ void color_block(const ColorJob& j, int block, int gpb, float* lds) {
  const int first = block * gpb, count = min(gpb, j.N - first);
  if (count <= 0) return;
  const int per = j.shs_rest ? (j.M - 1) * 3 : j.M * 3;
  const float* src = (j.shs_rest ? j.shs_rest : j.shs) + (size_t)first * per;
  const bool dma = (per & 1) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  const int i = first + (int)threadIdx.x;
  const bool mine = (int)threadIdx.x < count;
  if (dma) sh_stage_dma(src, count * per, lds);
  int rad = 0;
  float p[3] = {0.f, 0.f, 0.f}, dx[3] = {0.f, 0.f, 0.f}, dc0[3] = {0.f, 0.f, 0.f};
  if (mine) {
    rad = j.radii[i];
    p[0] = j.means3D[3 * i]; p[1] = j.means3D[3 * i + 1]; p[2] = j.means3D[3 * i + 2];
    if (j.d_xyz) { dx[0] = j.d_xyz[3 * i]; dx[1] = j.d_xyz[3 * i + 1]; dx[2] = j.d_xyz[3 * i + 2]; }
    if (j.shs_rest) { dc0[0] = j.shs[3 * i]; dc0[1] = j.shs[3 * i + 1]; dc0[2] = j.shs[3 * i + 2]; }
  }
  if (!dma && per > 0) sh_stage_in(src, per, count, lds, (int)blockDim.x);
  __syncthreads();
  if (!mine || rad <= 0) return;
  if (j.d_xyz) { p[0] = p[0] + dx[0]; p[1] = p[1] + dx[1]; p[2] = p[2] + dx[2]; }
  float rgbv[3];
  const uint8_t cl = sh_color(j.deg, p, j.campos, j.shs_rest != nullptr, dc0, lds + threadIdx.x * sh_lds_stride(per), rgbv);
  ColorF3 c; c.x = rgbv[0]; c.y = rgbv[1]; c.z = rgbv[2];
  *reinterpret_cast<ColorF3*>(&j.rgb[i]) = c;
  j.clamped[i] = cl;
}

}  // namespace riggs
