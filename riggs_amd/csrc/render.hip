// Per-tile alpha compositing, forward and backward.
//
// Forward: a 16x16 tile is split over eight workgroups of 8x4 pixels; a wave is 8 pixels x 8 instance lanes.
// Instances of the tile are staged 256 at a time through LDS as three float4 records (48 B / instance, gathered
// as whole 16-B words) after a cull against the workgroup's pixel block.
// Backward: chunk-parallel and instance-major (lane <-> instance, the wave walks the tile's pixels).
#include <stdlib.h>

#include <type_traits>

#include "raster_internal.h"

namespace riggs {

#define ALPHA_MIN (1.0f / 255.0f)
#define ALPHA_MAX 0.99f
#define T_EPS 0.0001f
#define LOG2E 1.4426950408889634f

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }

#define QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define QUAD_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (float)(v)), ctrl, 0xf, 0xf, true))
#define QUAD_U(v, ctrl) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(v), ctrl, 0xf, 0xf, true))

__device__ __forceinline__ float quad_sum(float v) {
  v += QUAD_F(v, QP(1, 0, 3, 2));
  v += QUAD_F(v, QP(2, 3, 0, 1));
  return v;
}

// ---- render forward: eight instance-lanes per pixel ---------------------------------------------------------------------
// Wave = 8 pixels x 8 instance lanes, workgroup = 4 waves = an 8 x 4 pixel block (eight per tile).  The per-pixel chain
// of a deep tile is what bounds this kernel (a tile whose pixels never saturate walks its whole list), and a wave alone
// on its SIMD is issue-bound (one VALU instruction per 4 cycles): one DPP scan step evaluates eight CONSECUTIVE
// instances of a pixel (exclusive product scan of the transmittances inside the eight lanes, the T < 1e-4 stop resolved
// with one ballot, colour accumulated per lane and folded only at checkpoints and at the end).  Most instances of a
// tile's list never reach a given 8 x 4 block (the lists are built from 3-sigma tile rectangles): while a round of 256
// is staged (4 chunks, one per wave), every instance is tested against the block with the axis-aligned extent of its
// alpha >= 1/255 ellipse (xyd.w / rgb.w, from the preprocess kernel; conservative) and only the survivors are kept,
// compacted per 64-instance chunk so that the checkpoints of the backward stay at multiples of 64 of the ORIGINAL list
// position.  Lane i < 4 of a pixel's eight holds the checkpoint of chunk i of the round.
#define ODPP_F(old, v, ctrl, bank) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(v)), ctrl, 0xf, bank, false))
#define ODPP_U(old, v, ctrl, bank) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, 0xf, bank, false))
#define DPP_ROW_SHL4 0x104
#define DPP_ROW_SHR4 0x114
#define DPP_HALF_MIRROR 0x141
__device__ __forceinline__ float oct_sum(float v) {
  v += QUAD_F(v, QP(1, 0, 3, 2));
  v += QUAD_F(v, QP(2, 3, 0, 1));
  v += ODPP_F(0.f, v, DPP_HALF_MIRROR, 0xf);
  return v;
}
// ---- segmented tiles ---------------------------------------------------------------------------------------------------
// The workgroup of a block's FIRST segment (the OWNER) walks the tile's list front to back exactly as an unsegmented tile is
// walked: sequential rule, exact stops, early termination, absolute checkpoints.  A list longer than RIGGS_SEG instances has
// further work items, one per later segment and block, dealt in level order behind ALL first segments (second segments, then
// third ones ...): HELPERS.  A helper composites its segment alone, from T = 1 (alpha compositing is associative:
//   C = sum_s Tin_s * S_s,   Tin_s = prod_{s' < s} P_s'),
// keeps every pixel's state at the start of each of its four rounds in LDS and, when done, publishes these and its end state in
// its SEGMENT SLOT (write-through stores) and ORs bit SUMMARY into the segment's hand-shake word.  The owner, one round before
// it reaches a segment boundary, ORs bit CLAIM into the next segment's word; a helper ORs bit STARTED into it when it begins:
// of two ORs on a word exactly one sees the other's bit, and whoever is first owns the segment's CHECKPOINTS (a helper that
// finds the claim leaves; an owner that finds STARTED walks the segment for its own state only and stores no checkpoints).
// Summary there -> the owner COMBINES it instead of walking 1024 instances (fw_owner_rest), and every further segment that is
// summarized as well.  A pixel whose running T times the segment's P falls below 1e-4 stops INSIDE the segment: the round
// where T_in * T_local crosses the threshold is known from the published round states, and the owner composites that round
// again from the true state, with the same code as the main loop — so the stop, n_contrib and final_T keep the sequential rule
// (test_T = T (1 - alpha) < 1e-4, not counted).  When every pixel of the block has stopped the owner posts dead_from: helpers
// that have not started leave at once, running ones at their next round.  Nobody ever waits for anybody.
// A deep tile whose pixels do not saturate is thus composited by as many workgroups as the chip has free; a tile whose pixels
// saturate early costs what it did before (its helpers start late — level order — find dead_from, and leave).
// Checkpoints of a helper's segment are segment-local: the backward multiplies them with the segment's prefix, which the owner
// stores in the segment's slot (the identity for the segments whose checkpoints are its own).  Which segments are combined depends on
// timing and T_in * (local product) rounds differently from the running product (1e-7 relative, the class of the 8-lane scans):
// cfg.deterministic turns the helpers off.
typedef __attribute__((address_space(1))) uint32_t seg_gu32;
typedef __attribute__((address_space(1))) float seg_gf32;
#define SEG_SUMMARY 1u
#define SEG_CLAIM 2u
#define SEG_STARTED 4u
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store((seg_gf32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load((seg_gf32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_agent_u(const uint32_t* p) { return __hip_atomic_load((seg_gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// value of lane j of the own group of eight
__device__ __forceinline__ float oct_get(float v, int j, int lane) { return __shfl(v, (lane & 56) | j); }
__device__ __forceinline__ float oct_max(float v) {
  v = fmaxf(v, QUAD_F(v, QP(1, 0, 3, 2)));
  v = fmaxf(v, QUAD_F(v, QP(2, 3, 0, 1)));
  return fmaxf(v, ODPP_F(-1.0f, v, DPP_HALF_MIRROR, 0xf));
}
__device__ __forceinline__ uint32_t oct_max_u(uint32_t v) {
  v = max(v, QUAD_U(v, QP(1, 0, 3, 2)));
  v = max(v, QUAD_U(v, QP(2, 3, 0, 1)));
  return max(v, ODPP_U(0u, v, DPP_HALF_MIRROR, 0xf));
}

#define FW_B 256                    // instances per round
#define FW_NR (RIGGS_SEG / FW_B)    // rounds per segment
// LDS of the forward (file scope: the main loop and the chain — a function of its own, see fw_seg_chain — share it)
__shared__ float4 fw_stage[3 * FW_B];     // the staged round: records of the survivors (and, in fw_owner_rest, the summaries being combined)
#define fw_xyd (fw_stage)
#define fw_con (fw_stage + FW_B)
#define fw_rgb (fw_stage + 2 * FW_B)
__shared__ unsigned short fw_pos[FW_B];  // position of the survivor inside its batch
__shared__ int fw_cnt[FW_B / 64];        // survivors per chunk
__shared__ uint32_t fw_wmax[4];
__shared__ float fw_rs[FW_NR - 1][8][32];  // LOCAL segments: (T, C0, C1, C2, D, A, last, -) of the block's pixels at the start of rounds 1 ..
__shared__ uint32_t fw_word;

// a pixel's compositing state while its lists are walked: T and the stop bookkeeping are the same in its eight lanes, the
// sums are per-lane partials
struct FwWalk {
  float T, C0, C1, C2, D, A, Tstop;
  uint32_t last;
  bool done;
};

// cull 64 instances (one per lane) against a box of pixels and compact the survivors into region `chunk` of the staging buffer
// (`pos`: the instance's position inside its round of 256, what n_contrib is counted from)
__device__ __forceinline__ int fw_stage_chunk(const int chunk, const int lane, const int pos, const bool in_range, const float4 xy,
                                              const float4 co, const float4 cc, const float bx0, const float bx1, const float by0, const float by1) {
  const bool keep = in_range && ((xy.x + xy.w >= bx0) && (xy.x - xy.w <= bx1) && (xy.y + cc.w >= by0) && (xy.y - cc.w <= by1));
  const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
  const int cnt = __builtin_popcountll(mask);
  const int slot = chunk * 64 + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
  if (keep) { fw_xyd[slot] = xy; fw_con[slot] = co; fw_rgb[slot] = cc; fw_pos[slot] = (unsigned short)pos; }
  if (lane >= cnt && lane < ((cnt + 7) & ~7)) {  // null records up to the next multiple of 8
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    fw_xyd[chunk * 64 + lane] = z; fw_con[chunk * 64 + lane] = z; fw_rgb[chunk * 64 + lane] = z; fw_pos[chunk * 64 + lane] = 0;
  }
  if (lane == 0) fw_cnt[chunk] = cnt;
  return cnt;
}
// the main loop's round: every wave stages one chunk of the 256, culled against the block's 8 x 4 pixels
__device__ __forceinline__ int fw_stage_round(const int tid, const bool in_range, const float4 xy, const float4 co, const float4 cc,
                                              const float bx0, const float by0) {
  return fw_stage_chunk(tid >> 6, tid & 63, tid, in_range, xy, co, cc, bx0, bx0 + 7.0f, by0, by0 + 3.0f);
}

// the survivors of chunk k of the staged round, eight per step, for this wave's eight pixels
template <bool TRACE>
__device__ __forceinline__ void fw_composite_chunk(FwWalk& w, const int k, const int base, const int lane, const float pfx, const float pfy,
                                                   uint32_t& st_iters, uint32_t& st_full) {
  const int i = lane & 7;
  const int nk = fw_cnt[k];
  for (int g = 64 * k; g < 64 * k + nk; g += 8) {
    // (no 'all pixels finished' test here: it costs eight instructions per step of every wave to save a few steps
    // once per wave; the chunk loop has it)
    const float4 xy = fw_xyd[g + i];
    const float dx = xy.x - pfx, dy = xy.y - pfy;
    const float4 co = fw_con[g + i];
    const float pw = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
    const float alpha = fminf(ALPHA_MAX, co.w * fast_exp(pw));
    const bool valid = (pw <= 0.0f) && (alpha >= ALPHA_MIN) && !w.done;
    if constexpr (TRACE) st_iters++;
    if (__builtin_amdgcn_ballot_w64(valid) == 0) continue;
    if constexpr (TRACE) st_full++;
    const float4 c = fw_rgb[g + i];
    const int pos1 = base + (int)fw_pos[g + i] + 1;
    // one step: eight consecutive instances (one per lane) of this lane's pixel
    const float om = valid ? 1.0f - alpha : 1.0f;
    // exclusive product scan over the eight lanes: inside each quad first ...
    float b1 = QUAD_F(om, QP(0, 0, 1, 2)); b1 = ((i & 3) >= 1) ? b1 : 1.0f;     // [1, o0, o1, o2]
    float s1 = QUAD_F(b1, QP(0, 0, 1, 2)); s1 = ((i & 3) >= 1) ? s1 : 1.0f;     // [1, 1, o0, o1]
    float s2 = QUAD_F(b1, QP(0, 0, 0, 1)); s2 = ((i & 3) >= 2) ? s2 : 1.0f;     // [1, 1, 1, o0]
    float E = b1 * s1 * s2;                                                     // [1, o0, o0 o1, o0 o1 o2] per quad
    // ... then the upper quad takes the lower quad's total (row_shr:4 written to banks 1 and 3 only)
    const float Pq = QUAD_F(E * om, QP(3, 3, 3, 3));                            // product of the own quad
    E *= ODPP_F(1.0f, Pq, DPP_ROW_SHR4, 0xA);
    const float Tj = w.T * E;
    const float test_T = Tj * om;
    const bool sc = valid && (test_T < T_EPS);
    const uint64_t bits = __builtin_amdgcn_ballot_w64(sc);
    const uint32_t ob = (uint32_t)(bits >> (lane & 56)) & 0xFFu;
    const bool first_stop_before = (ob & ((1u << i) - 1u)) != 0u;
    const bool use = valid && !sc && !first_stop_before;
    const float wt = use ? alpha * Tj : 0.f;
    w.C0 += c.x * wt; w.C1 += c.y * wt; w.C2 += c.z * wt;
    w.D += xy.z * wt; w.A += wt;
    w.last = use ? (uint32_t)pos1 : w.last;
    if (sc && !first_stop_before) w.Tstop = Tj;  // transmittance in front of the instance that ends the pixel
    // product of all eight: the upper quad's running total, handed down to the lower quad (row_shl:4, banks 0 and 2)
    const float X3 = QUAD_F(E * om, QP(3, 3, 3, 3));
    const float prod8 = ODPP_F(X3, X3, DPP_ROW_SHL4, 0x5);
    const bool nostop = (ob == 0u);
    w.T = (nostop && !w.done) ? w.T * prod8 : w.T;
    w.done = w.done || !nostop;
  }
}

// The kernel's argument block, read where it is needed: the compiler loads every kernel argument it sees into scalar registers
// at the kernel's entry and keeps them there — two registers per pointer, two dozen pointers that only the epilogue of a work
// item uses — and the main loop, which is short of them, then parks live scalars in vector-register lanes.  The epilogues
// read their pointers through an opaque copy of the kernarg pointer instead.
typedef const RenderArgs __attribute__((address_space(4))) FwLateArgs;
__device__ __forceinline__ FwLateArgs* fw_late_args() {
  FwLateArgs* p = (FwLateArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

struct FwItem {  // a work item of the forward: one 8 x 4 pixel block of one segment of one tile
  int tile, sub, seg, total, index;
  uint32_t list_start;  // range.x
};
struct FwPrefix {  // every pixel's state in front of a segment (the same in its eight lanes)
  float T, c0, c1, c2, D, A;
  uint32_t last;
  bool stop;
};
struct FwPixel { int pxi, pyi, pix; bool inside; };
__device__ __forceinline__ FwPixel fw_pixel(const int W, const int H, const int tile, const int sub, const int wave, const int pl) {
  const int gx = (W + RIGGS_TILE - 1) / RIGGS_TILE;
  const int prow = (sub >> 1) * 4 + wave, pcol = (sub & 1) * 8 + pl;  // pixel row / column inside the tile
  FwPixel p;
  p.pxi = (tile % gx) * RIGGS_TILE + pcol; p.pyi = (tile / gx) * RIGGS_TILE + prow;
  p.inside = p.pxi < W && p.pyi < H;
  p.pix = prow * 16 + pcol;  // pixel index inside the tile (checkpoint layout)
  return p;
}
__device__ __forceinline__ float* fw_state_ptr(float* seg_state, const FwItem& it, const int pix, int sg, int word) {
  const uint32_t sslot0 = it.list_start / RIGGS_SEG + (uint32_t)it.tile;  // segment slot of the tile's segment 0
  return seg_state + ((size_t)(sslot0 + sg) * RIGGS_SEG_WORDS + word) * 256 + pix;
}
__device__ __forceinline__ uint32_t* fw_flag_ptr(uint32_t* seg_flags, const FwItem& it, int sg) {
  const uint32_t sslot0 = it.list_start / RIGGS_SEG + (uint32_t)it.tile;
  return seg_flags + (size_t)(sslot0 + sg) * 8 + it.sub;
}
// the block's pixels between the main loop and fw_combine: (T — the final T of a stopped pixel —, C0, C1, C2, D, A, last, stopped)
__shared__ float fw_st[8][32];

// A HELPER's end: publish the states at the start of rounds 1 .. (from LDS: written by this wave) and the end state — lane i
// writes word i —, then the hand-shake.  (A function of its own, NOT inlined, like fw_combine below: their address arithmetic
// and state must not lengthen the live ranges of the main loop, which runs at exactly the register budget of six waves per
// SIMD — a spilled register there is reloaded through the vector memory queue, BEHIND the prefetched gathers of the next
// round (vmcnt retires in order), and the software pipeline of the rounds is gone: measured, 2.3 -> 7 us per round.
// Everything is passed by value, in registers: a reference to a kernel-side object would pin it in scratch memory.)
struct FwEnd { float T, k0, k1, k2, kd, ka; uint32_t lm; bool done; };
__device__ __attribute__((noinline)) void fw_publish(float* seg_state, uint32_t* seg_flags, const FwItem it, const int pix, const FwEnd end) {
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 7, bp = (tid >> 6) * 8 + (lane >> 3);
  for (int e = 0; e < FW_NR - 1; e++) st_agent(fw_state_ptr(seg_state, it, pix, it.seg, 8 * e + i), fw_rs[e][i][bp]);
  const float v = (i == 0) ? (end.done ? 0.f : end.T) : (i == 1) ? end.k0 : (i == 2) ? end.k1 : (i == 3) ? end.k2 : (i == 4) ? end.kd
                  : (i == 5) ? end.ka : __uint_as_float(end.lm);
  st_agent(fw_state_ptr(seg_state, it, pix, it.seg, 8 * (FW_NR - 1) + i), v);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // performed
  __syncthreads();
  if (tid == 0) (void)__hip_atomic_fetch_or((seg_gu32*)fw_flag_ptr(seg_flags, it, it.seg), SEG_SUMMARY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The OWNER's way from the first segment whose summary it found (see the comment above) to the end of the list — the rare,
// cold part of its walk, in a function of its own so that the main loop stays what it was.  The block's state travels through
// fw_st.  Here every WAVE goes its own way with its eight pixels (a pixel row of the block), without a barrier: it looks at the
// hand-shake words itself, keeps its instances in its own quarter of the staging buffer, culls against its own row.
//  * A RUN of consecutive summarized segments (up to FW_RUN) is combined at once: a scan per pixel over the segments' end states
//    finds the segment in which it stops (the first one where T_in * P falls below 1e-4) and leaves every segment's prefix for
//    the backward; then only the rounds in which a pixel of the wave stops are composited again, chunk by chunk with the code
//    of the main loop, each from the true state at its start (prefix x the helper's round state).
//  * A segment that is not summarized is claimed and walked the same way, chunk by chunk — with its checkpoints unless a helper
//    has started on it (then they are the helper's, and the prefix is the true one).
// Whether a wave sees a summary that another wave of the block just missed does not matter: prefixes and checkpoints are per
// pixel, and the claim / started words say who owns a segment's checkpoints for the whole block.
#define FW_RUN 4
__shared__ unsigned long long fw_wmask[4];  // per wave: rounds of the run in which one of its pixels stops
__shared__ int fw_wend[4];                  // per wave: the segment its walk ended in front of, or -1 while pixels are alive
template <bool TRACE>
__device__ __attribute__((noinline)) int fw_owner_rest(const RenderArgs* kernargs, const FwItem it, int s) {
  const RenderArgs& a = *kernargs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pl = lane >> 3, i = lane & 7, bp = wave * 8 + pl;
  const FwPixel px = fw_pixel(a.W, a.H, it.tile, it.sub, wave, pl);
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const float pfx = (float)px.pxi, pfy = (float)px.pyi;
  const float bx0 = (float)((it.tile % gx) * RIGGS_TILE + (it.sub & 1) * 8);
  const float wy = (float)((it.tile / gx) * RIGGS_TILE + (it.sub >> 1) * 4 + wave);  // the wave's pixel row
  const int total = it.total, nseg = (total + RIGGS_SEG - 1) / RIGGS_SEG;
  const uint32_t slot0 = a.slot_base[it.tile];
  const unsigned long long t_chain = (TRACE && a.trace) ? wall_clock64() : 0ull;
  uint32_t st_steps = 0, st_walk = 0, st_a = 0, st_b = 0, st_runs = 0, st_own = 0;
  FwPrefix p;
  p.T = fw_st[0][bp]; p.c0 = fw_st[1][bp]; p.c1 = fw_st[2][bp]; p.c2 = fw_st[3][bp]; p.D = fw_st[4][bp]; p.A = fw_st[5][bp];
  p.last = __float_as_uint(fw_st[6][bp]); p.stop = fw_st[7][bp] != 0.f;
  auto seed = [&](FwWalk& w, const FwPrefix& q) {
    w.T = q.T; w.Tstop = -1.0f; w.last = q.last; w.done = q.stop;
    w.C0 = (i == 0) ? q.c0 : 0.f; w.C1 = (i == 0) ? q.c1 : 0.f; w.C2 = (i == 0) ? q.c2 : 0.f; w.D = (i == 0) ? q.D : 0.f; w.A = (i == 0) ? q.A : 0.f;
  };
  auto harvest = [&](const FwWalk& w, FwPrefix& q) {  // (every lane of the eight takes part)
    const float f0 = oct_sum(w.C0), f1 = oct_sum(w.C1), f2 = oct_sum(w.C2), fd = oct_sum(w.D), fa = oct_sum(w.A);
    const float t2 = oct_max(w.Tstop);
    const uint32_t l2 = oct_max_u(w.last);
    q.c0 = f0; q.c1 = f1; q.c2 = f2; q.D = fd; q.A = fa; q.last = l2;
    q.stop = w.done;
    q.T = (t2 >= 0.f) ? t2 : w.T;  // (T of a stopped pixel: the transmittance in front of the instance that ended it)
  };
  auto put_prefix = [&](const int sg, const FwPrefix& q) {
    if (i < 5) *fw_state_ptr(a.seg_state, it, px.pix, sg, RIGGS_SEG_PREFIX + i) = (i == 0) ? q.T : (i == 1) ? q.c0 : (i == 2) ? q.c1 : (i == 3) ? q.c2 : q.D;
  };
  // the wave walks instances [from, to) of the list (from a multiple of 64) with its eight pixels, 64 at a time: list entry two
  // chunks ahead, records one ahead; `base` of a chunk's round = what n_contrib counts from; ck: store the (absolute) checkpoints
  auto walk = [&](FwWalk& w, const int from, const int to, const bool ck) {
    auto load_id = [&](const int cb) { return (cb + lane < to) ? a.point_list[it.list_start + cb + lane] : 0xFFFFFFFFu; };
    uint32_t id1 = load_id(from), id2 = load_id(from + 64);
    float4 xy = make_float4(0.f, 0.f, 0.f, 0.f), co = xy, cc = xy;
    if (id1 != 0xFFFFFFFFu) { xy = a.xyd[id1]; co = a.conic_o[id1]; cc = a.rgb[id1]; }
    for (int cb = from; cb < to; cb += 64) {
      if (__builtin_amdgcn_ballot_w64(!w.done) == 0) break;
      if constexpr (TRACE) st_walk++;
      fw_stage_chunk(wave, lane, (cb & (FW_B - 1)) + lane, id1 != 0xFFFFFFFFu, xy, co, cc, bx0, bx0 + 7.0f, wy, wy);
      id1 = id2;
      id2 = load_id(cb + 128);
      xy = make_float4(0.f, 0.f, 0.f, 0.f); co = xy; cc = xy;
      if (id1 != 0xFFFFFFFFu) { xy = a.xyd[id1]; co = a.conic_o[id1]; cc = a.rgb[id1]; }
      if (ck) {
        const float k0 = oct_sum(w.C0), k1 = oct_sum(w.C1), k2 = oct_sum(w.C2), kd = oct_sum(w.D);
        if (!w.done && i < 5)
          a.ckpt[((size_t)(slot0 + (cb >> 6)) * 5 + i) * 256 + px.pix] = (i == 0) ? w.T : (i == 1) ? k0 : (i == 2) ? k1 : (i == 3) ? k2 : kd;
      }
      fw_composite_chunk<false>(w, wave, cb & ~(FW_B - 1), lane, pfx, pfy, st_a, st_b);
    }
  };
  for (;;) {
    if (__builtin_amdgcn_ballot_w64(!p.stop) == 0 || s >= nseg) break;
    // the run of summarized segments that starts at s (lane j looks at segment s + j); none: claim segment s
    int k;
    bool started = false;
    {
      const bool in = lane < FW_RUN && s + lane < nseg;
      uint32_t f = in ? ld_agent_u(fw_flag_ptr(a.seg_flags, it, s + lane)) : 0u;
      const unsigned long long ready = __builtin_amdgcn_ballot_w64(in && (f & SEG_SUMMARY));
      k = __builtin_ctzll(~ready);
      if (k == 0) {
        if (lane == 0) f = __hip_atomic_fetch_or((seg_gu32*)fw_flag_ptr(a.seg_flags, it, s), SEG_CLAIM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
        if (f & SEG_SUMMARY) k = 1;
        started = (f & SEG_STARTED) != 0u;
      }
    }
    if constexpr (TRACE) { st_runs++; if (k == 0) st_own++; }
    if (k == 0) {
      // walk segment s
      const int s_lo = s * RIGGS_SEG, s_hi = min(total, s_lo + RIGGS_SEG);
      if (started) put_prefix(s, p);
      else if (i < 5) *fw_state_ptr(a.seg_state, it, px.pix, s, RIGGS_SEG_PREFIX + i) = (i == 0) ? 1.0f : 0.f;
      FwWalk w;
      seed(w, p);
      walk(w, s_lo, s_hi, !started);
      FwPrefix q;
      harvest(w, q);
      if (!p.stop) p = q;
      s++;
      continue;
    }
    // ---- combine segments s .. s + k - 1: lane i holds word i of their end states
    if constexpr (TRACE) st_steps += (uint32_t)k;
    float ev[FW_RUN];
#pragma unroll
    for (int j = 0; j < FW_RUN; j++) ev[j] = (j < k) ? ld_agent(fw_state_ptr(a.seg_state, it, px.pix, s + j, 8 * (FW_NR - 1) + i)) : 0.f;
    int jstar = -1;  // the segment of the run in which the pixel stops
#pragma unroll
    for (int j = 0; j < FW_RUN; j++) {
      const float eP = oct_get(ev[j], 0, lane), e0 = oct_get(ev[j], 1, lane), e1 = oct_get(ev[j], 2, lane), e2 = oct_get(ev[j], 3, lane),
                  eD = oct_get(ev[j], 4, lane), eA = oct_get(ev[j], 5, lane);
      const uint32_t eL = __float_as_uint(oct_get(ev[j], 6, lane));
      if (j < k && !p.stop && jstar < 0) {
        put_prefix(s + j, p);
        const float tp = p.T * eP;
        if (tp >= T_EPS) {
          p.c0 += p.T * e0; p.c1 += p.T * e1; p.c2 += p.T * e2; p.D += p.T * eD; p.A += p.T * eA;
          p.last = eL ? eL : p.last;
          p.T = tp;
        } else jstar = j;  // (p stays the prefix of segment s + jstar)
      }
    }
    // the round of that segment: the first one at whose END T_in * T_local is below the threshold; its start state
    int Rstar = 64;  // linear round index in the run (64: none)
    FwPrefix q = p;
    if (lane == 0) fw_wmask[wave] = 0ull;
    if (__builtin_amdgcn_ballot_w64(jstar >= 0) != 0ull) {
      float e3[FW_NR - 1];
#pragma unroll
      for (int e = 0; e < FW_NR - 1; e++) e3[e] = (jstar >= 0) ? ld_agent(fw_state_ptr(a.seg_state, it, px.pix, s + jstar, 8 * e + i)) : 0.f;
      int rstar = FW_NR - 1;
#pragma unroll
      for (int r = FW_NR - 2; r >= 0; r--) if (p.T * oct_get(e3[r], 0, lane) < T_EPS) rstar = r;
#pragma unroll
      for (int e = 0; e < FW_NR - 1; e++) {
        const float eT = oct_get(e3[e], 0, lane), e0 = oct_get(e3[e], 1, lane), e1 = oct_get(e3[e], 2, lane), e2 = oct_get(e3[e], 3, lane),
                    eD = oct_get(e3[e], 4, lane), eA = oct_get(e3[e], 5, lane);
        const uint32_t eL = __float_as_uint(oct_get(e3[e], 6, lane));
        if (jstar >= 0 && rstar == e + 1) {
          q.T = p.T * eT; q.c0 = p.c0 + p.T * e0; q.c1 = p.c1 + p.T * e1; q.c2 = p.c2 + p.T * e2; q.D = p.D + p.T * eD; q.A = p.A + p.T * eA;
          q.last = eL ? eL : p.last;
        }
      }
      if (jstar >= 0) {
        Rstar = jstar * FW_NR + rstar;
        if (i == 0) atomicOr(&fw_wmask[wave], 1ull << Rstar);
      }
      // composite again, from the true state: every round in which a pixel of the wave stops (and, should the rounding of T_in * P
      // have promised a stop that the instance-by-instance product does not find, the rounds behind it — until the run ends)
      unsigned long long todo = fw_wmask[wave];
      const int run_lo = s * RIGGS_SEG, run_hi = min(total, (s + k) * RIGGS_SEG);
      FwWalk w;
      w.T = 1.0f; w.C0 = 0.f; w.C1 = 0.f; w.C2 = 0.f; w.D = 0.f; w.A = 0.f; w.Tstop = -1.0f; w.last = 0u; w.done = true;
      bool walking = false;
      int R = __builtin_ctzll(todo);
      for (;;) {
        const int base = run_lo + R * FW_B;
        if (Rstar == R) { walking = true; seed(w, q); w.done = false; }
        if ((R % FW_NR) == 0) {  // (uniform: every lane takes part in the shuffles of harvest)
          FwPrefix t;
          harvest(w, t);
          if (walking && !w.done && R > Rstar) put_prefix(s + R / FW_NR, t);  // a walk that crosses into the next segment: its true prefix
        }
        todo &= ~(1ull << R);
        walk(w, base, min(base + FW_B, run_hi), false);
        const bool goes_on = __builtin_amdgcn_ballot_w64(walking && !w.done) != 0ull && base + FW_B < run_hi;
        const int Rn = goes_on ? R + 1 : (todo ? __builtin_ctzll(todo) : 64);
        if (Rn >= 64 || run_lo + Rn * FW_B >= run_hi) break;
        R = Rn;
      }
      FwPrefix t;
      harvest(w, t);
      if (walking) p = t;
    }
    s += k;
  }
  // the block's end: every pixel final -> later segments are nobody's business any more
  if (lane == 0) fw_wend[wave] = (__builtin_amdgcn_ballot_w64(!p.stop) == 0ull) ? s : -1;
  if (i == 0) {
    fw_st[0][bp] = p.T; fw_st[1][bp] = p.c0; fw_st[2][bp] = p.c1; fw_st[3][bp] = p.c2; fw_st[4][bp] = p.D; fw_st[5][bp] = p.A;
    fw_st[6][bp] = __uint_as_float(p.last); fw_st[7][bp] = p.stop ? 1.0f : 0.f;
  }
  __syncthreads();
  const int e0 = fw_wend[0], e1 = fw_wend[1], e2 = fw_wend[2], e3 = fw_wend[3];
  const int smax = max(max(e0, e1), max(e2, e3));
  if (tid == 0 && e0 >= 0 && e1 >= 0 && e2 >= 0 && e3 >= 0 && smax < nseg)
    __hip_atomic_store((seg_gu32*)(a.dead_from + (size_t)it.tile * 8 + it.sub), (uint32_t)smax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (TRACE && a.trace && lane == 0 && (uint64_t)it.index < a.trace_items)
    a.trace[((size_t)it.index * 4 + wave) * 8 + 7] = (wall_clock64() - t_chain) | ((unsigned long long)st_steps << 32) | ((unsigned long long)st_walk << 44) |
                                                       ((unsigned long long)st_runs << 54) | ((unsigned long long)st_own << 59);
  return max(smax, s);  // (segments the block's pixels went through)
}

template <bool TRACE>
__global__ __launch_bounds__(256, 6) void render_fwd_oct_kernel(RenderArgs a) {
  constexpr int B = FW_B;
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pl = lane >> 3, i = lane & 7;
  {
    // tiles without instances: background only, one pixel per thread (a.items == NULL: nothing was binned — no
    // Gaussians or an empty arena — and every tile is such a tile)
    const int n_empty = a.items ? (int)a.item_ctr[2] : gx * ((a.H + RIGGS_TILE - 1) / RIGGS_TILE);
    if ((int)blockIdx.x < n_empty) {  // (the launch has more workgroups than tiles)
      FwLateArgs& l = *fw_late_args();
      const size_t HW = (size_t)l.H * l.W;
      const int t = l.items ? (int)l.empties[blockIdx.x] : (int)blockIdx.x;
      const int px = (t % gx) * RIGGS_TILE + (tid & 15), py = (t / gx) * RIGGS_TILE + (tid >> 4);
      if (px < l.W && py < l.H) {
        const size_t pid = (size_t)py * l.W + px;
        l.final_T[pid] = 1.0f; l.n_contrib[pid] = 0u; l.final_acc[pid] = make_float4(0.f, 0.f, 0.f, 0.f);
        l.out_color[pid] = l.bg[0]; l.out_color[HW + pid] = l.bg[1]; l.out_color[2 * HW + pid] = l.bg[2];
        l.out_depth[pid] = 0.f; l.out_alpha[pid] = 0.f;
      }
    }
  }
  // work list: (tile, segment) entries in the order the extra workgroup of bin_scatter_kernel wrote them (first segments of
  // the longest lists first, deeper segments behind all first ones), eight 8x4 pixel blocks each
  // (one item per workgroup — the launch covers the capacity of the list: an outer loop over items makes every item-invariant
  // scalar a loop invariant that the compiler computes up front and keeps, in vector-register lanes once the scalar file is full)
  const int n_items = a.items ? (int)a.item_ctr[0] * 8 : 0;
  FwItem item;
  item.index = (int)blockIdx.x;
  {
    // the eight blocks of an entry get workgroup ids 8 apart = the same XCD / L2 (workgroup b runs on XCD b % 8)
    const int it = item.index, full = (n_items >> 6) << 6;
    if (it >= n_items) return;
    int p;
    if (it < full) { p = ((it >> 6) << 3) + (it & 7); item.sub = (it >> 3) & 7; }
    else { p = (full >> 3) + ((it - full) >> 3); item.sub = (it - full) & 7; }
    const uint32_t e = a.items[p];
    item.tile = (int)(e & 0xFFFFu); item.seg = (int)(e >> 16);
  }
  const int tile = item.tile, sub = item.sub, seg = item.seg;
  const uint2 range = a.ranges[tile];
  const int total = (int)(range.y - range.x);
  item.total = total; item.list_start = range.x;
  const bool helper = seg > 0;                            // (implies a segmented tile)
  const bool owner_multi = !helper && total > RIGGS_SEG;  // the owner of a segmented tile
  if (helper) {
    // leave if every pixel of the block stopped in front of this segment, if the owner has claimed it — or the segment in front
    // of it: it is walking that one now and will be here before this workgroup is done —, or in the reproducible mode
    if (tid == 0) {
      const uint32_t d = ld_agent_u(a.dead_from + (size_t)tile * 8 + sub);
      const uint32_t fb = seg > 1 ? ld_agent_u(fw_flag_ptr(a.seg_flags, item, seg - 1)) : 0u;  // (segment 0 is always the owner's: no word)
      uint32_t skip = ((d != 0u && (uint32_t)seg >= d) || (fb & SEG_CLAIM) || a.deterministic) ? 1u : 0u;
      if (!skip) {  // the segment's checkpoints are this workgroup's unless the owner was first
        const uint32_t old = __hip_atomic_fetch_or((seg_gu32*)fw_flag_ptr(a.seg_flags, item, seg), SEG_STARTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        skip = (old & SEG_CLAIM) ? 1u : 0u;
      }
      fw_word = skip;
    }
    __syncthreads();
    const uint32_t w0 = fw_word;
    __syncthreads();
    if (w0) return;
    if (tid < 32) for (int e = 0; e < FW_NR - 1; e++) fw_rs[e][0][tid] = 0.f;  // "not reached" until a round says otherwise
  }
  const int lo = helper ? seg * RIGGS_SEG : 0, hi = helper ? min(total, lo + RIGGS_SEG) : total;
  const FwPixel px = fw_pixel(a.W, a.H, tile, sub, wave, pl);
  const float pfx = (float)px.pxi, pfy = (float)px.pyi;
  const uint32_t slot0 = a.slot_base[tile];
  const float bx0 = (float)((tile % gx) * RIGGS_TILE + (sub & 1) * 8), by0 = (float)((tile / gx) * RIGGS_TILE + (sub >> 1) * 4);
  FwWalk w;
  w.done = !px.inside; w.T = 1.0f; w.Tstop = -1.0f; w.last = 0u;
  w.C0 = 0.f; w.C1 = 0.f; w.C2 = 0.f; w.D = 0.f; w.A = 0.f;
  const unsigned long long t_begin = (TRACE && a.trace) ? wall_clock64() : 0ull;
  uint32_t st_rounds = 0, st_surv = 0, st_iters = 0, st_full = 0;
  if (TRACE && a.trace && lane == 0 && (uint64_t)item.index < fw_late_args()->trace_items) a.trace[((size_t)item.index * 4 + wave) * 8 + 7] = 0ull;
  // checkpoints are held one round (lane i keeps chunk i's) and stored ahead of the next round's loads
  float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f;
  bool hv = false;
  int hbase = 0;
  bool ck_on = true;  // (owner) false while it walks a segment whose checkpoints a helper stores
  auto flush_ckpt = [&]() {
    if (hv) {
      float* ck = a.ckpt + ((size_t)(slot0 + (hbase >> 6) + i) * 5) * 256 + px.pix;
      ck[0] = h0; ck[256] = h1; ck[512] = h2; ck[768] = h3; ck[1024] = h4;
    }
    hv = false;
  };
  // prefetch registers for the next round (one instance per thread) and the list entry of the round after it
  float4 n_xy = make_float4(0.f, 0.f, 0.f, 0.f), n_co = n_xy, n_cc = n_xy;
  uint32_t n_id = 0u;
  if (lo + B + tid < hi) n_id = a.point_list[range.x + lo + B + tid];
  if (lo + tid < hi) {
    const uint32_t id = a.point_list[range.x + lo + tid];
    n_xy = a.xyd[id]; n_co = a.conic_o[id]; n_cc = a.rgb[id];
  }
  uint32_t dnext = 0u;              // (helper) the block's dead_from word, read a round ahead
  uint32_t claim = 0u;              // (owner, thread 0) what the claim of the next segment returned, requested a round ahead
  bool dead = false;
  int resume = 0;  // (owner) the segment from which fw_owner_rest takes over, if any
  for (int base = lo; base < hi; base += B) {
    if (helper) {
      dead = dead || (dnext != 0u && (uint32_t)seg >= dnext);
      w.done = w.done || dead;
      dnext = ld_agent_u(a.dead_from + (size_t)tile * 8 + sub);
    }
    const bool boundary = owner_multi && base > 0 && (base & (RIGGS_SEG - 1)) == 0;  // the owner enters a new segment
    if (boundary && tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(claim) :: "memory");  // (the claim's result, requested a round ago)
      fw_word = claim;
    }
    if (__syncthreads_count(w.done) == 256) break;
    if (boundary) {
      const uint32_t f = fw_word;
      if (f & SEG_SUMMARY) { resume = base / RIGGS_SEG; break; }  // a helper has composited this segment: the rest of the walk is fw_owner_rest's
      ck_on = !(f & SEG_STARTED);
      if (ck_on) {
        // the segment's checkpoints are this workgroup's, absolute ones: the backward finds the identity as the segment's prefix
        if (i < 5) *fw_state_ptr(a.seg_state, item, px.pix, base / RIGGS_SEG, RIGGS_SEG_PREFIX + i) = (i == 0) ? 1.0f : 0.f;
      } else {
        // a helper is at work on this segment and stores its (segment-local) checkpoints: walk it for the state only, and
        // leave the true prefix for the backward
        const float k0 = oct_sum(w.C0), k1 = oct_sum(w.C1), k2 = oct_sum(w.C2), kd = oct_sum(w.D);
        if (i < 5) *fw_state_ptr(a.seg_state, item, px.pix, base / RIGGS_SEG, RIGGS_SEG_PREFIX + i) = (i == 0) ? w.T : (i == 1) ? k0 : (i == 2) ? k1 : (i == 3) ? k2 : kd;
      }
    }
    if (owner_multi && (base & (RIGGS_SEG - 1)) == RIGGS_SEG - B && base + B < hi && tid == 0) {
      // the last round of a segment: claim the next one.  The returning atomic is written in assembly so that its result is
      // waited for where it is read, one round later (the compiler waits for a result defined under a divergent branch at the
      // end of the branch: a round trip to the memory side on the critical path of every segment).  The hardware retires
      // vector memory operations in order, so the compiler's own counts stay conservative.
      const uint32_t* fp = fw_flag_ptr(a.seg_flags, item, base / RIGGS_SEG + 1);
      asm volatile("global_atomic_or %0, %1, %2, off sc0" : "=v"(claim) : "v"(fp), "v"(SEG_CLAIM) : "memory");
    }
    {
      const int cnt = fw_stage_round(tid, base + tid < hi, n_xy, n_co, n_cc, bx0, by0);
      if constexpr (TRACE) st_surv += (uint32_t)cnt;
    }
    if constexpr (TRACE) st_rounds++;
    flush_ckpt();
    hbase = base;
    __syncthreads();
    {
      n_xy = make_float4(0.f, 0.f, 0.f, 0.f); n_co = n_xy; n_cc = n_xy;
      if (base + B + tid < hi) {
        const uint32_t id = n_id;
        n_xy = a.xyd[id]; n_co = a.conic_o[id]; n_cc = a.rgb[id];
      }
      if (base + 2 * B + tid < hi) n_id = a.point_list[range.x + base + 2 * B + tid];
    }
    if (helper && base > lo) {  // a helper keeps the state at the start of its rounds for the owner
      const float k0 = oct_sum(w.C0), k1 = oct_sum(w.C1), k2 = oct_sum(w.C2), kd = oct_sum(w.D), fa = oct_sum(w.A);
      const uint32_t lf = oct_max_u(w.last);
      if (i == 0) {
        const int bp = wave * 8 + pl;
        float (*e)[32] = fw_rs[(base - lo) / B - 1];
        e[0][bp] = w.done ? 0.f : w.T; e[1][bp] = k0; e[2][bp] = k1; e[3][bp] = k2; e[4][bp] = kd; e[5][bp] = fa; e[6][bp] = __uint_as_float(lf);
      }
    }
#pragma unroll 1
    for (int k = 0; k < B / 64; k++) {
      const int cbase = base + 64 * k;
      if (cbase >= hi) break;
      if (__builtin_amdgcn_ballot_w64(!w.done) == 0) break;
      {
        // checkpoint of the state BEFORE instance cbase: fold the eight lanes' partial sums
        const float k0 = oct_sum(w.C0), k1 = oct_sum(w.C1), k2 = oct_sum(w.C2), kd = oct_sum(w.D);
        if (i == k) { h0 = w.T; h1 = k0; h2 = k1; h3 = k2; h4 = kd; hv = !w.done && ck_on; }
      }
      fw_composite_chunk<TRACE>(w, k, base, lane, pfx, pfy, st_iters, st_full);
    }
  }
  if (TRACE && a.trace && lane == 0 && (uint64_t)item.index < fw_late_args()->trace_items) {
    unsigned long long* tr = a.trace + ((size_t)item.index * 4 + wave) * 8;
    tr[6] = t_begin;
    tr[0] = wall_clock64() - t_begin; tr[1] = st_rounds;
    tr[2] = (unsigned long long)st_surv | ((unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xFFFFu) << 32) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 0xFu) << 48);
    tr[3] = st_iters; tr[4] = st_full;
    tr[5] = (unsigned long long)(hi - lo) | ((unsigned long long)tile << 32) | ((unsigned long long)seg << 48) | (helper ? 1ull << 63 : 0ull);
  }
  flush_ckpt();
  // fold the eight lanes (every lane of a pixel ends up with the same values)
  const float k0 = oct_sum(w.C0), k1 = oct_sum(w.C1), k2 = oct_sum(w.C2), kd = oct_sum(w.D), ka = oct_sum(w.A);
  const float ts = oct_max(w.Tstop);
  const uint32_t lm = oct_max_u(w.last);
  if (helper) {
    if (__syncthreads_or(dead)) return;  // (the block is finished: nobody reads this segment)
    FwEnd end;
    end.T = w.T; end.k0 = k0; end.k1 = k1; end.k2 = k2; end.kd = kd; end.ka = ka; end.lm = lm; end.done = w.done;
    fw_publish(a.seg_state, a.seg_flags, item, px.pix, end);
    return;
  }
  float Tfin = (ts >= 0.f) ? ts : w.T, o0 = k0, o1 = k1, o2 = k2, od = kd, oa = ka;
  uint32_t on = lm;
  int reached = 1;  // (owner of a segmented tile) number of segments its pixels went through
  if (resume) {
    if (i == 0) {
      const int bp = wave * 8 + pl;
      fw_st[0][bp] = Tfin; fw_st[1][bp] = k0; fw_st[2][bp] = k1; fw_st[3][bp] = k2; fw_st[4][bp] = kd; fw_st[5][bp] = ka;
      fw_st[6][bp] = __uint_as_float(lm); fw_st[7][bp] = w.done ? 1.0f : 0.f;
    }
    __syncthreads();
    reached = fw_owner_rest<TRACE>((const RenderArgs*)(const void*)fw_late_args(), item, resume);
    __syncthreads();
    const int bp = wave * 8 + pl;
    Tfin = fw_st[0][bp]; o0 = fw_st[1][bp]; o1 = fw_st[2][bp]; o2 = fw_st[3][bp]; od = fw_st[4][bp]; oa = fw_st[5][bp];
    on = __float_as_uint(fw_st[6][bp]);
  } else if (owner_multi && tid == 0) {
    // every pixel of the block is final: helpers of the segments that were not reached have nothing to do
    reached = min((total + RIGGS_SEG - 1) / RIGGS_SEG, hbase / RIGGS_SEG + 1);
    __hip_atomic_store((seg_gu32*)(a.dead_from + (size_t)tile * 8 + sub), (uint32_t)reached, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // statistics for the NEXT frame's work list (bin_offsets_body): how deep the walks of segmented tiles get
  if (owner_multi && tid == 0) atomicAdd(fw_late_args()->seg_stats + min(max(reached, 1) - 1, 31), 1u);
  {
    FwLateArgs& l = *fw_late_args();
    if (px.inside && i == 0) {
      const size_t pid = (size_t)px.pyi * l.W + px.pxi, HW = (size_t)l.H * l.W;
      l.final_T[pid] = Tfin;
      l.n_contrib[pid] = on;
      l.final_acc[pid] = make_float4(o0, o1, o2, od);
      l.out_color[pid] = o0 + Tfin * l.bg[0];
      l.out_color[HW + pid] = o1 + Tfin * l.bg[1];
      l.out_color[2 * HW + pid] = o2 + Tfin * l.bg[2];
      l.out_depth[pid] = od;
      l.out_alpha[pid] = oa;
    }
  }
  uint32_t m = px.inside ? on : 0u;
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if (lane == 0) fw_wmax[wave] = m;
  __syncthreads();  // (fw_wmax is complete)
  // ---- work list of the backward.  A tile's instances past its last contributor (tile_max = max n_contrib over its 256
  // pixels) need no backward, so the list holds one entry per 64-instance chunk below it.  Every block of the tile folds
  // its maximum into tile_max and takes a ticket; the block that draws the last ticket knows the final value and appends
  // the tile's entries (tile, chunk, start of the tile's list, instances to walk) — a separate list-building launch used to
  // cost 10 us.  Both words are only ever touched with agent-scope atomics, and a block takes its ticket after its
  // maximum has RETURNED (the returning atomic has been performed), so the last ticket holder reads the final maximum.
  // The list's size word (work_ctr) lives on a cache line of ITS OWN: sharing one with item_ctr — which every one of the
  // launch's 20 000 workgroups reads when it starts — cost 17 us (each of the ~470 atomics throws the line out of the L2s,
  // and the late-dispatched workgroups queue behind it at the memory side).
  // Only wave 0 stays for this (two dependent atomic round trips to the memory side, ~4 us): the other three waves leave.
  if (wave == 0) {
    FwLateArgs& l = *fw_late_args();
    uint32_t n_c = 0u, base = 0u, limit = 0u;
    if (lane == 0) {
      const uint32_t mb = max(max(fw_wmax[0], fw_wmax[1]), max(fw_wmax[2], fw_wmax[3]));
      const uint32_t before = __hip_atomic_fetch_max(&l.tile_max[tile], mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t ticket = __hip_atomic_fetch_add(&l.tile_ticket[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket == 7u) {  // eight 8 x 4 blocks per tile
        limit = max(max(before, mb), __hip_atomic_load(&l.tile_max[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        limit = min((uint32_t)total, limit);
        n_c = (limit + 63u) >> 6;
        if (n_c) base = __hip_atomic_fetch_add(l.work_ctr, 4u * n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 2;  // (quarter-chunks)
      }
    }
    n_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_c);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    limit = (uint32_t)__builtin_amdgcn_readfirstlane((int)limit);
    for (uint32_t k = lane; k < n_c; k += 64) l.work[base + k] = make_uint4((uint32_t)tile, k, range.x, limit);
  }
}

// helper workgroups per forward launch (RIGGS_FWD_HELPERS overrides: a tuning knob for tools, 0 turns the helpers off)
int64_t forward_helper_budget() {
  static int64_t v = -1;
  if (v < 0) {
    const char* e = getenv("RIGGS_FWD_HELPERS");
    v = e ? atoll(e) : 8192;
  }
  return v;
}

int launch_render_fwd(const RenderArgs& a, hipStream_t s) {
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE, gy = (a.H + RIGGS_TILE - 1) / RIGGS_TILE;
  if (gx * gy == 0) return 0;
  // one workgroup per work item: the walkers of all tiles (they come first in the list; the ones of empty tiles only help with
  // the background and leave) and as many helpers as fit a BUDGET — helpers are optional, the list has the ones of the longest
  // lists first, and a helper that finds nothing to do still costs a workgroup launch and a round trip to memory (20 000 of
  // them behind the last walker: +8 us on a 60 us launch).  (tile_max, the tile tickets, the hand-shake words and the
  // work-list size were cleared by the extra workgroup of bin_scatter_kernel.)
  const int64_t walkers = (int64_t)gx * gy * 8;
  int64_t helpers = a.items ? a.n_item_slots * 8 - walkers : 0;
  const int64_t budget = forward_helper_budget();
  if (helpers > budget) helpers = budget;
  if (helpers < 0) helpers = 0;
  const int64_t blocks = walkers + helpers;
  if (blocks > 0x7FFFFFFF) { set_error("image too large for one forward launch"); return 2; }
  if (a.trace) hipLaunchKernelGGL(render_fwd_oct_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(render_fwd_oct_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return 0;
}

// ------------------------------------------------------------------ render backward
// ---- wave64 scans (DPP): two independent exclusive scans at once, hand-scheduled.  Written with generic builtins the
// compiler expands every step into v_mov (identity) + v_mov_dpp + v_op and pads the DPP read-after-write hazard (2 wait states)
// with s_nop, one chain after the other: ~21 slots per product scan.  Here a step is ONE v_op_dpp in place — lanes
// whose DPP source is invalid or masked off are simply not written (bound_ctrl:0), which IS the identity — and the
// second chain fills the first one's hazard slots: 6 x (2 ops + s_nop 0) + 2 shifts for both scans.
#define DUAL_SCAN_STEP(OP, CTRL) \
  OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
#define DUAL_SCAN_BODY(OP)                                                   \
  "s_nop 1\n\t"                                                              \
  DUAL_SCAN_STEP(OP, "row_shr:1 row_mask:0xf bank_mask:0xf")                 \
  DUAL_SCAN_STEP(OP, "row_shr:2 row_mask:0xf bank_mask:0xf")                 \
  DUAL_SCAN_STEP(OP, "row_shr:4 row_mask:0xf bank_mask:0xf")                 \
  DUAL_SCAN_STEP(OP, "row_shr:8 row_mask:0xf bank_mask:0xf")                 \
  DUAL_SCAN_STEP(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf")              \
  DUAL_SCAN_STEP(OP, "row_bcast:31 row_mask:0xc bank_mask:0xf")              \
  "v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"           \
  "v_mov_b32_dpp %3, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"           \
  "s_nop 0"
__device__ __forceinline__ void dual_excl_prod_scan(float& a, float& b) {
  float ea = 1.0f, eb = 1.0f;  // lane 0 has no source in the final shift: it keeps the identity
  asm volatile(DUAL_SCAN_BODY("v_mul_f32_dpp") : "+v"(a), "+v"(b), "+v"(ea), "+v"(eb));
  a = ea; b = eb;
}
__device__ __forceinline__ void dual_excl_sum_scan(float& a, float& b) {
  float ea = 0.0f, eb = 0.0f;
  asm volatile(DUAL_SCAN_BODY("v_add_f32_dpp") : "+v"(a), "+v"(b), "+v"(ea), "+v"(eb));
  a = ea; b = eb;
}

// Chunk-parallel, instance-major backward.  One wave64 owns one chunk of 64 consecutive
// instances of one tile: lane <-> instance, and the wave walks the tile's 256 pixels.  For a
// pixel the 64 transmittances are an exclusive product scan over the lanes seeded with the
// forward's checkpoint, and dL/dalpha needs one more scan (prefix of the projected colour
//   k = gC.c + gD.z + gA), the suffix being  total - prefix.  Per-instance gradients
// accumulate in registers over the 256 pixels — no cross-lane reduction, no atomic
// contention (one atomic per value per (tile, instance) at the end).  Chunks are independent,
// so a tile with thousands of contributing instances spreads over the whole chip instead of
// serialising on four waves.
// Persistent workgroups over the device-built list of (tile, 64-instance chunk) items, dealt round-robin: deep tiles
// (dozens of fully active chunks) spread over the chip.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v splat2(float v) { return f2v{v, v}; }

// NW waves per workgroup = parts the tile's 256 pixels are split into; ORDERED = rows instead of atomics (cfg.deterministic);
// TRACE = per-chunk statistics (both compiled out of the default instantiation: the ordered branch alone cost 4 VGPRs = one
// wave per SIMD = 7 us)
template <int NW, bool ORDERED, bool TRACE>
__global__ __launch_bounds__(64 * NW, 4) void render_bwd_kernel(RenderBwdArgs a) {
  constexpr int PPW = 256 / NW;    // pixels per wave
  constexpr int RSTEP = NW;        // a wave's rows are part, part + NW, ...
  // per-pixel state of the chunk, interleaved per PAIR of neighbouring pixels (A, B): one LDS read delivers the two
  // operands of a packed fp32 instruction in adjacent registers
  __shared__ float4 s_q0[NW][PPW / 2];  // (T_A, T_B, Pre_A, Pre_B)
  __shared__ float4 s_q1[NW][PPW / 2];  // (Qb_A, Qb_B, n_A, n_B as float bits)
  __shared__ float4 s_q2[NW][PPW / 2];  // (gC0_A, gC0_B, gC1_A, gC1_B)
  __shared__ float4 s_q3[NW][PPW / 2];  // (gC2_A, gC2_B, gD_A, gD_B)
  __shared__ float2 s_q4[NW][PPW / 2];  // (gA_A, gA_B)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t n_items = a.work_ctr[0];
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const size_t HW = (size_t)a.H * a.W;
  const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
  const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
  // static interleaving (a single dequeue word saturates at ~90 dequeues/us on this chip): workgroup b takes
  // chunks b, b + #workgroups, ...; its four waves are the four pixel-quarters of the chunk and meet in
  // LDS so that each (chunk, instance) still issues ONE set of atomics (float atomics are the scarce
  // resource: ~30 ns each once millions are in flight)
  __shared__ float s_red[NW - 1][64][11];  // ten partial sums and the 'has a contribution' flag
  const uint32_t n_chunks = n_items >> 2;
  const int quarter = wave;
  // a wave's pixels are rows part, part + NW, ... of the tile: every wave then sees the same mix of instances and
  // live pixels, and they meet at the barrier at about the same time
  const int spix = ((lane >> 4) * RSTEP + quarter) * 16 + (lane & 15);  // this lane's pixel while staging (lane < PPW)
  // A chunk costs about as much arithmetic as a few global round trips, and its inputs hang off a chain of them
  // (work entry -> list entry / n_contrib -> records / per-pixel state), so the chain is software-pipelined across
  // the workgroup's chunks: the work entry (self-contained: tile, chunk, list start, limit; read
  // through the scalar cache) is fetched three chunks ahead, the list entry of this lane's instance and the
  // n_contrib of this lane's pixel two ahead, and the records / per-pixel state of the NEXT chunk are requested
  // right after this chunk's arithmetic and consumed after the fold of the partial sums — BEFORE this chunk's float
  // atomics are issued: vmcnt retires in order, so loads queued behind the atomics would wait for every
  // acknowledgement (microseconds once millions are in flight) with all the workgroup's waves at the barrier.
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  typedef const u4v __attribute__((address_space(4))) c_u4;
  c_u4* work = (c_u4*)(uintptr_t)a.work;
  const uint32_t stride = gridDim.x;
  auto fetch_level2 = [&](const u4v wk, uint32_t& id_out, uint32_t& n_out) {
    const int tile = (int)wk.x, pos = (int)wk.y * 64 + lane;
    id_out = 0u; n_out = 0u;
    if (pos < (int)wk.w) id_out = a.point_list[wk.z + pos];
    const int pxi = (tile % gx) * RIGGS_TILE + (spix & 15), pyi = (tile / gx) * RIGGS_TILE + (spix >> 4);
    if (lane < PPW && pxi < a.W && pyi < a.H) n_out = a.n_contrib[(size_t)pyi * a.W + pxi];
  };
  struct Level3 {  // what a lane loads for a chunk: its instance's records, its pixel's state
    float4 xy, co, cc, acc;
    float Tn, g0, g1, g2, gD, gA, Ts, S0, S1, S2, Ds;
    float pT, p0, p1, p2, pD;  // segmented tiles: the prefix in front of the chunk's segment (checkpoints are segment-local)
  };
  auto issue_level3 = [&](const u4v wk, uint32_t id, uint32_t n, Level3& r) {
    const int tile = (int)wk.x, pos0 = (int)wk.y * 64;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    r.xy = z; r.co = z; r.cc = z; r.acc = z;
    r.Tn = 0.f; r.g0 = 0.f; r.g1 = 0.f; r.g2 = 0.f; r.gD = 0.f; r.gA = 0.f; r.Ts = 1.f; r.S0 = 0.f; r.S1 = 0.f; r.S2 = 0.f; r.Ds = 0.f;
    r.pT = 1.f; r.p0 = 0.f; r.p1 = 0.f; r.p2 = 0.f; r.pD = 0.f;
    if (pos0 + lane < (int)wk.w) { r.xy = a.xyd[id]; r.co = a.conic_o[id]; r.cc = a.rgb[id]; }
    if ((int)n > pos0) {  // (n is 0 for the lanes without a pixel and for the pixels outside the image)
      const int pxi = (tile % gx) * RIGGS_TILE + (spix & 15), pyi = (tile / gx) * RIGGS_TILE + (spix >> 4);
      const size_t pid = (size_t)pyi * a.W + pxi;
      const float* ck = a.ckpt + ((size_t)((wk.z >> 6) + wk.x + wk.y) * 5) * 256;
      r.Tn = a.final_T[pid];
      r.acc = a.final_acc[pid];
      r.g0 = a.dL_dcolor[pid]; r.g1 = a.dL_dcolor[HW + pid]; r.g2 = a.dL_dcolor[2 * HW + pid];
      r.gD = a.dL_ddepth ? a.dL_ddepth[pid] : 0.f;
      r.gA = a.dL_dalpha ? a.dL_dalpha[pid] : 0.f;
      r.Ts = ck[spix]; r.S0 = ck[256 + spix]; r.S1 = ck[512 + spix]; r.S2 = ck[768 + spix]; r.Ds = ck[1024 + spix];
      const uint32_t sg = wk.y / RIGGS_SEG_CHUNKS;
      if (sg > 0u) {  // (only lists longer than RIGGS_SEG have chunks beyond the first segment)
        const float* sp = a.seg_state + ((size_t)(wk.z / RIGGS_SEG + wk.x + sg) * RIGGS_SEG_WORDS + RIGGS_SEG_PREFIX) * 256 + spix;
        r.pT = sp[0]; r.p0 = sp[256]; r.p1 = sp[512]; r.p2 = sp[768]; r.pD = sp[1024];
      }
    }
  };
  auto stage_pixels = [&](const u4v wk, uint32_t n, const Level3& r) {  // this wave's pixels (one per lane) -> LDS
    const int pos0 = (int)wk.y * 64;
    float4 pa = make_float4(1.f, 0.f, 0.f, 0.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
    float pc = 0.f;
    if ((int)n > pos0) {
      const float Ts = r.pT * r.Ts;
      const float S0 = r.p0 + r.pT * r.S0, S1 = r.p1 + r.pT * r.S1, S2 = r.p2 + r.pT * r.S2, Ds = r.pD + r.pT * r.Ds;
      const float pre = r.g0 * S0 + r.g1 * S1 + r.g2 * S2 + r.gD * Ds + r.gA * (1.0f - Ts);
      const float qb = (r.g0 * r.acc.x + r.g1 * r.acc.y + r.g2 * r.acc.z + r.gD * r.acc.w + r.gA * (1.0f - r.Tn)) +
                       r.Tn * (bg0 * r.g0 + bg1 * r.g1 + bg2 * r.g2);
      pa = make_float4(Ts, pre, qb, __uint_as_float(n));
      pb = make_float4(r.g0, r.g1, r.g2, r.gD);
      pc = r.gA;
    }
    {
      if (lane < PPW) {
        const int pp = lane >> 1, h = lane & 1;
        float* q0 = reinterpret_cast<float*>(&s_q0[wave][pp]) + h; q0[0] = pa.x; q0[2] = pa.y;
        float* q1 = reinterpret_cast<float*>(&s_q1[wave][pp]) + h; q1[0] = pa.z; q1[2] = pa.w;
        float* q2 = reinterpret_cast<float*>(&s_q2[wave][pp]) + h; q2[0] = pb.x; q2[2] = pb.y;
        float* q3 = reinterpret_cast<float*>(&s_q3[wave][pp]) + h; q3[0] = pb.z; q3[2] = pb.w;
        reinterpret_cast<float*>(&s_q4[wave][pp])[h] = pc;
      }
    }
  };
  const u4v wk_zero = {0u, 0u, 0u, 0u};
  u4v wk_a = wk_zero, wk_b = wk_zero, wk_c = wk_zero;  // this chunk, the next, the one after
  uint32_t id_a = 0u, n_a = 0u, id_b = 0u, n_b = 0u;
  float4 xy = make_float4(0.f, 0.f, 0.f, 0.f), co = xy, cc = xy;
  if (blockIdx.x < n_chunks) {
    wk_a = work[blockIdx.x];
    fetch_level2(wk_a, id_a, n_a);
    if (blockIdx.x + stride < n_chunks) { wk_b = work[blockIdx.x + stride]; fetch_level2(wk_b, id_b, n_b); }
    if (blockIdx.x + 2 * stride < n_chunks) wk_c = work[blockIdx.x + 2 * stride];
    Level3 r;
    issue_level3(wk_a, id_a, n_a, r);
    stage_pixels(wk_a, n_a, r);
    xy = r.xy; co = r.co; cc = r.cc;
  }
  for (uint32_t chunk_item = blockIdx.x; chunk_item < n_chunks; chunk_item += stride) {
    const u4v wk = wk_a;
    const uint32_t id = id_a;
    const unsigned long long t_begin = (TRACE && a.trace) ? wall_clock64() : 0ull;
    const int tile = (int)wk.x, chunk = (int)wk.y;
    const int limit = (int)wk.w;
    const int pos0 = chunk * 64;
    const int tx0 = (tile % gx) * RIGGS_TILE, ty0 = (tile / gx) * RIGGS_TILE;
    const int pos = pos0 + lane;
    const bool active = pos < limit;
    float m_x = 0.f, m_y = 0.f;
    bool touched = false;  // this lane's instance contributes to at least one pixel of the tile
    float a_mx = 0.f, a_my = 0.f, a_ca = 0.f, a_cb = 0.f, a_cc = 0.f, a_op = 0.f, a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f;
    // (the wave only touches its own LDS region: LDS operations of one wave are ordered, no barrier needed)
    // two pixels per iteration: their scans are independent chains that the scheduler interleaves (a lone
    // chain leaves the SIMD idle through every DPP / transcendental latency)
    {
      // packed fp32 (v_pk_mul / v_pk_fma / v_pk_add_f32: two fp32 operations per lane and instruction) across the
      // pixel pair: the per-instance operands are splat, the per-pixel ones arrive as (A, B) pairs from LDS, and the
      // ten accumulators are pairs that are folded once per chunk
      f2v m_x2 = splat2(0.f), m_y2 = m_x2, ca2 = m_x2, cb2 = m_x2, cc2 = m_x2, op2 = m_x2, r2 = m_x2, g2 = m_x2, b2 = m_x2, d2 = m_x2;
      for (int pl = 0; pl < PPW; pl += 2) {
        const float4 q1 = s_q1[wave][pl >> 1];
        const int nA = (int)__float_as_uint(q1.z), nB = (int)__float_as_uint(q1.w);
        if (nA <= pos0 && nB <= pos0) continue;  // wave-uniform: this chunk lies behind both pixels' last contributors
        const int pix = ((pl >> 4) * RSTEP + quarter) * 16 + (pl & 15);  // pl is even: both pixels are in the same row
        const float pfx = (float)(tx0 + (pix & 15)), pfy = (float)(ty0 + (pix >> 4));
        const float dxA = xy.x - pfx, dy = xy.y - pfy;
        const f2v dx = {dxA, dxA - 1.0f};
        // cheap conservative reject (alpha >= 1/255 extents, as in the forward's cull) before the exponentials
        if (__builtin_amdgcn_ballot_w64(active && fabsf(dy) <= cc.w && fminf(fabsf(dx.x), fabsf(dx.y)) <= xy.w) == 0) continue;
        const float cyy = co.z * dy * dy, cody = co.y * dy;
        const f2v pw = -0.5f * (splat2(co.x) * dx * dx + splat2(cyy)) - splat2(cody) * dx;
        const f2v pl2 = pw * splat2(LOG2E);
        const f2v Gr = {__builtin_amdgcn_exp2f(pl2.x), __builtin_amdgcn_exp2f(pl2.y)};
        const f2v ar = splat2(co.w) * Gr;
        const float alA_ = fminf(ALPHA_MAX, ar.x), alB_ = fminf(ALPHA_MAX, ar.y);
        const bool vA = active && (pos < nA) && (pw.x <= 0.0f) && (alA_ >= ALPHA_MIN);
        const bool vB = active && (pos < nB) && (pw.y <= 0.0f) && (alB_ >= ALPHA_MIN);
        if (__builtin_amdgcn_ballot_w64(vA || vB) == 0) continue;
        touched = touched || vA || vB;
        const f2v al = {vA ? alA_ : 0.f, vB ? alB_ : 0.f};
        const f2v G = {vA ? Gr.x : 0.f, vB ? Gr.y : 0.f};
        const f2v om = splat2(1.0f) - al;
        float scA = om.x, scB = om.y;
        dual_excl_prod_scan(scA, scB);
        const float4 q0 = s_q0[wave][pl >> 1], q2 = s_q2[wave][pl >> 1], q3 = s_q3[wave][pl >> 1];
        const float2 q4 = s_q4[wave][pl >> 1];
        const f2v Tl = f2v{q0.x, q0.y} * f2v{scA, scB};
        const f2v w = al * Tl;
        const f2v gc0 = {q2.x, q2.y}, gc1 = {q2.z, q2.w}, gc2 = {q3.x, q3.y}, gd = {q3.z, q3.w};
        const f2v k = gc0 * splat2(cc.x) + gc1 * splat2(cc.y) + gc2 * splat2(cc.z) + gd * splat2(xy.z) + f2v{q4.x, q4.y};
        const f2v wk2 = w * k;
        float ssA = wk2.x, ssB = wk2.y;
        dual_excl_sum_scan(ssA, ssB);
        const f2v pre = f2v{q0.z, q0.w} + f2v{ssA, ssB};
        // dL/dalpha = T k - (suffix + T_final * bg.g) / (1 - alpha),  suffix = total - prefix - own
        // (no select on dLa: an invalid pair has G = 0 and every use below is multiplied by G)
        const f2v rom = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
        const f2v dLa = Tl * k - (f2v{q1.x, q1.y} - pre - wk2) * rom;
        // raw moments of q = dL/dalpha * G; opacity and the conic factors of dL/dmean2D are applied once per chunk
        const f2v q = dLa * G;
        const f2v qx = q * dx, qy = q * splat2(dy);
        m_x2 += qx; m_y2 += qy;
        ca2 += qx * dx; cb2 += qx * splat2(dy); cc2 += qy * splat2(dy);
        op2 += q;
        r2 += w * gc0; g2 += w * gc1; b2 += w * gc2; d2 += w * gd;
      }
      m_x = m_x2.x + m_x2.y; m_y = m_y2.x + m_y2.y;
      a_ca = ca2.x + ca2.y; a_cb = cb2.x + cb2.y; a_cc = cc2.x + cc2.y; a_op = op2.x + op2.y;
      a_r = r2.x + r2.y; a_g = g2.x + g2.y; a_b = b2.x + b2.y; a_d = d2.x + d2.y;
    }
    a_mx = -co.w * (co.x * m_x + co.y * m_y);
    a_my = -co.w * (co.z * m_y + co.y * m_x);
    a_ca *= co.w; a_cb *= co.w; a_cc *= co.w;
    // ---- request the next chunk's records / pixel state, the list entry and n_contrib of the one after, and the
    // work entry after that
    const bool has_next = chunk_item + stride < n_chunks;
    Level3 nr;
    if (has_next) issue_level3(wk_b, id_b, n_b, nr);
    uint32_t id_c = 0u, n_c = 0u;
    if (chunk_item + 2 * stride < n_chunks) fetch_level2(wk_c, id_c, n_c);
    u4v wk_d = wk_zero;
    if (chunk_item + 3 * stride < n_chunks) wk_d = work[chunk_item + 3 * stride];
    // ---- fold the pixel parts
    __syncthreads();
    if (wave > 0) {
      float* r = s_red[wave - 1][lane];
      r[0] = a_mx; r[1] = a_my; r[2] = a_ca; r[3] = a_cb; r[4] = a_cc; r[5] = a_op; r[6] = a_r; r[7] = a_g; r[8] = a_b; r[9] = a_d; r[10] = touched ? 1.f : 0.f;
    }
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
      for (int q = 0; q < NW - 1; q++) {
        const float* r = s_red[q][lane];
        a_mx += r[0]; a_my += r[1]; a_ca += r[2]; a_cb += r[3]; a_cc += r[4]; a_op += r[5]; a_r += r[6]; a_g += r[7]; a_b += r[8]; a_d += r[9];
        touched = touched || (r[10] != 0.f);
      }
    }
    // ---- the next chunk's pixels go to LDS (each wave only touches its own region, and is done with it)
    if (has_next) { stage_pixels(wk_b, n_b, nr); xy = nr.xy; co = nr.co; cc = nr.cc; }
    // ---- only now the atomics, and only for the instances that reach a pixel of this tile: the lists are built
    // from 3-sigma rectangles, most of a tile's instances never get to alpha >= 1/255 inside it, and adding their
    // exact zeros cost a quarter of the kernel (the memory-side atomic units were its one saturated resource)
    if constexpr (ORDERED) {
      // ordered-reduction mode: the row of this tile instance, summed per Gaussian by launch_ordered_gather
      if (wave == 0 && active && touched) {
        float* r = a.det_rows + ((size_t)wk.z + (size_t)pos) * 10;
        r[0] = a_mx * ddelx_dx; r[1] = a_my * ddely_dy; r[2] = -0.5f * a_ca; r[3] = -a_cb; r[4] = -0.5f * a_cc;
        r[5] = a_op; r[6] = a_r; r[7] = a_g; r[8] = a_b; r[9] = a_d;
      }
    } else if (wave == 0 && active && touched) {
      float* g = a.gacc + (size_t)id * RIGGS_GACC;
      atomicAdd(g + 0, a_mx * ddelx_dx); atomicAdd(g + 1, a_my * ddely_dy);
      atomicAdd(g + 2, -0.5f * a_ca); atomicAdd(g + 3, -a_cb); atomicAdd(g + 4, -0.5f * a_cc);
      atomicAdd(g + 5, a_op); atomicAdd(g + 6, a_r); atomicAdd(g + 7, a_g); atomicAdd(g + 8, a_b);
      if (a.dL_ddepth) atomicAdd(g + 9, a_d);
    }
    wk_a = wk_b; id_a = id_b; n_a = n_b;
    wk_b = wk_c; id_b = id_c; n_b = n_c;
    wk_c = wk_d;
    if (TRACE && a.trace && threadIdx.x == 0) {  // per chunk: {start, end (100 MHz ticks), hardware id, workgroup}
      unsigned long long* tr = a.trace + (size_t)chunk_item * 4;
      tr[0] = t_begin; tr[1] = wall_clock64();
      tr[2] = (unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xFFFFu) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 0xFu) << 16);
      tr[3] = ((unsigned long long)blockIdx.x << 32) | (unsigned long long)((wk.x << 16) | (wk.y & 0xFFFFu));
    }
  }
}

// ---- ordered-reduction mode (cfg.deterministic) ----------------------------------------------------------------------
// exclusive scan of tiles_touched (one workgroup; a test / debugging mode: not tuned)
__global__ __launch_bounds__(1024) void ordered_offsets_kernel(int N, const uint32_t* __restrict__ tiles, uint32_t* __restrict__ off) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0u;
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + tid;
    const uint32_t c = (i < N) ? tiles[i] : 0u;
    uint32_t v = c;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)v, o);
      if (lane >= o) v += u;
    }
    if (lane == 63) s_w[wave] = v;
    __syncthreads();
    uint32_t run = s_carry;
    for (int w = 0; w < wave; w++) run += s_w[w];
    if (i < N) off[i] = run + v - c;
    __syncthreads();
    if (tid == 1023) s_carry = run + v;
    __syncthreads();
  }
  if (tid == 0) off[N] = s_carry;
}
// inv[off[g] + k] = list position of the k-th tile (row-major inside g's rectangle = ascending tile id) of Gaussian g
__global__ __launch_bounds__(256) void ordered_fill_kernel(int grid_x, int64_t cap, const uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ point_list,
                                                          const ushort4* __restrict__ rect, const uint32_t* __restrict__ off,
                                                          uint32_t* __restrict__ inv) {
  const int t = blockIdx.x, tx = t % grid_x, ty = t / grid_x;
  const uint2 rg = ranges[t];
  for (uint32_t p = rg.x + threadIdx.x; p < rg.y && (int64_t)p < cap; p += 256) {
    const uint32_t g = point_list[p];
    const ushort4 rc = rect[g];
    const int k = (ty - (int)rc.y) * ((int)rc.z - (int)rc.x) + (tx - (int)rc.x);
    inv[off[g] + (uint32_t)k] = p;
  }
}
__global__ __launch_bounds__(256) void ordered_gather_kernel(int N, int64_t cap, const uint32_t* __restrict__ tiles,
                                                            const uint32_t* __restrict__ off, const uint32_t* __restrict__ inv,
                                                            const float* __restrict__ rows, float* __restrict__ gacc,
                                                            int want_depth) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N) return;
  float acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const uint32_t n = tiles[g], o = off[g];
  for (uint32_t k = 0; k < n; k++) {
    if ((int64_t)(o + k) >= cap) break;  // (an overflowed arena: the frame is flagged invalid anyway)
    const float* r = rows + (size_t)inv[o + k] * 10;
#pragma unroll
    for (int c = 0; c < 10; c++) acc[c] += r[c];
  }
  float* out = gacc + (size_t)g * RIGGS_GACC;
#pragma unroll
  for (int c = 0; c < 9; c++) out[c] = acc[c];
  out[9] = want_depth ? acc[9] : 0.f;
  out[10] = 0.f; out[11] = 0.f;
}
int launch_ordered_gather(int N, int n_tiles, int grid_x, int64_t cap, const uint2* ranges, const uint32_t* point_list,
                          const uint32_t* tiles, const ushort4* rect, const float* det_rows, uint32_t* inv, uint32_t* off,
                          float* gacc, int want_depth, hipStream_t s) {
  hipLaunchKernelGGL(ordered_offsets_kernel, dim3(1), dim3(1024), 0, s, N, tiles, off);
  hipLaunchKernelGGL(ordered_fill_kernel, dim3(n_tiles), dim3(256), 0, s, grid_x, cap, ranges, point_list, rect, off, inv);
  hipLaunchKernelGGL(ordered_gather_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, cap, tiles, off, inv, det_rows, gacc,
                     want_depth);
  return 0;
}

int launch_render_bwd(const RenderBwdArgs& a, hipStream_t s) {
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE, gy = (a.H + RIGGS_TILE - 1) / RIGGS_TILE;
  if (gx * gy == 0) return 0;
  // 8 workgroups of 4 waves per CU: every SIMD holds 8 pulling waves (grid sizes from 3 to 128 per CU: within 2 %)
  const int64_t max_blocks = 256 * 8;
  const unsigned blocks = (unsigned)((a.n_slots < max_blocks) ? a.n_slots : max_blocks);
  if (a.trace) hipLaunchKernelGGL((render_bwd_kernel<4, false, true>), dim3(blocks), dim3(256), 0, s, a);
  else if (a.det_rows) hipLaunchKernelGGL((render_bwd_kernel<4, true, false>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((render_bwd_kernel<4, false, false>), dim3(blocks), dim3(256), 0, s, a);
  return 0;
}

}  // namespace riggs
