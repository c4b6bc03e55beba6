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
// ---- wide blocks ---------------------------------------------------------------------------------------------------------
// A workgroup's time for its block is (rounds of 256 instances it walks) x (one round), and a round is issue-bound: four
// waves, one per SIMD, ~45 vector instructions per scan step, 32 steps per round with eight lanes per pixel — 2.3 us.  A tile
// whose list is long and whose pixels do not saturate is walked to the end, alone: the kernel then lasts as long as the longest
// such list (headline: 25 rounds = 57 us for 35 - 40 us of issue; a scene of thin shells seen edge-on: 70 rounds = 160 us for
// 30 us of issue).  What shortens the walk is more SIMDs for the same pixels, and that means more workgroups per tile (a
// workgroup lives on one CU: more waves in it share the same four SIMDs).  So the tiles whose walk went deep in the previous
// frame (the first `n_wide` entries of the work list — binning.hip, bin_offsets_body: "fwd_wide_min" instances and more,
// at most "fwd_wide_tiles" tiles (riggs_set_option); a list's LENGTH says nothing: the bench scene's longest lists saturate within 800 instances)
// are composited with THIRTY-TWO lanes per pixel: a wave is 2 pixels x 32 instance lanes, a workgroup a 4 x 2 pixel block, a
// tile 32 workgroups instead of 8.  A scan step then covers 32 consecutive instances of a pixel (8 steps per round instead of
// 32) at about the same instruction count — the exclusive product scan is five DPP multiplies instead of the quad permutes —,
// the sequential stop rule is resolved by the same ballot, checkpoints and n_contrib keep their meaning, nothing is speculated
// and no two workgroups ever talk to each other.  The price is the staging: every one of the 32 workgroups gathers and culls
// the whole list (4x the L2 traffic of the 8-lane form: why this is for the few long lists only), against a 4 x 2 box that
// lets fewer instances through than the 8 x 4 one.
// (Round 3 first tried the other way — splitting long LISTS into segments of 1024 that helper workgroups composite from T = 1
// and the walking workgroup combines, DESIGN.md §4: it needs hand-shakes through memory (3 us round trips), re-composites the
// rounds in which pixels stop, speculates on segments nobody reaches, and gained 4 %.)
template <int LPP> struct FwGeom;
template <> struct FwGeom<8> {
  static constexpr int LEAD = 0;  // the lane of a pixel's group that holds folded values: every lane does
  static constexpr int BW = 8, BH = 4;
};
template <> struct FwGeom<32> {
  static constexpr int LEAD = 16;  // folded values land in the upper row of the group (lanes 16 .. 31)
  static constexpr int BW = 4, BH = 2;
};
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_BCAST15 0x142
#define ODPP_FR(old, v, ctrl, rows) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(v)), ctrl, rows, 0xf, false))
#define ODPP_UR(old, v, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, rows, 0xf, false))
// folds over the LPP lanes of a pixel; the result is valid in the lanes i >= LEAD of the group
template <int LPP> __device__ __forceinline__ float grp_sum(float v) {
  v += QUAD_F(v, QP(1, 0, 3, 2));
  v += QUAD_F(v, QP(2, 3, 0, 1));
  v += ODPP_F(0.f, v, DPP_HALF_MIRROR, 0xf);
  if constexpr (LPP == 32) {
    v += ODPP_F(0.f, v, DPP_ROW_MIRROR, 0xf);
    v += ODPP_FR(0.f, v, DPP_ROW_BCAST15, 0xa);
  }
  return v;
}
template <int LPP> __device__ __forceinline__ float grp_max(float v) {  // (of values >= -1)
  v = fmaxf(v, QUAD_F(v, QP(1, 0, 3, 2)));
  v = fmaxf(v, QUAD_F(v, QP(2, 3, 0, 1)));
  v = fmaxf(v, ODPP_F(-1.0f, v, DPP_HALF_MIRROR, 0xf));
  if constexpr (LPP == 32) {
    v = fmaxf(v, ODPP_F(-1.0f, v, DPP_ROW_MIRROR, 0xf));
    v = fmaxf(v, ODPP_FR(-1.0f, v, DPP_ROW_BCAST15, 0xa));
  }
  return v;
}
template <int LPP> __device__ __forceinline__ uint32_t grp_max_u(uint32_t v) {
  v = max(v, QUAD_U(v, QP(1, 0, 3, 2)));
  v = max(v, QUAD_U(v, QP(2, 3, 0, 1)));
  v = max(v, ODPP_U(0u, v, DPP_HALF_MIRROR, 0xf));
  if constexpr (LPP == 32) {
    v = max(v, ODPP_U(0u, v, DPP_ROW_MIRROR, 0xf));
    v = max(v, ODPP_UR(0u, v, DPP_ROW_BCAST15, 0xa));
  }
  return v;
}

// The four folds of a checkpoint at once, hand-scheduled: each step is ONE v_add_f32_dpp (the first writes the copy, the later
// ones work in place: every lane reads its partner inside its own row of 16 before the row is written back), and the four
// independent chains fill each other's DPP read-after-write slots.  12 instructions for eight lanes per pixel (20 for 32); the
// compiler's version of the same folds is mov_dpp + add pairs with zero-initialised temporaries: 26 (54) — a tenth of the
// instructions of a round in a kernel that is bound by their issue.
template <int LPP>
__device__ __forceinline__ void grp_sum4(const float a, const float b, const float c, const float d, float& ka, float& kb, float& kc, float& kd) {
#define FW_F4(CTRL) \
  "v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\tv_add_f32_dpp %2, %2, %2 " CTRL "\n\tv_add_f32_dpp %3, %3, %3 " CTRL "\n\t"
  if constexpr (LPP == 8) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        FW_F4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        FW_F4("row_half_mirror row_mask:0xf bank_mask:0xf")
        "s_nop 0"
        : "=&v"(ka), "=&v"(kb), "=&v"(kc), "=&v"(kd)
        : "v"(a), "v"(b), "v"(c), "v"(d));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        FW_F4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        FW_F4("row_half_mirror row_mask:0xf bank_mask:0xf")
        FW_F4("row_mirror row_mask:0xf bank_mask:0xf")
        FW_F4("row_bcast:15 row_mask:0xa bank_mask:0xf")
        "s_nop 0"
        : "=&v"(ka), "=&v"(kb), "=&v"(kc), "=&v"(kd)
        : "v"(a), "v"(b), "v"(c), "v"(d));
  }
#undef FW_F4
}

#define FW_B 256                    // instances per round
// LDS of the forward
__shared__ float4 fw_stage[3 * FW_B];     // the staged round: records of the survivors
#define fw_xyd (fw_stage)
#define fw_con (fw_stage + FW_B)
#define fw_rgb (fw_stage + 2 * FW_B)
__shared__ unsigned short fw_pos[FW_B];  // position of the survivor inside its batch
__shared__ int fw_cnt[FW_B / 64];        // survivors per chunk
__shared__ uint32_t fw_wmax[4];

// A workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the vector-memory counter: behind
// the staging of a round that is the checkpoint stores just issued (flush_ckpt) — a store's round trip in front of every round's
// compositing, for ordering nobody needs (no wave reads another's global stores inside the loop).
#define FW_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// a pixel's compositing state while its list is walked: T and the stop bookkeeping are the same in all lanes of its group,
// the sums are per-lane partials
struct FwWalk {
  float T, C0, C1, C2, D, A, Tstop;
  uint32_t last;
  bool done;
};

// the main loop's round: every wave stages one chunk of the 256 (one instance per lane), culled against the block's pixels,
// the survivors compacted (`pos`: the instance's position inside its round, what n_contrib is counted from)
template <int LPP>
__device__ __forceinline__ int fw_stage_round(const int tid, const bool in_range, const float4 xy, const float4 co, const float4 cc,
                                              const float bx0, const float by0) {
  const int chunk = tid >> 6, lane = tid & 63;
  const float bx1 = bx0 + (float)(FwGeom<LPP>::BW - 1), by1 = by0 + (float)(FwGeom<LPP>::BH - 1);
  const bool keep = in_range && ((xy.x + xy.w >= bx0) && (xy.x - xy.w <= bx1) && (xy.y + cc.w >= by0) && (xy.y - cc.w <= by1));
  const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
  const int cnt = __builtin_popcountll(mask);
  const int slot = chunk * 64 + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
  if (keep) { fw_xyd[slot] = xy; fw_con[slot] = co; fw_rgb[slot] = cc; fw_pos[slot] = (unsigned short)tid; }
  if (lane >= cnt && lane < ((cnt + LPP - 1) & ~(LPP - 1))) {  // null records up to the next multiple of a step
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    fw_xyd[chunk * 64 + lane] = z; fw_con[chunk * 64 + lane] = z; fw_rgb[chunk * 64 + lane] = z; fw_pos[chunk * 64 + lane] = 0;
  }
  if (lane == 0) fw_cnt[chunk] = cnt;
  return cnt;
}

// the survivors of chunk k of the staged round, LPP per step, for this wave's 64 / LPP pixels
template <int LPP, bool TRACE>
__device__ __forceinline__ void fw_composite_chunk(FwWalk& w, const int k, const int base, const int lane, const float pfx, const float pfy,
                                                   uint32_t& st_iters, uint32_t& st_full) {
  const int i = lane & (LPP - 1);
  const int nk = fw_cnt[k];
  for (int g = 64 * k; g < 64 * k + nk; g += LPP) {
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
    // one step: LPP consecutive instances (one per lane) of this lane's pixel
    const float om = valid ? 1.0f - alpha : 1.0f;
    float E, prod;  // exclusive product of the group's lanes in front of this one; product of the whole group
    if constexpr (LPP == 8) {
      // inside each quad first ...
      float b1 = QUAD_F(om, QP(0, 0, 1, 2)); b1 = ((i & 3) >= 1) ? b1 : 1.0f;     // [1, o0, o1, o2]
      float s1 = QUAD_F(b1, QP(0, 0, 1, 2)); s1 = ((i & 3) >= 1) ? s1 : 1.0f;     // [1, 1, o0, o1]
      float s2 = QUAD_F(b1, QP(0, 0, 0, 1)); s2 = ((i & 3) >= 2) ? s2 : 1.0f;     // [1, 1, 1, o0]
      E = b1 * s1 * s2;                                                           // [1, o0, o0 o1, o0 o1 o2] per quad
      // ... then the upper quad takes the lower quad's total (row_shr:4 written to banks 1 and 3 only)
      const float Pq = QUAD_F(E * om, QP(3, 3, 3, 3));                            // product of the own quad
      E *= ODPP_F(1.0f, Pq, DPP_ROW_SHR4, 0xA);
      // product of all eight: the upper quad's running total, handed down to the lower quad (row_shl:4, banks 0 and 2)
      const float X3 = QUAD_F(E * om, QP(3, 3, 3, 3));
      prod = ODPP_F(X3, X3, DPP_ROW_SHL4, 0x5);
    } else {
      // inclusive scan in place (a lane whose DPP source is invalid or masked off is not written: the identity), one row of 16
      // at a time, then the upper row of each half takes the lower row's total; the shift by one lane makes it exclusive
      float inc = om;
      E = 1.0f;  // (lane 0 has no source in the final shift)
      asm volatile(
          "s_nop 1\n\t"
          "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
          "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
          "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
          "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
          "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
          "v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 0"
          : "+v"(inc), "+v"(E));
      E = (i == 0) ? 1.0f : E;  // (lane 32 received the other pixel's total)
      const float p_lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inc), 31));
      const float p_hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inc), 63));
      prod = (lane & 32) ? p_hi : p_lo;
    }
    const float Tj = w.T * E;
    const float test_T = Tj * om;
    const bool sc = valid && (test_T < T_EPS);
    const uint64_t bits = __builtin_amdgcn_ballot_w64(sc);
    const uint32_t ob = (uint32_t)(bits >> (lane & (64 - LPP))) & (uint32_t)((1ull << LPP) - 1ull);
    const bool first_stop_before = (ob & ((1u << i) - 1u)) != 0u;
    const bool use = valid && !sc && !first_stop_before;
    const float wt = use ? alpha * Tj : 0.f;
    w.C0 += c.x * wt; w.C1 += c.y * wt; w.C2 += c.z * wt;
    w.D += xy.z * wt; w.A += wt;
    w.last = use ? (uint32_t)pos1 : w.last;
    if (sc && !first_stop_before) w.Tstop = Tj;  // transmittance in front of the instance that ends the pixel
    const bool nostop = (ob == 0u);
    w.T = (nostop && !w.done) ? w.T * prod : w.T;
    w.done = w.done || !nostop;
  }
}

// The kernel's argument block, read where it is needed: the compiler loads every kernel argument it sees into scalar registers
// at the kernel's entry and keeps them there — two registers per pointer, a dozen pointers that only the epilogue of a work
// item uses — and the main loop, which is short of them, then parks live scalars in vector-register lanes.  The epilogues
// read their pointers through an opaque copy of the kernarg pointer instead.
typedef const RenderArgs __attribute__((address_space(4))) FwLateArgs;
__device__ __forceinline__ FwLateArgs* fw_late_args() {
  FwLateArgs* p = (FwLateArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

// the end of a work item: the block's pixels leave, and the tile's last block appends the backward's work
template <int LPP>
__device__ __forceinline__ void fw_finish(const int tile, const int total, const uint2 range, const bool inside, const int pxi, const int pyi,
                                          const float k0, const float k1, const float k2, const float kd, const float ka, const float Tfin,
                                          const uint32_t lm) {
  constexpr int LEAD = FwGeom<LPP>::LEAD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & (LPP - 1);
  {
    FwLateArgs& l = *fw_late_args();
    if (inside && i == LEAD) {
      const size_t pid = (size_t)pyi * l.W + pxi, HW = (size_t)l.H * l.W;
      l.final_T[pid] = Tfin;
      l.n_contrib[pid] = lm;
      l.final_acc[pid] = make_float4(k0, k1, k2, kd);
      l.out_color[pid] = k0 + Tfin * l.bg[0];
      l.out_color[HW + pid] = k1 + Tfin * l.bg[1];
      l.out_color[2 * HW + pid] = k2 + Tfin * l.bg[2];
      l.out_depth[pid] = kd;
      l.out_alpha[pid] = ka;
    }
  }
  uint32_t m = (inside && i >= LEAD) ? lm : 0u;
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
    uint32_t n_c = 0u, wbase = 0u, limit = 0u;
    if (lane == 0) {
      const uint32_t mb = max(max(fw_wmax[0], fw_wmax[1]), max(fw_wmax[2], fw_wmax[3]));
      const uint32_t before = __hip_atomic_fetch_max(&l.tile_max[tile], mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t ticket = __hip_atomic_fetch_add(&l.tile_ticket[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket == (uint32_t)LPP - 1u) {  // LPP blocks of 256 / LPP pixels per tile
        limit = max(max(before, mb), __hip_atomic_load(&l.tile_max[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        limit = min((uint32_t)total, limit);
        // (for this view's next frame's work list: how deep this tile's walk went — into the slot the tile sort chose)
        l.walk_hist[RIGGS_HIST_HDR + (size_t)l.walk_hist[1] * l.hist_slot_words + tile] = limit;
        n_c = (limit + 63u) >> 6;
        if (n_c) wbase = __hip_atomic_fetch_add(l.work_ctr, 4u * n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 2;  // (quarter-chunks)
      }
    }
    n_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_c);
    wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
    limit = (uint32_t)__builtin_amdgcn_readfirstlane((int)limit);
    for (uint32_t k = lane; k < n_c; k += 64) l.work[wbase + k] = make_uint4((uint32_t)tile, k, range.x, limit);
  }
}

// one work item: a block of 256 / LPP pixels of one tile (LPP = 8: 8 x 4, one pixel row per wave; LPP = 32: 4 x 2, two
// neighbouring pixels per wave), its list walked front to back
template <int LPP, bool TRACE>
__device__ __forceinline__ void fw_block(const RenderArgs& a, const int tile, const int sub, const int index) {
  constexpr int B = FW_B, LEAD = FwGeom<LPP>::LEAD;
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & (LPP - 1);
  int prow, pcol, brow, bcol;  // the pixel and the block's first pixel inside the tile
  if constexpr (LPP == 8) {
    brow = (sub >> 1) * 4; bcol = (sub & 1) * 8;
    prow = brow + wave; pcol = bcol + (lane >> 3);
  } else {
    brow = (sub >> 2) * 2; bcol = (sub & 3) * 4;
    const int p = wave * 2 + (lane >> 5);
    prow = brow + (p >> 2); pcol = bcol + (p & 3);
  }
  const int pxi = (tile % gx) * RIGGS_TILE + pcol, pyi = (tile / gx) * RIGGS_TILE + prow;
  const bool inside = pxi < a.W && pyi < a.H;
  const int pix = prow * 16 + pcol;  // pixel index inside the tile (checkpoint layout)
  const float pfx = (float)pxi, pfy = (float)pyi;
  const uint2 range = a.ranges[tile];
  const int total = (int)(range.y - range.x);
  const uint32_t slot0 = a.slot_base[tile];
  const float bx0 = (float)((tile % gx) * RIGGS_TILE + bcol), by0 = (float)((tile / gx) * RIGGS_TILE + brow);
  FwWalk w;
  w.done = !inside; w.T = 1.0f; w.Tstop = -1.0f; w.last = 0u;
  w.C0 = 0.f; w.C1 = 0.f; w.C2 = 0.f; w.D = 0.f; w.A = 0.f;
  const unsigned long long t_begin = (TRACE && a.trace) ? wall_clock64() : 0ull;
  uint32_t st_rounds = 0, st_surv = 0, st_iters = 0, st_full = 0;
  // checkpoints are held one round (lane LEAD + k of a pixel's group keeps chunk k's) and stored ahead of the next round's loads
  float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f;
  bool hv = false;
  int hbase = 0;
  auto flush_ckpt = [&]() {
    if (hv) {
      float* ck = a.ckpt + ((size_t)(slot0 + (hbase >> 6) + (i - LEAD)) * 5) * 256 + pix;
      ck[0] = h0; ck[256] = h1; ck[512] = h2; ck[768] = h3; ck[1024] = h4;
    }
    hv = false;
  };
  // prefetch registers for the next round (one instance per thread) and the list entry of the round after it.  The loads are
  // unconditional, with the index clamped into the list (a lane past the end re-reads the last instance; fw_stage_round drops
  // it): a conditional load of a float4 makes the compiler split the vector and copy a component out right behind the load —
  // an s_waitcnt in the middle of the gathers, one exposed memory latency per round.
  const uint32_t last_entry = range.y - 1u;  // (total >= 1: only non-empty tiles are work items)
  float4 n_xy, n_co, n_cc;
  uint32_t n_id = a.point_list[min(range.x + (uint32_t)(B + tid), last_entry)];
  {
    const uint32_t id = a.point_list[min(range.x + (uint32_t)tid, last_entry)];
    n_xy = a.xyd[id]; n_co = a.conic_o[id]; n_cc = a.rgb[id];
  }
  unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, tc = TRACE ? clock64() : 0ull;  // (TRACE) shader clocks per phase of the rounds
  auto lap = [&](unsigned long long& acc) { if constexpr (TRACE) { const unsigned long long t = clock64(); acc += t - tc; tc = t; } };
  for (int base = 0; base < total; base += B) {
    if (__syncthreads_count(w.done) == 256) break;
    lap(ph0);
    {
      const int cnt = fw_stage_round<LPP>(tid, base + tid < total, n_xy, n_co, n_cc, bx0, by0);
      if constexpr (TRACE) st_surv += (uint32_t)cnt;
    }
    if constexpr (TRACE) st_rounds++;
    flush_ckpt();
    hbase = base;
    FW_LDS_BARRIER();
    lap(ph1);
    {
      const uint32_t id = n_id;
      n_xy = a.xyd[id]; n_co = a.conic_o[id]; n_cc = a.rgb[id];
      n_id = a.point_list[min(range.x + (uint32_t)(base + 2 * B + tid), last_entry)];
    }
    lap(ph2);
#pragma unroll 1
    for (int k = 0; k < B / 64; k++) {
      const int cbase = base + 64 * k;
      if (cbase >= total) break;
      if (__builtin_amdgcn_ballot_w64(!w.done) == 0) break;
      {
        // checkpoint of the state BEFORE instance cbase: fold the lanes' partial sums
        float k0, k1, k2, kd;
        grp_sum4<LPP>(w.C0, w.C1, w.C2, w.D, k0, k1, k2, kd);
        if (i == LEAD + k) { h0 = w.T; h1 = k0; h2 = k1; h3 = k2; h4 = kd; hv = !w.done; }
      }
      fw_composite_chunk<LPP, TRACE>(w, k, base, lane, pfx, pfy, st_iters, st_full);
    }
    lap(ph3);
  }
  if (TRACE && a.trace && lane == 0 && (uint64_t)index < fw_late_args()->trace_items) {
    unsigned long long* tr = a.trace + ((size_t)index * 4 + wave) * 8;
    tr[6] = t_begin;
    tr[0] = wall_clock64() - t_begin; tr[1] = st_rounds;
    tr[2] = (unsigned long long)st_surv | ((unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xFFFFu) << 32) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 0xFu) << 48);
    tr[3] = st_iters; tr[4] = st_full;
    tr[5] = (unsigned long long)total | ((unsigned long long)tile << 32) | (LPP == 32 ? 1ull << 63 : 0ull);
    tr[7] = min(ph0 >> 6, 0xFFFFull) | (min(ph1 >> 6, 0xFFFFull) << 16) | (min(ph2 >> 6, 0xFFFFull) << 32) | (min(ph3 >> 6, 0xFFFFull) << 48);
  }
  flush_ckpt();
  // fold the lanes (the lanes i >= LEAD of a pixel end up with the same values)
  const float k0 = grp_sum<LPP>(w.C0), k1 = grp_sum<LPP>(w.C1), k2 = grp_sum<LPP>(w.C2), kd = grp_sum<LPP>(w.D), ka = grp_sum<LPP>(w.A);
  const float ts = grp_max<LPP>(w.Tstop);
  const uint32_t lm = grp_max_u<LPP>(w.last);
  const float Tfin = (ts >= 0.f) ? ts : w.T;
  fw_finish<LPP>(tile, total, range, inside, pxi, pyi, k0, k1, k2, kd, ka, Tfin, lm);
}

// ---- the wide block, round 6: the round's survivors composited as ONE list, checkpoints out of the scans.
// What a wide block's time was made of (profiles/round5_fwd_trace_dense.txt: 39 rounds of 256 entries = 96 us, ~5 500 clocks per
// round): ~7 of a round's 256 entries survive the cull against the 4 x 2 pixel box, but the round paid FOUR trips through the
// chunk loop — a checkpoint fold of 20 DPP instructions and a scan step for one or two survivors in 32 lanes each: ~860 clocks a
// trip, 3 400 of the round's 5 500, a lone wave per SIMD waiting out every latency.  Here the survivors of the round's four
// chunks are walked as one list in list order (lane j of a step finds its record in the chunk's compacted region through the
// chunks' counts: no second compaction, no further barrier), a scan step takes 32 of them whatever chunks they come from, and
// the checkpoint of chunk k — the state in front of the chunk's first instance — is the EXCLUSIVE state of the first survivor at
// or behind the chunk's start: the step computes exclusive prefixes anyway (the transmittance product as before, the colour /
// depth sums as four more DPP scans on top of running totals that are kept folded), and lane LEAD + k of the pixel's group
// fetches its chunk's values from that survivor's lane (ds_bpermute); a chunk without a later survivor takes the state behind
// the round.  Same arithmetic per (pixel, instance) as the 8-lane form, sums folded in another order (as the wide form always
// did).  (Built first with rounds of 1024 entries — four per thread, only position + y extent gathered for the cull, the
// survivors' records fetched behind it: 38 spilled registers under this kernel's 80, 0.482 -> 0.520 ms on the opaque-skin scene;
// a kernel of its own would serialise behind the 8-lane blocks it overlaps with today.)
// four inclusive sum scans over the 32 lanes of a pixel's group at once (two rows of 16; the chains fill each other's DPP slots)
__device__ __forceinline__ void fww_scan4(float& a, float& b, float& c, float& d) {
#define FWW_S4(CTRL) \
  "v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\tv_add_f32_dpp %2, %2, %2 " CTRL "\n\tv_add_f32_dpp %3, %3, %3 " CTRL "\n\t"
  asm volatile(
      "s_nop 1\n\t"
      FWW_S4("row_shr:1 row_mask:0xf bank_mask:0xf")
      FWW_S4("row_shr:2 row_mask:0xf bank_mask:0xf")
      FWW_S4("row_shr:4 row_mask:0xf bank_mask:0xf")
      FWW_S4("row_shr:8 row_mask:0xf bank_mask:0xf")
      FWW_S4("row_bcast:15 row_mask:0xa bank_mask:0xf")
      "s_nop 0"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef FWW_S4
}
__device__ __forceinline__ float fww_lane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float fww_fetch(float v, int byte_addr) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}

template <bool TRACE>
__device__ __forceinline__ void fw_block_wide(const RenderArgs& a, const int tile, const int sub, const int index) {
  constexpr int LPP = 32, LEAD = FwGeom<32>::LEAD, B = FW_B;
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & (LPP - 1), half = lane & 32;
  const int brow = (sub >> 2) * 2, bcol = (sub & 3) * 4;
  const int pp = wave * 2 + (lane >> 5);
  const int prow = brow + (pp >> 2), pcol = bcol + (pp & 3);
  const int pxi = (tile % gx) * RIGGS_TILE + pcol, pyi = (tile / gx) * RIGGS_TILE + prow;
  const bool inside = pxi < a.W && pyi < a.H;
  const int pix = prow * 16 + pcol;
  const float pfx = (float)pxi, pfy = (float)pyi;
  const uint2 range = a.ranges[tile];
  const int total = (int)(range.y - range.x);
  const uint32_t slot0 = a.slot_base[tile];
  const float bx0 = (float)((tile % gx) * RIGGS_TILE + bcol), by0 = (float)((tile / gx) * RIGGS_TILE + brow);
  // the pixel's state: transmittance and the FOLDED sums (the same in every lane of the group); alpha and the stop
  // bookkeeping as in the other form
  float T = 1.0f, R0 = 0.f, R1 = 0.f, R2 = 0.f, RD = 0.f, accA = 0.f, Tstop = -1.0f;
  uint32_t last = 0u;
  bool done = !inside;
  const unsigned long long t_begin = (TRACE && a.trace) ? wall_clock64() : 0ull;
  uint32_t st_rounds = 0, st_surv = 0, st_iters = 0, st_full = 0;
  // checkpoints of the round's four chunks: lane LEAD + k of the group holds chunk k's until the next round's flush
  float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f;
  bool hv = false, hres = true;
  int hbase = 0;
  auto flush_ckpt = [&]() {
    if (hv) {
      float* ck = a.ckpt + ((size_t)(slot0 + (hbase >> 6) + (i - LEAD)) * 5) * 256 + pix;
      ck[0] = h0; ck[256] = h1; ck[512] = h2; ck[768] = h3; ck[1024] = h4;
    }
    hv = false;
  };
  const uint32_t last_entry = range.y - 1u;
  float4 n_xy, n_co, n_cc;
  uint32_t n_id = a.point_list[min(range.x + (uint32_t)(B + tid), last_entry)];
  {
    const uint32_t id = a.point_list[min(range.x + (uint32_t)tid, last_entry)];
    n_xy = a.xyd[id]; n_co = a.conic_o[id]; n_cc = a.rgb[id];
  }
  unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, tc = TRACE ? clock64() : 0ull;
  auto lap = [&](unsigned long long& acc) { if constexpr (TRACE) { const unsigned long long t = clock64(); acc += t - tc; tc = t; } };
  for (int base = 0; base < total; base += B) {
    if (__syncthreads_count(done) == 256) break;
    lap(ph0);
    {
      const int cnt = fw_stage_round<LPP>(tid, base + tid < total, n_xy, n_co, n_cc, bx0, by0);
      if constexpr (TRACE) st_surv += (uint32_t)cnt;
    }
    if constexpr (TRACE) st_rounds++;
    flush_ckpt();
    hbase = base;
    hres = !(i >= LEAD && i < LEAD + B / 64 && base + 64 * (i - LEAD) < total && inside);  // (this lane's chunk exists: its checkpoint is to be resolved)
    FW_LDS_BARRIER();
    lap(ph1);
    {
      const uint32_t id = n_id;
      n_xy = a.xyd[id]; n_co = a.conic_o[id]; n_cc = a.rgb[id];
      n_id = a.point_list[min(range.x + (uint32_t)(base + 2 * B + tid), last_entry)];
    }
    lap(ph2);
    // the round's survivors as one list: chunk k's sit at 64 k .. 64 k + cnt_k of the staged records
    const int p1 = fw_cnt[0], p2 = p1 + fw_cnt[1], p3 = p2 + fw_cnt[2], n_surv = p3 + fw_cnt[3];
    const int my_idx = (i == LEAD + 1) ? p1 : ((i == LEAD + 2) ? p2 : ((i == LEAD + 3) ? p3 : 0));  // survivors in front of this lane's chunk
    if (__builtin_amdgcn_ballot_w64(!done || !hres) != 0) {
      for (int g = 0; g < n_surv; g += LPP) {
        const int j0 = g + i;  // this lane's survivor
        const int ch = (j0 >= p1 ? 1 : 0) + (j0 >= p2 ? 1 : 0) + (j0 >= p3 ? 1 : 0);
        const int rec = 64 * ch + (j0 - (ch == 0 ? 0 : (ch == 1 ? p1 : (ch == 2 ? p2 : p3))));
        const bool live = j0 < n_surv;
        const int ri = live ? rec : 0;
        const float4 xy = fw_xyd[ri];
        const float4 c = fw_rgb[ri];
        const float dx = xy.x - pfx, dy = xy.y - pfy;
        const float4 co = fw_con[ri];
        const float pw = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
        const float alpha = fminf(ALPHA_MAX, co.w * fast_exp(pw));
        const bool valid = live && (pw <= 0.0f) && (alpha >= ALPHA_MIN) && !done;
        if constexpr (TRACE) st_iters++;
        const int pos1 = base + (int)fw_pos[ri] + 1;
        const float om = valid ? 1.0f - alpha : 1.0f;
        float inc = om, E = 1.0f;
        asm volatile(
            "s_nop 1\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
            "v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0"
            : "+v"(inc), "+v"(E));
        E = (i == 0) ? 1.0f : E;
        const float prod = half ? fww_lane(inc, 63) : fww_lane(inc, 31);
        const float Tj = T * E;
        const float test_T = Tj * om;
        const bool sc = valid && (test_T < T_EPS);
        const uint64_t bits = __builtin_amdgcn_ballot_w64(sc);
        const uint32_t ob = (uint32_t)(bits >> half);
        const bool first_stop_before = (ob & ((1u << i) - 1u)) != 0u;
        const bool use = valid && !sc && !first_stop_before;
        const float wt = use ? alpha * Tj : 0.f;
        if constexpr (TRACE) st_full += (__builtin_amdgcn_ballot_w64(use) != 0) ? 1u : 0u;
        // the sums in front of every lane: inclusive scans of the lanes' contributions on top of the folded totals
        float s0 = c.x * wt, s1 = c.y * wt, s2 = c.z * wt, sd = xy.z * wt;
        const float q0 = s0, q1 = s1, q2 = s2, qd = sd;
        fww_scan4(s0, s1, s2, sd);
        accA += wt;
        last = use ? (uint32_t)pos1 : last;
        if (sc && !first_stop_before) Tstop = Tj;
        // checkpoints whose first survivor sits in this step: lane LEAD + k fetches that survivor's exclusive state
        if (__builtin_amdgcn_ballot_w64(!hres) != 0) {
          const int j = my_idx - g;
          const bool mine = !hres && j >= 0 && j < LPP && my_idx < n_surv;
          const int src = (half + (mine ? j : 0)) << 2;
          const float f0 = fww_fetch(Tj, src), f1 = fww_fetch(R0 + (s0 - q0), src), f2 = fww_fetch(R1 + (s1 - q1), src),
                      f3 = fww_fetch(R2 + (s2 - q2), src), f4 = fww_fetch(RD + (sd - qd), src);
          if (mine) {
            h0 = f0; h1 = f1; h2 = f2; h3 = f3; h4 = f4;
            hv = !done && (ob & ((1u << j) - 1u)) == 0u;  // (the pixel has not stopped in front of that survivor)
            hres = true;
          }
        }
        const bool nostop = (ob == 0u);
        if (!done) {
          R0 += half ? fww_lane(s0, 63) : fww_lane(s0, 31);
          R1 += half ? fww_lane(s1, 63) : fww_lane(s1, 31);
          R2 += half ? fww_lane(s2, 63) : fww_lane(s2, 31);
          RD += half ? fww_lane(sd, 63) : fww_lane(sd, 31);
          T = nostop ? T * prod : T;
        }
        done = done || !nostop;
      }
    }
    // chunks without a survivor at or behind their start: the state behind the round
    if (!hres) { h0 = T; h1 = R0; h2 = R1; h3 = R2; h4 = RD; hv = !done; hres = true; }
    lap(ph3);
  }
  if (TRACE && a.trace && lane == 0 && (uint64_t)index < fw_late_args()->trace_items) {
    unsigned long long* tr = a.trace + ((size_t)index * 4 + wave) * 8;
    tr[6] = t_begin;
    tr[0] = wall_clock64() - t_begin; tr[1] = st_rounds;
    tr[2] = (unsigned long long)st_surv | ((unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xFFFFu) << 32) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 0xFu) << 48);
    tr[3] = st_iters; tr[4] = st_full;
    tr[5] = (unsigned long long)total | ((unsigned long long)tile << 32) | (1ull << 63);
    tr[7] = min(ph0 >> 6, 0xFFFFull) | (min(ph1 >> 6, 0xFFFFull) << 16) | (min(ph2 >> 6, 0xFFFFull) << 32) | (min(ph3 >> 6, 0xFFFFull) << 48);
  }
  flush_ckpt();
  const float ka = grp_sum<LPP>(accA);
  const float ts = grp_max<LPP>(Tstop);
  const uint32_t lm = grp_max_u<LPP>(last);
  const float Tfin = (ts >= 0.f) ? ts : T;
  fw_finish<LPP>(tile, total, range, inside, pxi, pyi, R0, R1, R2, RD, ka, Tfin, lm);
}

template <bool TRACE>
__global__ __launch_bounds__(256, 6) void render_fwd_oct_kernel(RenderArgs a) {
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE;
  const int tid = threadIdx.x;
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
  // work list: the non-empty tiles in the order the extra workgroup of bin_scatter_kernel wrote them (longest lists first);
  // the first item_ctr[1] entries are WIDE (32 blocks of 4 x 2 pixels each), the others 8 blocks of 8 x 4.  One block per
  // workgroup (the launch covers the most the list can hold: an outer loop over items makes every item-invariant scalar a
  // loop invariant that the compiler computes up front and keeps, in vector-register lanes once the scalar file is full).
  if (!a.items) return;
  const int n_entries = (int)a.item_ctr[0], n_wide = (int)a.item_ctr[1];
  const int it = (int)blockIdx.x, wide_items = n_wide * 32;
  // the blocks of an entry get workgroup ids 8 apart = the same XCD / L2 (workgroup b runs on XCD b % 8)
  if (it < wide_items) {
    const int full = (n_wide >> 3) << 8;
    int p, sub;
    if (it < full) { p = ((it >> 8) << 3) + (it & 7); sub = (it >> 3) & 31; }
    else { p = (full >> 5) + ((it - full) >> 5); sub = (it - full) & 31; }
    fw_block_wide<TRACE>(a, (int)a.items[p], sub, it);
  } else {
    const int j = it - wide_items, n_items = (n_entries - n_wide) * 8, full = (n_items >> 6) << 6;
    if (j >= n_items) return;
    int p, sub;
    if (j < full) { p = ((j >> 6) << 3) + (j & 7); sub = (j >> 3) & 7; }
    else { p = (full >> 3) + ((j - full) >> 3); sub = (j - full) & 7; }
    fw_block<8, TRACE>(a, (int)a.items[n_wide + p], sub, it);
  }
}

// the tiles that may be composited wide per launch (default 256; 0 turns the wide form off), and the walk depth / list length
// from which a tile is (default 4096): riggs_set_option("fwd_wide_tiles" / "fwd_wide_min"), for tools and tests
uint32_t forward_wide_tiles() { return (uint32_t)option(OPT_FWD_WIDE_TILES); }
uint32_t forward_wide_min() { return (uint32_t)option(OPT_FWD_WIDE_MIN); }

int launch_render_fwd(const RenderArgs& a, hipStream_t s) {
  const int gx = (a.W + RIGGS_TILE - 1) / RIGGS_TILE, gy = (a.H + RIGGS_TILE - 1) / RIGGS_TILE;
  if (gx * gy == 0) return 0;
  // one workgroup per block: eight per tile, 24 more for every tile that may be wide; the ones past the end of the list only
  // help with the background of the empty tiles and leave (tile_max, the tile tickets and the work-list size were cleared by
  // the extra workgroup of bin_scatter_kernel)
  const int64_t T = (int64_t)gx * gy;
  const int64_t wide = a.items ? (T < (int64_t)forward_wide_tiles() ? T : (int64_t)forward_wide_tiles()) : 0;
  const int64_t blocks = T * 8 + wide * 24;
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
__global__ __launch_bounds__(64 * NW) void render_bwd_kernel(RenderBwdArgs a) {
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
  };
  auto issue_level3 = [&](const u4v wk, uint32_t id, uint32_t n, Level3& r) {
    const int tile = (int)wk.x, pos0 = (int)wk.y * 64;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    r.xy = z; r.co = z; r.cc = z; r.acc = z;
    r.Tn = 0.f; r.g0 = 0.f; r.g1 = 0.f; r.g2 = 0.f; r.gD = 0.f; r.gA = 0.f; r.Ts = 1.f; r.S0 = 0.f; r.S1 = 0.f; r.S2 = 0.f; r.Ds = 0.f;
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
    }
  };
  auto stage_pixels = [&](const u4v wk, uint32_t n, const Level3& r) {  // this wave's pixels (one per lane) -> LDS
    const int pos0 = (int)wk.y * 64;
    float4 pa = make_float4(1.f, 0.f, 0.f, 0.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
    float pc = 0.f;
    if ((int)n > pos0) {
      const float Ts = r.Ts;
      const float pre = r.g0 * r.S0 + r.g1 * r.S1 + r.g2 * r.S2 + r.gD * r.Ds + r.gA * (1.0f - Ts);
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
