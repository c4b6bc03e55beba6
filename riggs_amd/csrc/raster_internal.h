// Internal argument blocks and launchers shared by the rasterizer translation units.
#pragma once
#include "common.h"

namespace riggs {

// SH colour evaluation as a job of Gaussian blocks (color_job.h) that the tile sort's scatter launch hosts.  The record lives
// in the geometry arena: preprocess_fwd writes it (N = 0: it evaluated the colours itself), the hosting launch reads it.
struct ColorJob {
  int N, deg, M, pad_;
  const float *shs, *shs_rest, *means3D, *d_xyz /* or NULL */, *campos;
  const int32_t* radii;  // (0: culled — no colour)
  float4* rgb;           // .xyz written by the job; .w (a culling extent) is preprocess_fwd's
  uint8_t* clamped;
};
// Is the colour job of a frame with these sizes hosted by the tile sort (launch_binning)?  Asked by riggs_raster_preprocess
// (which then leaves the colours out of preprocess_fwd) and by riggs_raster_render (which then launches the extra workgroups):
// both must see the same options ("color_side_jobs", "bin_grouped").
bool binning_hosts_color(int N, int T, int sh_coeffs);

struct PreArgs {
  int N, deg, M, W, H, glue, isotropic, tight;
  int defer_color;      // forward only: the SH colours are left to the tile sort's scatter launch (color_job.h)
  ColorJob* job_rec;    // forward only: the job's record in the geometry arena (written by this launch), or NULL
  float tanx, tany, mod;
  const float *view, *proj, *campos;
  const float *means3D, *shs, *shs_rest, *colors_precomp, *opac, *scales, *rots, *cov3D_precomp, *d_xyz, *d_rot, *d_scaling;
  int32_t* radii;
  float4 *xyd, *conic_o, *rgb;
  float* cov3D;
  uint8_t* clamped;
  uint32_t* tiles;
  ushort4* rect;
  uint32_t* depth_key;
  uint32_t* block_tiles;  // per workgroup: [0, n): sums of tiles_touched; [n, 2n): (min << 8) | max of the visible keys' top bytes
};

struct PreBwdArgs {
  PreArgs f;                     // forward inputs (radii/xyd/... unused here except radii, cov3D, clamped)
  const uint32_t* counters;      // the forward's {R, overflow flag}: an overflowed frame back-propagates exact zeros
  float* gacc;                   // workspace: per Gaussian [mean2D.x, mean2D.y, gA, gB, gC, g_opacity, g_r, g_g, g_b, g_depth, -, -];
                                 // zero on entry of the compositing backward, zero again when preprocess_bwd returns
  float *dL_dmeans3D, *dL_dmeans2D, *dL_dsh, *dL_dcolors, *dL_dopac, *dL_dscales, *dL_drots, *dL_dcov3D, *dL_dd_scaling, *dL_dsh_rest;
  // workspace, behind the accumulators: which Gaussians received a gradient this frame (one bit each, a 64-bit word per
  // wave) and how many per block of 256 — rewritten by every launch; read by the gradient-row exchange (exchange.hip)
  unsigned long long* touched_bits;
  uint32_t* block_touched;
  int sparse_zero;               // cfg.sparse_zero: rows without a gradient now AND in the previous call are not rewritten
};
#define RIGGS_GACC 12  // floats per Gaussian in the render-backward accumulator (padded to 48 B)
// backward workspace: [accumulators N x 48 B | touched bits, one 64-bit word per 64 Gaussians | touched count per 256]
static inline size_t ws_bits_offset(int32_t N) { return align_up((size_t)(N > 0 ? N : 1) * RIGGS_GACC * 4); }
static inline size_t ws_blocks_offset(int32_t N) { return ws_bits_offset(N) + align_up((size_t)((N > 0 ? N : 1) + 63) / 64 * 8); }


int launch_preprocess_fwd(const PreArgs& a, hipStream_t s);
int launch_preprocess_bwd(const PreBwdArgs& b, hipStream_t s);

struct RenderArgs {
  int W, H;
  unsigned long long* trace;  // optional per-workgroup statistics of render_fwd (riggs_raster_set_trace), else NULL
  const uint2* ranges;
  const uint32_t* point_list;
  const float4 *xyd, *conic_o, *rgb;
  const float* bg;
  float* final_T;
  uint32_t* n_contrib;
  float *out_color, *out_depth, *out_alpha;
  // state for the chunk-parallel backward
  float4* final_acc;          // per pixel (C0, C1, C2, D) accumulated WITHOUT background
  uint32_t* tile_max;         // per tile: max n_contrib
  uint32_t* tile_ticket;      // per tile: forward blocks that have finished it (the last one appends the tile's backward work)
  uint4* work;                // backward work list: (tile, chunk, start of the tile's list, instances to walk) per active chunk
  uint32_t* work_ctr;         // its size in quarter-chunks (zeroed by the extra workgroup of bin_scatter_kernel)
  const uint32_t* slot_base;  // per tile: first checkpoint slot ((range.x >> 6) + tile)
  float* ckpt;                // [slot][5][256]: (T, C0, C1, C2, D) per pixel at every 64th instance
  // work list (NULL: nothing was binned, every tile is empty): items = the non-empty tiles, longest lists first — the
  // first item_ctr[1] of them are composited WIDE (render.hip) —; empties = the tiles without instances;
  // item_ctr = {n_nonempty, n_wide, n_empty}
  const uint32_t* items;
  const uint32_t* empties;
  uint32_t* item_ctr;
  uint32_t* walk_hist;        // the walk histories (BinOut::walk_hist): the forward writes max n_contrib per tile into the slot of THIS view
  uint32_t hist_slot_words;   // (riggs_hist_slot_words(T))
  uint64_t trace_items;       // capacity of the trace buffer in work items (tools; 0: 8 per tile)
};
int launch_render_fwd(const RenderArgs& a, hipStream_t s);
uint32_t forward_wide_tiles();  // tiles that may be composited wide per launch (bounds the grid)
uint32_t forward_wide_min();    // list length from which a tile is composited wide

struct RenderBwdArgs {
  unsigned long long* trace;  // optional per-chunk statistics (riggs_raster_set_trace), else NULL
  int W, H;
  const uint2* ranges;
  const uint32_t* point_list;
  const float4 *xyd, *conic_o, *rgb;
  const float* bg;
  const float* final_T;
  const uint32_t* n_contrib;
  const float *dL_dcolor, *dL_ddepth, *dL_dalpha;
  float* gacc;  // (N, RIGGS_GACC) accumulators: all zero on entry (see PreBwdArgs::gacc)
  float* det_rows;  // ordered-reduction mode: one row of 10 floats per tile instance instead of the atomics (else NULL)
  int n_points;
  const float4* final_acc;
  const uint32_t* tile_max;
  const uint32_t* slot_base;
  const float* ckpt;
  int n_tiles;
  int64_t n_slots;
  const uint4* work;  // per active chunk: (tile, chunk, start of the tile's list, instances to walk); built by the forward
  const uint32_t* work_ctr;  // number of quarter-items
};
int launch_render_bwd(const RenderBwdArgs& a, hipStream_t s);
// ordered-reduction mode: sum every Gaussian's instance rows in ascending tile order into the accumulators
int launch_ordered_gather(int N, int n_tiles, int grid_x, int64_t cap, const uint2* ranges, const uint32_t* point_list,
                          const uint32_t* tiles, const ushort4* rect, const float* det_rows, uint32_t* inv, uint32_t* off,
                          float* gacc, int want_depth, hipStream_t s);

// binning
size_t depth_sort_table_bytes(int N);
int launch_depth_sort(int N, const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, void* table_mem,
                      const uint32_t* block_info, uint32_t* counters, hipStream_t s);
size_t bin_table_bytes(int N, int T, int grid_x);
size_t bin_scratch_bytes(int64_t cap, int T, int grid_x);  // scratch of the grouped binning (<= the checkpoint area it is given)
// The forward's walk histories: how deep every tile's list was walked, PER VIEW.  A trainer draws another camera every iteration
// out of a fixed set (train_rig.py:389), so "the previous frame" is a frame seen from elsewhere — but the same views come back.
// The binning arena keeps RIGGS_HIST_SLOTS histories, each with the view matrix it belongs to; the tile sort's extra workgroup
// takes the slot whose view matches this frame's (riggs_set_option("fwd_hist_view_tol")) or recycles one round robin (no history
// for this frame then), and tells the forward which one to write (header word 1).
//   words: [0] round-robin cursor, [1] this frame's slot, [2 .. 15] spare; then per slot: [T] depths (bit 31: composited wide in
//   this frame), [T] the stamp, [T + 1] spare, [T + 2 .. T + 17] the view matrix — riggs_hist_slot_words(T) words each
#define RIGGS_HIST_SLOTS 128
#define RIGGS_HIST_HDR 16
__host__ __device__ static inline uint32_t riggs_hist_slot_words(uint32_t T) { return (T + 18u + 15u) & ~15u; }
static inline size_t riggs_hist_words(uint32_t T) { return RIGGS_HIST_HDR + (size_t)RIGGS_HIST_SLOTS * riggs_hist_slot_words(T); }

struct BinOut {  // what the extra workgroup of bin_scatter_kernel writes once per frame
  uint2* ranges;
  uint32_t *slot_base, *tile_max, *counters, *fwd_items, *fwd_empty, *fwd_ctr;
  uint32_t* walk_hist;   // the histories (layout above); a slot's [t]: how deep the forward's walk went in that view's last frame; read, then
                         // bit 31 = composited wide in this frame; [T] = a stamp that says the words are a history
  uint32_t hist_stamp;   // the value of walk_hist[T] that marks a history of THIS scene size and tile grid
  float view_tol;           // ... a history counts when no entry of the view matrix moved by more than this since its frame
  const float* viewmatrix;  // this frame's view matrix (16 device floats) or NULL (one history, whatever the view: slot 0)
  uint32_t wide_tiles;   // at most this many tiles are composited wide by the forward (0: none) ...
  uint32_t wide_min;     // ... the ones whose walk was, and whose list is, this many instances deep (forward_wide_tiles / forward_wide_min)
};
int launch_binning(int N, int T, int grid_x, int64_t cap, const uint32_t* order, const uint32_t* tiles,
                   const ushort4* rect, void* table_mem, void* scratch, uint32_t* point_list, uint32_t* tile_keys, const BinOut& out,
                   hipStream_t s, const ColorJob* job_rec = nullptr /* device record, when the launch hosts the colour job */,
                   int job_sh_coeffs = 0);

}  // namespace riggs
