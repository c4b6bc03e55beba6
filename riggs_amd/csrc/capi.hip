// extern "C" boundary of libriggs_hip.so (see include/riggs_hip.h) — rasterizer part.
#include <stdarg.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "raster_internal.h"

namespace riggs {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- library options -------------------------------------------------------------------
static const char* kOptNames[OPT_COUNT] = {"fwd_wide_tiles", "fwd_wide_min", "bin_grouped", "cnode_bwd_atomics", "color_side_jobs", "preprocess_bwd_lean", "pose_mlp_layered", "fwd_hist_view_tol", "lbs_scalar"};
static const int kOptDefaults[OPT_COUNT] = {256, 4096, -1, 0, 1, -1, 0, 20, 0};
static int g_opt[OPT_COUNT] = {256, 4096, -1, 0, 1, -1, 0, 20, 0};
int option(int id) { return g_opt[id]; }

// ---- event-based kernel timing -------------------------------------------------------
static const char* kProfNames[PROF_COUNT] = {
    "preprocess_fwd", "depth_sort", "tile_sort", "render_fwd", "render_bwd",
    "preprocess_bwd", "fk_fwd", "lbs_fwd", "lbs_bwd", "fk_bwd", "knn", "pose_mlp_fwd", "pose_mlp_bwd", "adam", "loss_fwd", "loss_bwd"};
struct ProfSlot {
  std::vector<hipEvent_t> start, stop;
  size_t used = 0;
};
static uint32_t g_prof_mask = 0;
static ProfSlot g_prof[PROF_COUNT];
void prof_begin(int id, hipStream_t s) {
  if (!(g_prof_mask >> id & 1u)) return;
  ProfSlot& p = g_prof[id];
  if (p.used == p.start.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    p.start.push_back(a); p.stop.push_back(b);
  }
  (void)hipEventRecord(p.start[p.used], s);
}
void prof_end(int id, hipStream_t s) {
  if (!(g_prof_mask >> id & 1u)) return;
  ProfSlot& p = g_prof[id];
  if (p.used < p.start.size()) { (void)hipEventRecord(p.stop[p.used], s); p.used++; }
}

GeomLayout geom_layout(int N) {
  GeomLayout L;
  size_t n = (size_t)(N > 0 ? N : 1), o = 0;
  L.xyd = o; o = align_up(o + n * 16);
  L.conic_o = o; o = align_up(o + n * 16);
  L.rgb = o; o = align_up(o + n * 16);
  L.cov3D = o; o = align_up(o + n * 24);
  L.clamped = o; o = align_up(o + n);
  L.tiles = o; o = align_up(o + n * 4);
  L.rect = o; o = align_up(o + n * 8);
  L.depth_key = o; o = align_up(o + n * 4);
  L.depth_key_sorted = o; o = align_up(o + n * 4);
  L.order = o; o = align_up(o + n * 4);
  L.block_tiles = o; o = align_up(o + 2 * ((n + 255) / 256) * 4);  // per workgroup of preprocess_fwd: tile sums | depth top bytes
  L.color_job = o; o = align_up(o + sizeof(ColorJob));            // the colour job's record (color_job.h)
  L.sort_table = o; o += align_up(depth_sort_table_bytes(N));
  L.total = o;
  return L;
}

ImageLayout image_layout(int H, int W) {
  ImageLayout L;
  size_t hw = (size_t)H * W, o = 0;
  size_t T = (size_t)((W + RIGGS_TILE - 1) / RIGGS_TILE) * ((H + RIGGS_TILE - 1) / RIGGS_TILE);
  L.final_T = o; o = align_up(o + hw * 4);
  L.n_contrib = o; o = align_up(o + hw * 4);
  L.final_acc = o; o = align_up(o + hw * 16);
  L.ranges = o; o = align_up(o + (T + 1) * 8);   // ranges, tile_max (+ tile tickets) and slot_base are adjacent: one memset clears them
  L.tile_max = o; o = align_up(o + 2 * (T + 1) * 4);  // [tile_max (T + 1) | arrival tickets of the tile's forward blocks (T + 1)]
  L.slot_base = o; o = align_up(o + (T + 2) * 4);
  // the empty tiles (written by the extra workgroup of bin_scatter_kernel, like the forward's work list in the binning arena)
  L.fwd_empty = o; o = align_up(o + (T + 1) * 4);
  // {n_nonempty, -, n_empty} read by every forward workgroup; [64] = size of the backward's work list (in quarter-chunks),
  // appended to by the forward with atomics — on a cache line of its own (see render_fwd_oct_kernel)
  L.fwd_ctr = o; o = align_up(o + 512);
  L.total = o;
  return L;
}

BinLayout bin_layout(int64_t cap, int N, int H, int W) {
  BinLayout L;
  const size_t T = (size_t)((W + RIGGS_TILE - 1) / RIGGS_TILE) * ((H + RIGGS_TILE - 1) / RIGGS_TILE);
  size_t n = (size_t)(cap > 0 ? cap : 1), o = 0;
  L.point_list = o; o = align_up(o + n * 4);  // sorted point list first (RIGGS_BIN_POINT_LIST)
  L.tile_keys = o; o = align_up(o + n * 4);   // per-instance tile id (written with cfg.debug only)
  L.n_slots = (n >> 6) + T + 1;  // tile t, chunk c -> slot (range.x(t) >> 6) + t + c
  L.ckpt = o; o += align_up(L.n_slots * RIGGS_CKPT_FLOATS * 4);
  L.table = o; o += align_up(bin_table_bytes(N, (int)T, (W + RIGGS_TILE - 1) / RIGGS_TILE));
  L.work = o; o += align_up(L.n_slots * 16);  // backward work list: 16-byte entry per active chunk
  L.fwd_items = o; o += align_up((T + 1) * 4);  // forward work list: the non-empty tiles, longest lists first
  // how deep the forward walked every tile's list: read by the NEXT frame's work-list builder (the binning arena is the one that
  // persists from frame to frame: riggs_amd.rasterizer.RasterArena, captured frames)
  L.walk_hist = o; o += align_up(riggs_hist_words((uint32_t)T) * 4);  // (per view: raster_internal.h)
  L.total = o;
  return L;
}

}  // namespace riggs

using namespace riggs;

extern "C" {

// Which frames left their SH colours to the tile sort's scatter launch: riggs_raster_preprocess decides (from process-wide options)
// and riggs_raster_render must launch the workgroups that do it — a decision re-taken there would disagree when an option changed
// between the two calls, and then NOTHING would write the colours.  Remembered per geometry arena (the frame's identity between
// the calls), a few frames deep; a render that finds "deferred" but would not host the job fails instead of compositing garbage.
struct DeferNote { const void* geom; int deferred; };
static DeferNote g_defer[16];
static unsigned g_defer_at = 0;
static void defer_note(const void* geom, int deferred) {
  for (auto& d : g_defer) if (d.geom == geom) { d.deferred = deferred; return; }
  g_defer[g_defer_at++ % 16] = DeferNote{geom, deferred};
}
static int defer_lookup(const void* geom) {  // -1: unknown (the arena of a frame more than 16 preprocess calls ago)
  for (auto& d : g_defer) if (d.geom == geom) return d.deferred;
  return -1;
}

int riggs_version(void) { return 100; }

int riggs_prof_count(void) { return PROF_COUNT; }
const char* riggs_prof_name(int32_t id) { return (id >= 0 && id < PROF_COUNT) ? kProfNames[id] : ""; }
int riggs_prof_enable(uint32_t mask) { g_prof_mask = mask; return 0; }
int riggs_prof_reset(void) {
  for (int i = 0; i < PROF_COUNT; i++) g_prof[i].used = 0;
  return 0;
}
int riggs_prof_read(int32_t id, float* total_ms, int32_t* launches) {
  RIGGS_REQUIRE(id >= 0 && id < PROF_COUNT, "bad stage id");
  ProfSlot& p = g_prof[id];
  float tot = 0.f;
  for (size_t i = 0; i < p.used; i++) {
    RIGGS_HIP_CHECK(hipEventSynchronize(p.stop[i]));
    float ms = 0.f;
    RIGGS_HIP_CHECK(hipEventElapsedTime(&ms, p.start[i], p.stop[i]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = (int32_t)p.used;
  return 0;
}
const char* riggs_last_error(void) { return g_err; }

static int opt_id(const char* name) {
  for (int i = 0; i < OPT_COUNT; i++) if (name && !strcmp(name, kOptNames[i])) return i;
  return -1;
}
int riggs_set_option(const char* name, int32_t value) {
  const int id = opt_id(name);
  if (id < 0) { set_error("riggs_set_option: unknown option '%s'", name ? name : "(null)"); return 2; }
  int v = value;
  if (id == OPT_FWD_WIDE_TILES) { if (v < 0) v = kOptDefaults[id]; if (v > 65535) v = 65535; }
  if (id == OPT_FWD_WIDE_MIN) { if (v < 0) v = kOptDefaults[id]; if (v < 256) v = 256; }
  if (id == OPT_FWD_HIST_VIEW_TOL) { if (v < 0) v = kOptDefaults[id]; }
  if (id == OPT_BIN_GROUPED) { if (v < -1 || v > 1) { set_error("riggs_set_option: bin_grouped takes -1 (by size), 0 or 1"); return 2; } }
  if (id == OPT_PREPROCESS_BWD_LEAN) { if (v < -1 || v > 1) { set_error("riggs_set_option: preprocess_bwd_lean takes -1 (with cfg.sparse_zero), 0 or 1"); return 2; } }
  if (id == OPT_CNODE_BWD_ATOMICS || id == OPT_COLOR_SIDE_JOBS || id == OPT_POSE_MLP_LAYERED) v = v ? 1 : 0;
  if (id == OPT_LBS_SCALAR) { if (v < -1 || v > 1) { set_error("riggs_set_option: lbs_scalar takes -1 (never), 0 (by size) or 1 (always)"); return 2; } }
  g_opt[id] = v;
  return 0;
}
int riggs_get_option(const char* name, int32_t* value) {
  const int id = opt_id(name);
  if (id < 0 || !value) { set_error("riggs_get_option: unknown option '%s'", name ? name : "(null)"); return 2; }
  *value = g_opt[id];
  return 0;
}

size_t riggs_raster_geom_bytes(int32_t N) { return geom_layout(N).total; }
size_t riggs_raster_image_bytes(int32_t H, int32_t W) { return image_layout(H, W).total; }
size_t riggs_raster_binning_bytes(int64_t cap, int32_t N, int32_t H, int32_t W) { return bin_layout(cap, N, H, W).total; }
size_t riggs_raster_backward_workspace_bytes(int32_t N) { return ws_blocks_offset(N) + align_up((size_t)((N > 0 ? N : 1) + 255) / 256 * 4); }
int riggs_raster_backward_workspace_rows(int32_t N, size_t* offset, size_t* bytes) {
  *offset = ws_bits_offset(N);
  *bytes = (size_t)((N > 0 ? N : 1) + 63) / 64 * 8;
  return 0;
}
size_t riggs_raster_backward_workspace_bytes_ordered(int32_t N, int64_t cap) {
  const size_t c = (size_t)(cap > 0 ? cap : 1), n = (size_t)(N > 0 ? N : 1);
  return riggs_raster_backward_workspace_bytes(N) + align_up(c * 40) + align_up(c * 4) + align_up((n + 1) * 4);
}

int riggs_raster_geom_layout(int32_t N, size_t* o) {
  GeomLayout L = geom_layout(N);
  o[RIGGS_GEOM_XYD] = L.xyd; o[RIGGS_GEOM_CONIC_O] = L.conic_o; o[RIGGS_GEOM_RGB] = L.rgb;
  o[RIGGS_GEOM_COV3D] = L.cov3D; o[RIGGS_GEOM_CLAMPED] = L.clamped; o[RIGGS_GEOM_TILES] = L.tiles;
  o[RIGGS_GEOM_RECT] = L.rect; o[RIGGS_GEOM_DEPTH_ORDER] = L.order;
  return 0;
}
int riggs_raster_image_layout(int32_t H, int32_t W, size_t* o) {
  ImageLayout L = image_layout(H, W);
  o[RIGGS_IMG_FINAL_T] = L.final_T; o[RIGGS_IMG_N_CONTRIB] = L.n_contrib; o[RIGGS_IMG_RANGES] = L.ranges;
  o[RIGGS_IMG_FWD_CTR] = L.fwd_ctr;
  return 0;
}
int riggs_raster_binning_layout(int64_t cap, int32_t N, int32_t H, int32_t W, size_t* o) {
  BinLayout L = bin_layout(cap, N, H, W);
  o[RIGGS_BIN_POINT_LIST] = L.point_list; o[RIGGS_BIN_TILE_KEYS] = L.tile_keys; o[RIGGS_BIN_WALK_HIST] = L.walk_hist;
  return 0;
}

static int fill_pre_args(PreArgs& a, const riggs_raster_cfg* c, const float* means3D, const float* shs,
                         const float* shs_rest, const float* colors_precomp, const float* opac, const float* scales, const float* rots,
                         const float* cov3D_precomp, const float* d_xyz, const float* d_rot, const float* d_scaling,
                         char* geom, int32_t* radii) {
  RIGGS_REQUIRE(c != nullptr, "cfg is NULL");
  RIGGS_REQUIRE(c->num_points >= 0 && c->image_height > 0 && c->image_width > 0, "bad sizes");
  if (c->num_points > 0) {  // (empty tensors have NULL data pointers)
    RIGGS_REQUIRE((shs != nullptr) != (colors_precomp != nullptr), "Please provide excatly one of either SHs or precomputed colors!");
    RIGGS_REQUIRE(((scales != nullptr && rots != nullptr) != (cov3D_precomp != nullptr)) && ((scales != nullptr) == (rots != nullptr)),
                  "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
  }
  RIGGS_REQUIRE(c->sh_degree >= 0 && c->sh_degree <= 3, "sh_degree must be 0..3");
  RIGGS_REQUIRE(shs == nullptr || c->sh_coeffs >= (c->sh_degree + 1) * (c->sh_degree + 1), "sh_coeffs too small for sh_degree");
  RIGGS_REQUIRE(!(c->glue && cov3D_precomp), "glue mode needs scales/rotations");
  GeomLayout L = geom_layout(c->num_points);
  a.N = c->num_points; a.deg = c->sh_degree; a.M = c->sh_coeffs; a.W = c->image_width; a.H = c->image_height;
  a.glue = c->glue; a.isotropic = c->isotropic; a.tight = c->tight_lists; a.defer_color = 0; a.job_rec = nullptr;
  a.tanx = c->tanfovx; a.tany = c->tanfovy; a.mod = c->scale_modifier;
  a.view = c->viewmatrix; a.proj = c->projmatrix; a.campos = c->campos;
  a.means3D = means3D; a.shs = shs; a.shs_rest = shs_rest; a.colors_precomp = colors_precomp; a.opac = opac; a.scales = scales; a.rots = rots;
  a.cov3D_precomp = cov3D_precomp; a.d_xyz = d_xyz; a.d_rot = d_rot; a.d_scaling = d_scaling;
  a.radii = radii;
  a.xyd = (float4*)(geom + L.xyd); a.conic_o = (float4*)(geom + L.conic_o); a.rgb = (float4*)(geom + L.rgb);
  a.cov3D = (float*)(geom + L.cov3D); a.clamped = (uint8_t*)(geom + L.clamped); a.tiles = (uint32_t*)(geom + L.tiles);
  a.rect = (ushort4*)(geom + L.rect); a.depth_key = (uint32_t*)(geom + L.depth_key);
  a.block_tiles = (uint32_t*)(geom + L.block_tiles);
  return 0;
}

static unsigned long long* g_raster_trace = nullptr;
static uint64_t g_raster_trace_items = 0;
// debugging aid: device buffer of 6 u64 per forward WAVE (work items * 4 waves; a work item = one 8 x 4 pixel block of one
// segment of one tile, in launch order), see render_fwd_oct_kernel; followed by the backward's per-chunk records; NULL disables
int riggs_raster_set_trace(void* dev_u64) { g_raster_trace = (unsigned long long*)dev_u64; return 0; }
// capacity of the forward part in work items (0 = 8 per tile: enough while no tile is segmented)
int riggs_raster_set_trace_items(uint64_t n_items) { g_raster_trace_items = n_items; return 0; }

int riggs_raster_preprocess(const riggs_raster_cfg* cfg, const float* means3D, const float* shs,
                            const float* shs_rest, const float* colors_precomp, const float* opacities, const float* scales,
                            const float* rotations, const float* cov3D_precomp, const float* d_xyz,
                            const float* d_rotation, const float* d_scaling, void* geom_, int32_t* radii,
                            uint32_t* counters, riggs_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  char* geom = (char*)geom_;
  PreArgs a;
  int rc = fill_pre_args(a, cfg, means3D, shs, shs_rest, colors_precomp, opacities, scales, rotations, cov3D_precomp, d_xyz,
                         d_rotation, d_scaling, geom, radii);
  if (rc) return rc;
  const int N = cfg->num_points;
  // counters = {R, overflow, "depth sort needs its third pass", -}: all four words are written by the first kernel of the
  // depth sort (a memset node in front of it is 5 us of a captured frame)
  if (N == 0) RIGGS_HIP_CHECK(hipMemsetAsync(counters, 0, 16, s));
  if (N == 0) { if (geom) defer_note(geom, 0); return 0; }
  GeomLayout L = geom_layout(N);
  // SH colours: by extra workgroups of the tile sort's scatter launch where that sort runs (color_job.h), else here
  a.job_rec = (ColorJob*)(geom + L.color_job);
  {
    const int T = ((a.W + RIGGS_TILE - 1) / RIGGS_TILE) * ((a.H + RIGGS_TILE - 1) / RIGGS_TILE);
    a.defer_color = (shs && binning_hosts_color(N, T, a.M)) ? 1 : 0;
    defer_note(geom, a.defer_color);
  }
  // counters[0] = R (and [1..3] = 0) is written by the first kernel of the depth sort
  { ProfScope ps(PROF_PREPROCESS_FWD, s); launch_preprocess_fwd(a, s); }
  if (debug_sync(cfg->debug, s, "preprocess_fwd")) return 1;
  // depth sort of the Gaussians (stable: equal depths keep ascending index): two or three counting-sort passes (csrc/binning.hip)
  {
    ProfScope ps(PROF_DEPTH_SORT, s);
    launch_depth_sort(N, (const uint32_t*)(geom + L.depth_key), (uint32_t*)(geom + L.depth_key_sorted),
                      (uint32_t*)(geom + L.order), geom + L.sort_table, a.block_tiles, counters, s);
  }
  if (debug_sync(cfg->debug, s, "depth sort")) return 1;
  return 0;
}

int riggs_raster_binning_reset_history(void* binning_, int64_t cap, int32_t N, int32_t H, int32_t W, riggs_stream stream_) {
  RIGGS_REQUIRE(binning_ != nullptr && H > 0 && W > 0, "bad arguments");
  const size_t T = (size_t)((W + RIGGS_TILE - 1) / RIGGS_TILE) * ((H + RIGGS_TILE - 1) / RIGGS_TILE);
  BinLayout B = bin_layout(cap, N, H, W);
  RIGGS_HIP_CHECK(hipMemsetAsync((char*)binning_ + B.walk_hist, 0, riggs_hist_words((uint32_t)T) * 4, (hipStream_t)stream_));
  return 0;
}

int riggs_raster_render(const riggs_raster_cfg* cfg, const void* geom_, void* binning_, int64_t cap, size_t binning_bytes,
                        void* image_, float* out_color, float* out_depth, float* out_alpha, uint32_t* counters,
                        riggs_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  RIGGS_REQUIRE(cfg != nullptr, "cfg is NULL");
  const int N = cfg->num_points, H = cfg->image_height, W = cfg->image_width;
  if (binning_bytes < bin_layout(cap, N, H, W).total) {
    set_error("riggs_raster_render: the binning arena (%zu bytes) is smaller than riggs_raster_binning_bytes(%lld, %d, %d, %d) = %zu",
              binning_bytes, (long long)cap, N, H, W, bin_layout(cap, N, H, W).total);
    return 2;
  }
  const int gx = (W + RIGGS_TILE - 1) / RIGGS_TILE, gy = (H + RIGGS_TILE - 1) / RIGGS_TILE, T = gx * gy;
  const char* geom = (const char*)geom_;
  char* bin = (char*)binning_;
  char* img = (char*)image_;
  GeomLayout G = geom_layout(N);
  ImageLayout I = image_layout(H, W);
  BinLayout B = bin_layout(cap, N, H, W);
  const bool binned = N > 0 && cap > 0;
  if (!binned) {  // (the extra workgroup of bin_scatter_kernel writes every entry of ranges / tile_max / slot_base / the counters itself)
    RIGGS_HIP_CHECK(hipMemsetAsync(img + I.ranges, 0, (I.slot_base - I.ranges) + (size_t)(T + 2) * 4, s));
    RIGGS_HIP_CHECK(hipMemsetAsync(img + I.fwd_ctr, 0, 512, s));
  }
  const uint32_t* point_list = (const uint32_t*)(bin + B.point_list);
  if (binned && defer_lookup(geom) == 1)  // (nothing binned = no instance = nobody reads a colour)
    RIGGS_REQUIRE(binning_hosts_color(N, T, cfg->sh_coeffs),
                  "riggs_raster_render: riggs_raster_preprocess left this frame's SH colours to the tile sort's scatter launch, but an "
                  "option (color_side_jobs / bin_grouped) changed since: options must not change between the two calls of a frame");
  if (binned) {
    // stable counting sort by tile (csrc/binning.hip)
    ProfScope ps(PROF_TILE_SORT, s);
    BinOut bo;
    bo.ranges = (uint2*)(img + I.ranges); bo.slot_base = (uint32_t*)(img + I.slot_base); bo.tile_max = (uint32_t*)(img + I.tile_max);
    bo.counters = counters; bo.fwd_items = (uint32_t*)(bin + B.fwd_items); bo.fwd_empty = (uint32_t*)(img + I.fwd_empty);
    bo.fwd_ctr = (uint32_t*)(img + I.fwd_ctr); bo.wide_tiles = cfg->deterministic ? 0u : forward_wide_tiles(); bo.wide_min = forward_wide_min();
    bo.walk_hist = (uint32_t*)(bin + B.walk_hist);
    bo.hist_stamp = 0x5EED0000u ^ ((uint32_t)T * 2654435761u) ^ ((uint32_t)N * 0x9E3779B1u);
    bo.viewmatrix = option(OPT_FWD_HIST_VIEW_TOL) > 0 ? cfg->viewmatrix : nullptr;
    bo.view_tol = 0.01f * (float)option(OPT_FWD_HIST_VIEW_TOL);
    int rcb = launch_binning(N, T, gx, cap, (const uint32_t*)(geom + G.order), (const uint32_t*)(geom + G.tiles),
                             (const ushort4*)(geom + G.rect), bin + B.table, bin + B.ckpt /* free until the compositing */,
                             (uint32_t*)(bin + B.point_list),
                             // the per-instance tile id is only a debugging aid here (2M scattered 4-byte stores):
                             // it is implied by `ranges`, so it is written with cfg.debug only
                             cfg->debug ? (uint32_t*)(bin + B.tile_keys) : nullptr, bo, s,
                             (const ColorJob*)(geom + G.color_job), cfg->sh_coeffs);
    if (rcb) return rcb;
    if (debug_sync(cfg->debug, s, "binning (counting sort)")) return 1;
  }
  RenderArgs r;
  r.W = W; r.H = H;
  r.trace = g_raster_trace;
  r.ranges = (const uint2*)(img + I.ranges);
  r.point_list = point_list;
  r.xyd = (const float4*)(geom + G.xyd); r.conic_o = (const float4*)(geom + G.conic_o); r.rgb = (const float4*)(geom + G.rgb);
  r.bg = cfg->bg;
  r.final_T = (float*)(img + I.final_T); r.n_contrib = (uint32_t*)(img + I.n_contrib);
  r.out_color = out_color; r.out_depth = out_depth; r.out_alpha = out_alpha;
  r.final_acc = (float4*)(img + I.final_acc); r.tile_max = (uint32_t*)(img + I.tile_max);
  r.tile_ticket = r.tile_max + (T + 1);
  r.work = (uint4*)(bin + B.work); r.work_ctr = (uint32_t*)(img + I.fwd_ctr) + 64;
  r.slot_base = (const uint32_t*)(img + I.slot_base); r.ckpt = (float*)(bin + B.ckpt);
  // longest-list-first work list of the forward (the extra workgroup of bin_scatter_kernel builds it; NULL: every tile is empty)
  r.items = nullptr; r.empties = nullptr; r.item_ctr = nullptr;
  if (binned) { r.items = (const uint32_t*)(bin + B.fwd_items); r.empties = (const uint32_t*)(img + I.fwd_empty); r.item_ctr = (uint32_t*)(img + I.fwd_ctr); }
  r.walk_hist = (uint32_t*)(bin + B.walk_hist);
  r.hist_slot_words = riggs_hist_slot_words((uint32_t)T);
  r.trace_items = g_raster_trace_items ? g_raster_trace_items : (uint64_t)T * 8;
  { ProfScope ps(PROF_RENDER_FWD, s); launch_render_fwd(r, s); }
  if (debug_sync(cfg->debug, s, "render_fwd")) return 1;
  return 0;
}

int riggs_raster_backward(const riggs_raster_cfg* cfg, const float* means3D, const float* shs,
                          const float* shs_rest, const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations, const float* cov3D_precomp, const float* d_xyz,
                          const float* d_rotation, const float* d_scaling, const int32_t* radii, const void* geom_,
                          const void* binning_,
                          int64_t cap, const void* image_, const uint32_t* counters, const float* dL_dcolor,
                          const float* dL_ddepth, const float* dL_dalpha, void* workspace, float* dL_dmeans3D,
                          float* dL_dmeans2D, float* dL_dsh, float* dL_dcolors_precomp, float* dL_dopacities,
                          float* dL_dscales, float* dL_drotations, float* dL_dcov3D, float* dL_dd_scaling,
                          float* dL_dsh_rest, riggs_stream stream_) {
  hipStream_t s = (hipStream_t)stream_;
  PreBwdArgs b;
  b.counters = counters;
  int rc = fill_pre_args(b.f, cfg, means3D, shs, shs_rest, colors_precomp, opacities, scales, rotations, cov3D_precomp, d_xyz,
                         d_rotation, d_scaling, (char*)geom_, (int32_t*)radii);
  if (rc) return rc;
  RIGGS_REQUIRE(dL_dcolor && dL_dmeans3D && dL_dmeans2D && dL_dopacities && workspace, "missing gradient buffers");
  RIGGS_REQUIRE(shs == nullptr || dL_dsh != nullptr, "dL_dsh required with shs");
  RIGGS_REQUIRE(colors_precomp == nullptr || dL_dcolors_precomp != nullptr, "dL_dcolors_precomp required");
  RIGGS_REQUIRE(cov3D_precomp == nullptr || dL_dcov3D != nullptr, "dL_dcov3D required");
  RIGGS_REQUIRE(scales == nullptr || (dL_dscales != nullptr && dL_drotations != nullptr), "dL_dscales/dL_drotations required");
  const int N = cfg->num_points, H = cfg->image_height, W = cfg->image_width;
  if (N == 0) return 0;
  const char* geom = (const char*)geom_;
  const char* bin = (const char*)binning_;
  const char* img = (const char*)image_;
  GeomLayout G = geom_layout(N);
  ImageLayout I = image_layout(H, W);
  BinLayout B = bin_layout(cap, N, H, W);
  RenderBwdArgs r;
  r.n_points = N;
  // (the backward's statistics follow the forward's: 8 blocks x 4 waves x 6 words per tile)
  {
    const size_t fwd_items = g_raster_trace_items ? (size_t)g_raster_trace_items
                                                  : (size_t)(((W + RIGGS_TILE - 1) / RIGGS_TILE) * ((H + RIGGS_TILE - 1) / RIGGS_TILE)) * 8;
    r.trace = g_raster_trace ? g_raster_trace + fwd_items * 4 * 8 : nullptr;
  }
  r.W = W; r.H = H;
  r.ranges = (const uint2*)(img + I.ranges);
  r.point_list = (const uint32_t*)(bin + B.point_list);
  r.xyd = (const float4*)(geom + G.xyd); r.conic_o = (const float4*)(geom + G.conic_o); r.rgb = (const float4*)(geom + G.rgb);
  r.bg = cfg->bg;
  r.final_T = (const float*)(img + I.final_T); r.n_contrib = (const uint32_t*)(img + I.n_contrib);
  r.dL_dcolor = dL_dcolor; r.dL_ddepth = dL_ddepth; r.dL_dalpha = dL_dalpha;
  r.gacc = (float*)workspace;
  r.final_acc = (const float4*)(img + I.final_acc); r.tile_max = (const uint32_t*)(img + I.tile_max);
  r.slot_base = (const uint32_t*)(img + I.slot_base); r.ckpt = (const float*)(bin + B.ckpt);
  r.n_tiles = ((W + RIGGS_TILE - 1) / RIGGS_TILE) * ((H + RIGGS_TILE - 1) / RIGGS_TILE);
  r.n_slots = (int64_t)B.n_slots;
  r.work = (const uint4*)(bin + B.work); r.work_ctr = (const uint32_t*)(img + I.fwd_ctr) + 64;
  r.det_rows = nullptr;
  if (cfg->deterministic && cap > 0) {
    // ordered-reduction mode: rows instead of atomics, then a fixed-order sum per Gaussian that overwrites the accumulators
    char* ws = (char*)workspace + riggs_raster_backward_workspace_bytes(N);
    r.det_rows = (float*)ws;
    uint32_t* inv = (uint32_t*)(ws + align_up((size_t)cap * 40));
    uint32_t* off = (uint32_t*)((char*)inv + align_up((size_t)cap * 4));
    RIGGS_HIP_CHECK(hipMemsetAsync(r.det_rows, 0, (size_t)cap * 40, s));
    { ProfScope ps(PROF_RENDER_BWD, s); launch_render_bwd(r, s); }
    const int gx = (W + RIGGS_TILE - 1) / RIGGS_TILE;
    launch_ordered_gather(N, r.n_tiles, gx, cap, r.ranges, r.point_list, (const uint32_t*)(geom + G.tiles),
                          (const ushort4*)(geom + G.rect), r.det_rows, inv, off, r.gacc, dL_ddepth != nullptr, s);
  } else if (cap > 0) {  // (cap == 0: nothing was composited and the accumulators — zero on entry by contract — stay zero)
    ProfScope ps(PROF_RENDER_BWD, s); launch_render_bwd(r, s);
  }
  if (debug_sync(cfg->debug, s, "render_bwd")) return 1;
  b.gacc = (float*)workspace;
  b.touched_bits = (unsigned long long*)((char*)workspace + ws_bits_offset(N));
  b.block_touched = (uint32_t*)((char*)workspace + ws_blocks_offset(N));
  b.sparse_zero = cfg->sparse_zero ? 1 : 0;
  b.dL_dmeans3D = dL_dmeans3D; b.dL_dmeans2D = dL_dmeans2D; b.dL_dsh = dL_dsh; b.dL_dcolors = dL_dcolors_precomp;
  b.dL_dopac = dL_dopacities; b.dL_dscales = dL_dscales; b.dL_drots = dL_drotations; b.dL_dcov3D = dL_dcov3D;
  b.dL_dd_scaling = dL_dd_scaling; b.dL_dsh_rest = dL_dsh_rest;
  RIGGS_REQUIRE(shs_rest == nullptr || dL_dsh_rest != nullptr, "dL_dsh_rest required with shs_rest");
  { ProfScope ps(PROF_PREPROCESS_BWD, s); launch_preprocess_bwd(b, s); }
  if (debug_sync(cfg->debug, s, "preprocess_bwd")) return 1;
  return 0;
}

}  // extern "C"
