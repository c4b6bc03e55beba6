/*
 * riggs_hip.h — C ABI of libriggs_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for RigGS's per-frame hot path.  Every entry point names the
 * reference interface it replaces (paths relative to the RigGS tree).  Plain
 * pointers and sizes only: all `const float*` / `float*` arguments are DEVICE
 * pointers (HBM) unless stated, `stream` is a hipStream_t, nothing here owns
 * memory — the caller (PyTorch's caching allocator, or any hipMalloc) does.
 * Every function returns 0 on success, non-zero on error (see riggs_last_error()).
 *
 * Binding stubs for the reference side are shown in INTEGRATION.md.
 */
#ifndef RIGGS_HIP_H
#define RIGGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* riggs_stream; /* hipStream_t */

int riggs_version(void);
const char* riggs_last_error(void);

/* Library options (process-wide; the library reads NOTHING from the environment).  Names and defaults:
 *   "fwd_wide_tiles"    256   at most this many tiles per frame are composited by the forward's 32-lanes-per-pixel blocks (0 = none)
 *   "fwd_wide_min"      4096  ... the tiles whose walk went this many instances deep in the previous frame of the same arena
 *                             and whose list is that long now (minimum 256; a negative value restores the default)
 *   "fwd_hist_view_tol" 20    ... in the last frame of the SAME VIEW: the arena keeps 128 walk histories, each with the view matrix of its
 *                             frames; a frame uses (and updates) the one no entry of whose matrix is further than value / 100 from
 *                             cfg.viewmatrix, or starts a new one (round robin).  0 = one history whatever the view
 *   "bin_grouped"       -1    which tile sort riggs_raster_render runs where both fit: -1 = chosen by the number of Gaussians
 *                             and of tiles, 0 = the direct counting sort, 1 = the two-level sort (identical lists, bit for bit).
 *                             riggs_raster_binning_bytes reserves the larger of the two layouts, so flipping this never
 *                             invalidates an arena
 *   "cnode_bwd_atomics" 0     1 = riggs_cnode_backward's first design (LDS float atomics); set it BEFORE sizing its workspace
 *   "color_side_jobs"   1     the SH colours of a frame are evaluated by extra workgroups of the tile sort's scatter launch
 *                             (riggs_raster_render) wherever the direct tile sort runs, instead of by riggs_raster_preprocess's
 *                             kernel: the same colours and clamp bits, bit for bit; the geometry arena's colours are then
 *                             complete after riggs_raster_render, not after riggs_raster_preprocess.  0 = always in preprocess.
 *                             Like "bin_grouped", it must not change between the two calls of one frame
 *   "preprocess_bwd_lean" -1  riggs_raster_backward's per-Gaussian kernel as one wave per 256 Gaussians (no LDS image, every block
 *                             resident at once): -1 = with cfg.sparse_zero (sparse gradient rows: -3 .. -6 us of a frame; with
 *                             every row written the 256-thread form is 18 us faster), 0 / 1 = never / always.  Same results
 *   "pose_mlp_layered"  0     1 = riggs_pose_mlp_forward / _backward(_fk) run one launch per layer (no workgroup hand-off inside a
 *                             launch, nothing that can time out; ~20 launches instead of 2) whatever the network's size: what a
 *                             host re-runs a frame with after the one-launch kernels reported a lost hand-off.  It must not
 *                             change between the forward and the backward of one frame
 * Unknown names fail.  Set them between frames, not while a launch that reads them is being issued from another thread. */
int riggs_set_option(const char* name, int32_t value);
int riggs_get_option(const char* name, int32_t* value);

/* =====================================================================
 * Rasterizer.  Replaces the un-vendored CUDA extension
 *   diff_gaussian_rasterization._C.rasterize_gaussians / rasterize_gaussians_backward
 * reached from gaussian_renderer/__init__.py:14,57-72 (settings) and :133-141 (call).
 * ===================================================================== */

/* Mirror of GaussianRasterizationSettings (gaussian_renderer/__init__.py:57-70).
 * bg / viewmatrix / projmatrix / campos are device tensors exactly as render() builds them
 * (4x4 matrices in the transposed row-vector convention of scene/cameras.py:61-71). */
typedef struct riggs_raster_cfg {
  int32_t num_points;     /* N */
  int32_t sh_degree;      /* active degree 0..3 */
  int32_t sh_coeffs;      /* M = shs.shape[1] (16 for max degree 3); 0 with colors_precomp */
  int32_t image_height;
  int32_t image_width;
  float tanfovx;
  float tanfovy;
  float scale_modifier;
  const float* bg;         /* (3,) */
  const float* viewmatrix; /* (4,4) */
  const float* projmatrix; /* (4,4) */
  const float* campos;     /* (3,) */
  int32_t debug;           /* pipe.debug: synchronise + validate after each stage */
  /* Fused "render glue" (gaussian_renderer/__init__.py:74-92, scene/gaussian_model.py:104-132):
   * when glue != 0 the rasterizer takes RAW parameters and applies
   *   means3D = xyz + d_xyz, opacity = sigmoid(_opacity), scales = exp(_scaling) + d_scaling,
   *   rotations = normalize(_rotation + d_rotation)
   * in-kernel, and the backward returns gradients w.r.t. the raw tensors. */
  int32_t glue;
  int32_t isotropic;       /* glue only: _scaling is (N,1) repeated (gaussian_model.py:105-108) */
  /* Ordered-reduction mode of the compositing backward ("deterministic-reduction mode for grads"): instead of
   * float atomics into the per-Gaussian accumulators, every (tile instance) writes its partial gradient row and a second
   * kernel sums each Gaussian's rows in ascending tile order: gradients are bitwise reproducible run to run.  Needs the
   * larger workspace of riggs_raster_backward_workspace_bytes_ordered; slower (tests / debugging).
   * riggs_raster_render: with it the forward never uses the previous frame's walk history, so it is bitwise reproducible
   * frame after frame; without it the forward is bitwise reproducible only while no tile is composited wide — always in a
   * fresh (zero-filled) binning arena — because a tile composited by the 32-lane blocks folds its sums in another order
   * (same values to ~1e-7; "fwd_wide_tiles" = 0 turns that off as well). */
  int32_t deterministic;
  /* riggs_raster_backward only.  1 = the caller guarantees that the gradient output buffers are the SAME buffers the previous
   * riggs_raster_backward with this workspace wrote and that nobody has written them since (or that buffers and workspace
   * were zero-filled together): rows that receive no gradient now and received none then are not rewritten — they still
   * hold their zeros.  The zero fill of those rows (86-93 % of them per frame) is two thirds of the per-Gaussian backward's
   * HBM traffic.  A captured frame with static gradient buffers qualifies (riggs_amd.graph.GraphedFrame); an eager
   * autograd backward, whose outputs are fresh allocations, does not.  riggs_grad_rows_unpack keeps the guarantee intact
   * when it is given the workspace (it records the rows it writes). */
  int32_t sparse_zero;
  /* 1 = tight instance lists: a Gaussian's tile rectangle (upstream: every tile its ceil(3 sigma) square overlaps) is cut down
   * to the tiles in which alpha = o * exp(power) can reach 1/255 at some pixel centre, by the axis-aligned box of that
   * region (conservative margins).  The dropped instances contribute nothing to any pixel or gradient — they fail the
   * alpha test at every pixel of their tile — so images and gradients are the canonical ones to rounding (the lists'
   * positions move, and with them the grouping of the transmittance products); `radii` is unchanged; the instance list is
   * the canonical list with those instances removed, in the same order.  About a fifth of the instances of a translucent
   * scene (tools/dead_instances.py); the canonical lists (0) are the default and what the ordering parity tests pin. */
  int32_t tight_lists;
} riggs_raster_cfg;

/* Opaque arenas (upstream: geomBuffer / binningBuffer / imgBuffer byte tensors).  The binning arena's size depends on ALL
 * four arguments (the tile sort's tables are sized by the number of Gaussians and of tiles, not only by the instance
 * capacity) and is NOT monotonic in them in general: ask again whenever num_points, image_height or image_width change
 * (in either direction), and pass the arena's byte size to riggs_raster_render, which rejects an arena that is too small
 * for its arguments.  A freshly allocated binning arena must be zero-filled once (its walk-history stamp). */
size_t riggs_raster_geom_bytes(int32_t num_points);
size_t riggs_raster_image_bytes(int32_t image_height, int32_t image_width);
size_t riggs_raster_binning_bytes(int64_t instance_capacity, int32_t num_points, int32_t image_height,
                                  int32_t image_width);
/* The binning arena carries ONE piece of state from frame to frame: how deep the forward walked every tile's list (which
 * tiles the next frame composites with 32 lanes per pixel), kept PER VIEW (128 histories, each with the view matrix of its
 * frames: "fwd_hist_view_tol"), valid when a stamp word derived from the tile and Gaussian counts follows it.  Call this after allocating an arena and whenever (capacity, N, H, W) change for an arena in use
 * (the words' offset depends on them): the next frame then starts without a history.  Equivalent: zero-fill the arena. */
int riggs_raster_binning_reset_history(void* binning, int64_t instance_capacity, int32_t num_points, int32_t image_height,
                                       int32_t image_width, riggs_stream stream);

/* Field offsets (bytes) inside the arenas, for tests / debugging tools. */
enum {
  RIGGS_GEOM_XYD = 0,      /* float4 (px, py, depth, x half extent of the alpha >= 1/255 box) */
  RIGGS_GEOM_CONIC_O,      /* float4 (A, B, C, opacity) */
  RIGGS_GEOM_RGB,          /* float4 (r, g, b, y half extent of that box); complete after riggs_raster_render ("color_side_jobs") */
  RIGGS_GEOM_COV3D,        /* float[6] */
  RIGGS_GEOM_CLAMPED,      /* uint8 bitmask (bit ch) */
  RIGGS_GEOM_TILES,        /* uint32 tiles_touched */
  RIGGS_GEOM_RECT,         /* ushort4 (x0, y0, x1, y1) */
  RIGGS_GEOM_DEPTH_ORDER,  /* uint32: Gaussian indices sorted by depth bits */
  RIGGS_GEOM_NFIELDS_
};
enum {
  RIGGS_IMG_FINAL_T = 0,   /* float  H*W */
  RIGGS_IMG_N_CONTRIB,     /* uint32 H*W */
  RIGGS_IMG_RANGES,        /* uint2  tiles */
  RIGGS_IMG_FWD_CTR,       /* uint32[3] of the last forward: non-empty tiles, tiles composited by the 32-lane blocks, empty tiles */
  RIGGS_IMG_NFIELDS_
};
enum {
  RIGGS_BIN_POINT_LIST = 0, /* uint32 [capacity]: Gaussian index per sorted instance */
  RIGGS_BIN_TILE_KEYS,      /* uint32 [capacity]: tile id per sorted instance (written with cfg.debug only) */
  RIGGS_BIN_WALK_HIST,      /* uint32: the forward's walk histories, one per view — [0] round-robin cursor, [1] the slot of the last frame,
                               [16 ..] 128 slots of ((tiles + 18) rounded up to 16) words: depths per tile, a stamp, the view matrix */
  RIGGS_BIN_NFIELDS_
};
int riggs_raster_geom_layout(int32_t num_points, size_t* offsets /*[RIGGS_GEOM_NFIELDS_]*/);
int riggs_raster_image_layout(int32_t image_height, int32_t image_width, size_t* offsets /*[RIGGS_IMG_NFIELDS_]*/);
int riggs_raster_binning_layout(int64_t instance_capacity, int32_t num_points, int32_t image_height,
                                int32_t image_width, size_t* offsets /*[RIGGS_BIN_NFIELDS_]*/);

/* Stage 1 (per Gaussian): projection, covariance, SH colour, tile rectangle; then the
 * depth sort of the Gaussians and the scan of tiles_touched.  Writes the number of tile
 * instances R into counters[0] (device).  Inputs follow GaussianRasterizer.forward
 * (gaussian_renderer/__init__.py:133-141): exactly one of shs / colors_precomp and exactly
 * one of (scales, rotations) / cov3D_precomp is non-NULL.
 * With cfg->glue: means3D=_xyz, opacities=_opacity(logit), scales=_scaling(log), rotations=_rotation,
 * and d_xyz (N,3) / d_rotation (N,4) / d_scaling (N,3) may be NULL (treated as 0, the Python-float case;
 * scales = exp(_scaling) + d_scaling as at gaussian_renderer/__init__.py:89). */
int riggs_raster_preprocess(const riggs_raster_cfg* cfg, const float* means3D, const float* shs,
                            const float* shs_rest /* NULL, or (N,M-1,3) with shs = (N,1,3): the reference's
                            _features_dc / _features_rest pair read in place, no torch.cat */,
                            const float* colors_precomp, const float* opacities, const float* scales,
                            const float* rotations, const float* cov3D_precomp, const float* d_xyz,
                            const float* d_rotation, const float* d_scaling, void* geom, int32_t* radii,
                            uint32_t* counters /* device [4]: R, overflow, rsv, rsv */, riggs_stream stream);

/* Stage 2: stable counting sort of the (depth-ordered) Gaussians' tile instances by tile — the order of upstream's
 * duplicateWithKeys + 64-bit key sort + identifyTileRanges — and the per-tile alpha compositing.
 * Limits: at most 65 535 tiles and ~4 200 groups of eight tiles in a row (3840 x 2160 px is fine): beyond 25 600 tiles, and
 * from 500 000 Gaussians over 4 096 tiles on, the sort runs in two levels (by tile group, then by tile; the same list bit
 * for bit; riggs_set_option("bin_grouped", 0 / 1) forces the choice where both fit) — larger images are rejected with an
 * error, never mis-rendered.  `instance_capacity` bounds R: if R > capacity the launch is
 * still memory-safe, counters[1] is set to 1 and the image is undefined (caller retries
 * with a larger arena; riggs_amd.rasterizer does that). */
int riggs_raster_render(const riggs_raster_cfg* cfg, const void* geom, void* binning, int64_t instance_capacity,
                        size_t binning_bytes /* size of `binning`: must be >= riggs_raster_binning_bytes(capacity, N, H, W) */,
                        void* image_state, float* out_color /*(3,H,W)*/, float* out_depth /*(1,H,W)*/,
                        float* out_alpha /*(1,H,W)*/, uint32_t* counters, riggs_stream stream);

/* Backward of both stages.  Gradient outputs are fully written (no pre-zeroing needed).
 * `counters` is the forward's: when its overflow flag (counters[1]) is set the frame composited truncated
 * lists, and the backward writes exact ZERO gradients (device-side guard: an optimizer step queued behind it —
 * e.g. inside a captured hipGraph — sees zeros, never garbage); NULL skips the guard.
 * dL_ddepth / dL_dalpha may be NULL (RigGS stage 2 uses only "render": train_rig.py:499).
 * Outputs that do not apply (e.g. dL_dsh with colors_precomp) may be NULL.
 * With cfg->glue the outputs are gradients w.r.t. the raw tensors: dL_dmeans3D = dL/d_xyz = dL/dd_xyz,
 * dL_dopacities = dL/d_opacity(logit), dL_dscales = dL/d_scaling(log; (N,1) when isotropic),
 * dL_drotations = dL/d_rotation = dL/dd_rotation. */
int riggs_raster_backward(const riggs_raster_cfg* cfg, const float* means3D, const float* shs,
                          const float* shs_rest, const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations, const float* cov3D_precomp, const float* d_xyz,
                          const float* d_rotation, const float* d_scaling, const int32_t* radii, const void* geom, const void* binning,
                          int64_t instance_capacity, const void* image_state, const uint32_t* counters,
                          const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                          void* workspace /* riggs_raster_backward_workspace_bytes(N); must be ALL ZERO on first use; the
                          accumulators in it are all zero again on return (self-cleaning): zero it once after allocating and
                          keep it — for THIS num_points: behind the accumulators the call leaves one bit per Gaussian "received a gradient"
                          (riggs_grad_rows_*), at an offset that depends on num_points; re-zero the buffer before another size uses it */,
                          float* dL_dmeans3D,
                          float* dL_dmeans2D /*(N,3)*/, float* dL_dsh, float* dL_dcolors_precomp,
                          float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                          float* dL_dd_scaling /* glue only, may be NULL */,
                          float* dL_dsh_rest /* with shs_rest: (N,M-1,3), dL_dsh is then (N,1,3) */,
                          riggs_stream stream);
size_t riggs_raster_backward_workspace_bytes(int32_t num_points);
/* where the workspace keeps its row list (one bit per Gaussian: "holds a gradient"): a caller that has written the gradient
 * buffers itself (e.g. a dense all-reduce in place) fills these bytes with 0xFF before the next backward with
 * cfg.sparse_zero, which then rewrites every row */
int riggs_raster_backward_workspace_rows(int32_t num_points, size_t* offset, size_t* bytes);
/* cfg->deterministic: [accumulators | one row of 10 floats per tile instance | the instances of every Gaussian in tile order |
 * their offsets]; need not be zeroed */
size_t riggs_raster_backward_workspace_bytes_ordered(int32_t num_points, int64_t instance_capacity);

/* =====================================================================
 * Skeleton deformation.  Replaces the torch-op graph of
 *   SkeletonWarp.deform_by_pose  skeleton_utils/skeleton_warp.py:130-172
 * ===================================================================== */

/* Forward kinematics + helpers on ONE workgroup:
 *   quaternion_to_matrix      utils/time_utils.py:115-132
 *   chain_product_transform   skeleton_utils/skeleton_warp.py:242-273
 *   matrix_to_quaternion      utils/time_utils.py:146-205 (on the detached global rotations)
 * local_rot (J,4) un-normalised wxyz; joints (J,3); parents (J,) int32 with parents[i] < i.
 * Outputs: transforms (J,12) = rows of [R|t] 3x4, node_rot (J,4), d_nodes (J,3) = posed + global_trans. */
int riggs_fk_forward(int32_t num_joints, const float* local_rot, const float* joints, const int32_t* parents,
                     const float* global_trans, float* transforms, float* node_rot, float* d_nodes,
                     riggs_stream stream);
/* Reverse sweep: (dL/dtransforms (J,12), dL/dd_nodes (J,3) or NULL) -> dL/dlocal_rot (J,4);
 * accumulates sum_j dL/dd_nodes into dL_dglobal_trans (3) (+=). */
int riggs_fk_backward(int32_t num_joints, const float* local_rot, const float* joints, const int32_t* parents,
                      const float* dL_dtransforms, const float* dL_dd_nodes, float* dL_dlocal_rot,
                      float* dL_dglobal_trans, riggs_stream stream);

/* Bone-distance skinning weights + linear blend skinning, fused:
 *   cal_nn_weight_skeleton        skeleton_warp.py:41-76   (gs_kernel, K = -1 or top-K)
 *   line_segment_distance         skeleton_warp.py:215-238 (squared)
 *   LBS of means / quaternion     skeleton_warp.py:149-165
 * x (N,3), motion_mask (N,) or NULL (= ones), node_radius_log (J,) = SkeletonWarp._node_radius.
 * Outputs d_xyz (N,3), d_rotation (N,4); optional nn_weight (N,Kp) / nn_idx (N,Kp) int64 where
 * Kp = J-1 (K<=0) or K — needed by render_rig.py:156-158, not by training (may be NULL). */
int riggs_lbs_forward(int32_t num_points, int32_t num_joints, int32_t K, const float* x, const float* joints,
                      const int32_t* parents, const float* node_radius_log, const float* transforms,
                      const float* node_rot, const float* global_trans, const float* motion_mask,
                      const float* weight_mod /* NULL, or (N, J-1) = sigmoid(WeightMLP(x)): skeleton_warp.py:56-69; K = -1 only */,
                      float* d_xyz, float* d_rotation, float* nn_weight, int64_t* nn_idx,
                      void* bone_table /* may be NULL; riggs_lbs_bone_table_bytes() of device scratch, see below */,
                      riggs_stream stream);
/* bone_table (riggs_lbs_forward / riggs_lbs_forward_fk; may be NULL): device scratch of riggs_lbs_bone_table_bytes() bytes.  With
 * it the all-bones forward (K = -1, no weight_mod, no nn outputs) of a LARGE scene (>= 1 M Gaussians, >= 16 joints) reads the bone
 * records through the scalar cache as SGPR operands — a one-workgroup launch writes the table (and, in the _fk form, runs the
 * chain) in front — instead of staging them in every workgroup's LDS: 2 M x 64 joints 129 -> 104 us, results bit-identical.
 * riggs_set_option("lbs_scalar", 1 / -1) forces / forbids the form at every size (default 0: by size). */
size_t riggs_lbs_bone_table_bytes(void);
/* The same with the forward kinematics INSIDE the launch (every workgroup runs the chain of J - 1 dependent 3x4 products
 * itself — 2 us — instead of a launch of its own in front — 6.5 us of a captured frame): takes the pose, and workgroup 0
 * leaves what riggs_fk_forward would have written (transforms (J,12), node_rot (J,4), d_nodes (J,3)) for the backward and
 * the caller.  SkeletonWarp.forward(x, t, mask) uses it (riggs_amd/skeleton.py: _PoseDeform). */
int riggs_lbs_forward_fk(int32_t num_points, int32_t num_joints, int32_t K, const float* x, const float* joints,
                         const int32_t* parents, const float* node_radius_log, const float* local_rot,
                         const float* global_trans, const float* motion_mask, const float* weight_mod,
                         float* transforms, float* node_rot, float* d_nodes, float* d_xyz, float* d_rotation,
                         void* bone_table, riggs_stream stream);
/* Backward: cotangents g_xyz (N,3), g_rot (N,4) -> dL/dtransforms (J,12), dL/dnode_radius_log (J),
 * dL/dglobal_trans (3), optional dL/dmotion_mask (N).  Reduction over N is done in-kernel
 * (registers -> workgroup partials -> a fixed-order second stage: run-to-run deterministic).
 * No gradient to x or joints (both detached in the reference: skeleton_warp.py:16,44,131). */
int riggs_lbs_backward(int32_t num_points, int32_t num_joints, int32_t K, const float* x, const float* joints,
                       const int32_t* parents, const float* node_radius_log, const float* transforms,
                       const float* node_rot, const float* global_trans, const float* motion_mask,
                       const float* weight_mod, const float* g_xyz, const float* g_rot, float* dL_dtransforms,
                       float* dL_dnode_radius_log, float* dL_dglobal_trans, float* dL_dmotion_mask,
                       float* dL_dweight_mod /* (N, J-1), required with weight_mod */,
                       void* workspace /* riggs_lbs_backward_workspace_bytes(N, J) */, riggs_stream stream);
size_t riggs_lbs_backward_workspace_bytes(int32_t num_points, int32_t num_joints);

/* =====================================================================
 * PoseMLP (time -> J quaternions + root translation), batch of ONE row:
 *   PoseMLP.forward   skeleton_utils/network_utils.py:134-150  (8 x Linear(256) + ReLU, the
 *   embedding re-concatenated IN FRONT of h after layer `skip`, two linear heads) with the
 *   embedding of utils/time_utils.py:208-256 for input_dims = 1.
 * weights[l] / biases[l] are HOST arrays of `depth` DEVICE pointers to torch's nn.Linear tensors
 * (row-major (out, in)).  `acts` (riggs_pose_mlp_acts_floats floats) is written by forward and
 * read by backward (which also uses its tail — zeroed by forward — for its own hand-off state).  backward writes every parameter gradient into ONE flat buffer laid out as
 *   [W_0, b_0, ..., W_{depth-1}, b_{depth-1}, W_rot, b_rot, W_tr, b_tr]   (no gradient to t).
 * sync_state (may be NULL): riggs_pose_mlp_sync_bytes bytes of device memory, ZEROED ONCE by the caller
 * and then owned by this network (one launch in flight at a time): the forward runs as one launch whose
 * workgroups hand the layer outputs to each other through it, and a generation counter inside makes
 * every launch's tags unique.  With NULL a private copy inside `acts` is cleared by a memset node per call.
 * rot_bias4 (4 floats, may be NULL) is added to every predicted quaternion: the identity bias
 * [1,0,0,0] of skeleton_warp.py:118 folded into the head instead of a separate elementwise op.
 * ===================================================================== */
size_t riggs_pose_mlp_acts_floats(int32_t depth, int32_t width, int32_t multires);
size_t riggs_pose_mlp_sync_bytes(int32_t depth, int32_t width);
/* Index (32-bit words) of the STICKY status word inside sync_state.  The one-launch kernels pass data between
 * workgroups by bounded spinning, which needs all of a launch's workgroups co-resident (<= 96 workgroups of 512
 * threads: true on an otherwise idle MI355X; NOT guaranteed when another process or a concurrent stream holds
 * CUs for long).  A spin that times out poisons that launch's outputs with NaN and sets bit 0 of this word; no
 * kernel ever clears it.  The host reads it after a step (PoseMLP.check_status / GraphedFrame.check) and raises.
 * The word behind it is a TEST HOOK: bit 0 / bit 1 make one workgroup of the forward / backward launch keep a layer's
 * hand-off to itself, so that the time-out path can be exercised on an idle GPU (tests/test_gpu_deform.py); leave it 0. */
size_t riggs_pose_mlp_status_word(int32_t depth, int32_t width);
/* Placement of the one-launch kernels' chain (process-wide; default 1): 1 = its workgroups are every eighth workgroup of the
 * launch, i.e. on ONE XCD when the dispatcher deals round-robin, and hand their layers over through that XCD's L2 (verified
 * inside every launch; falls back by itself when one XCD cannot hold the chain) — 5-7 us faster per launch on an idle device;
 * 0 = the launch's first workgroups, on all XCDs: choose it when another stream keeps compute units busy for long (RCCL
 * collectives overlapped with the deformation backward: the chain needs EVERY compute unit of its XCD to hold one of its
 * workgroups, and waits for the collective's kernel where it cannot).  riggs_amd.dist selects 0 for world sizes > 1. */
int riggs_pose_mlp_set_placement(int32_t one_xcd);
/* debugging aid: 128 device u64 that workgroup 0 of the one-launch kernels stamps with the 100 MHz wall clock
 * per stage (forward [0,64), backward [64,128)); NULL (the default) disables it. */
int riggs_pose_mlp_set_trace(void* dev_u64x128);
size_t riggs_pose_mlp_backward_workspace_floats(int32_t depth, int32_t width, int32_t multires);
int riggs_pose_mlp_forward(int32_t depth, int32_t width, int32_t multires, int32_t skip, int32_t n_rot,
                           const float* const* weights, const float* const* biases, const float* W_rot,
                           const float* b_rot, const float* W_tr, const float* b_tr, const float* t,
                           const float* rot_bias4, void* sync_state, float* acts, float* rotation,
                           float* translation, riggs_stream stream);
int riggs_pose_mlp_backward(int32_t depth, int32_t width, int32_t multires, int32_t skip, int32_t n_rot,
                            const float* const* weights, const float* const* biases, const float* W_rot,
                            const float* b_rot, const float* W_tr, const float* b_tr, float* acts,
                            const float* g_rotation, const float* g_translation, float* workspace,
                            float* flat_grads, void* sync_state /* the forward's, or NULL */, riggs_stream stream);

/* riggs_pose_mlp_backward with the reverse sweep of the kinematic chain (riggs_fk_backward) in front, in the same launch:
 * the gradients of the heads' outputs are computed from (dL/dtransforms (J,12), dL/dd_nodes (J,3) or NULL) by every workgroup
 * of the chain while its weights load, instead of by a launch of one workgroup in front (10 us of a captured frame).
 * g_rotation (4J, may be NULL) is ADDED to the sweep's dL/dlocal_rot (other consumers of the predicted quaternions);
 * g_translation (3, may be NULL) is the gradient of the predicted translation from elsewhere (the skinning's
 * dL/dglobal_trans), to which sum_j dL/dd_nodes_j is added.  dL_dlocal_rot (J,4) and dL_dglobal_trans (3) receive the two
 * totals (required for networks that run one launch per layer, else optional).  n_rot must be 4 * num_joints.
 * template_fixed_coef (device scalar, may be NULL): the template frame's pose regulariser of the stage-2 objective
 * (train_rig.py:474-482: lambda_template_fixed * mean((local_rotation - (1,0,0,0))^2), template camera only) as a cotangent:
 * dL/dlocal_rot += coef * (local_rot - unit), coef = 2 lambda / (4 J) on the template frame and 0 elsewhere, refreshed by the host
 * between replays of a captured iteration; template_fixed_loss (device float, may be NULL) receives mean((local_rot - unit)^2),
 * the value the reference logs.  No launch of its own. */
int riggs_pose_mlp_backward_fk(int32_t depth, int32_t width, int32_t multires, int32_t skip, int32_t n_rot,
                               const float* const* weights, const float* const* biases, const float* W_rot,
                               const float* b_rot, const float* W_tr, const float* b_tr, float* acts, int32_t num_joints,
                               const float* local_rot, const float* joints, const int32_t* parents,
                               const float* transforms /* (J,12) as the forward wrote them, or NULL: the chain is re-run */,
                               const float* dL_dtransforms, const float* dL_dd_nodes, const float* g_rotation,
                               const float* g_translation, float* dL_dlocal_rot, float* dL_dglobal_trans,
                               const float* template_fixed_coef, float* template_fixed_loss, float* workspace,
                               float* flat_grads, void* sync_state, riggs_stream stream);

/* =====================================================================
 * The whole frame behind ONE call per direction — what train_rig.py:535-554 issues per iteration as skeleton.step() + render()
 * + loss.backward(): riggs_pose_mlp_forward -> riggs_lbs_forward_fk -> riggs_raster_preprocess -> riggs_raster_render, and
 * riggs_raster_backward -> riggs_lbs_backward -> riggs_pose_mlp_backward_fk.  Same kernels, same results; the point is the
 * host: an eagerly issued frame crosses the binding seven times and is host-bound (0.65 ms per frame at the bench workload
 * against 0.36 ms of device time); through these two entries it is two crossings.  Every field has the meaning of the
 * like-named argument of the entry points above.  The rasterizer runs with the fused render glue (cfg.glue must be 1): xyz /
 * opacity / scaling / rotation are the RAW parameters, features_dc / features_rest the two SH tensors read in place; `xyz` is
 * also the (detached) input of the skinning.  The binning arena must hold `instance_capacity` instances (a caller that does
 * not know R yet runs its first frame through the separate entry points, as riggs_amd.rasterizer.RasterArena does).
 * ===================================================================== */
typedef struct riggs_frame {
  /* PoseMLP (riggs_pose_mlp_forward) */
  int32_t depth, width, multires, skip, n_rot;
  const float* const* weights; const float* const* biases;       /* host arrays of `depth` device pointers */
  const float *W_rot, *b_rot, *W_tr, *b_tr, *t, *rot_bias4;
  void* sync_state; float* acts;
  float* local_rot;      /* out (J,4) */
  float* global_trans;   /* out (3,) */
  /* skeleton (riggs_lbs_forward_fk) */
  int32_t num_joints, K;
  const float* joints; const int32_t* parents; const float* node_radius_log; const float* motion_mask; const float* weight_mod;
  float *transforms, *node_rot, *d_nodes;   /* out (J,12), (J,4), (J,3) */
  float *d_xyz, *d_rotation;                /* out (N,3), (N,4): the residuals the rasterizer's glue adds */
  /* rasterizer (riggs_raster_preprocess + riggs_raster_render) */
  riggs_raster_cfg cfg;
  const float *xyz, *features_dc, *features_rest, *opacity, *scaling, *rotation, *d_scaling /* or NULL */;
  void* geom; int32_t* radii; uint32_t* counters;
  void* binning; int64_t instance_capacity; size_t binning_bytes; void* image_state;
  float *out_color, *out_depth, *out_alpha;
} riggs_frame;

typedef struct riggs_frame_grads {
  const float *dL_dcolor, *dL_ddepth /* or NULL */, *dL_dalpha /* or NULL */;
  void* raster_workspace;                    /* riggs_raster_backward_workspace_bytes(N), see riggs_raster_backward */
  float *dL_dxyz, *dL_dmeans2D, *dL_dfeatures_dc, *dL_dfeatures_rest, *dL_dopacity, *dL_dscaling, *dL_drotation;
  float* dL_dd_scaling;                      /* or NULL */
  float* dL_dtransforms;                     /* (J,12) scratch: the skinning's gradient, consumed by the chain's reverse sweep */
  float* dL_dnode_radius_log;                /* (J,) */
  float* dL_dglobal_trans_skinning;          /* (3,) scratch */
  float *dL_dmotion_mask, *dL_dweight_mod;   /* or NULL */
  void* lbs_workspace;                       /* riggs_lbs_backward_workspace_bytes(N, J) */
  const float *dL_dd_nodes, *g_local_rot, *g_global_trans; /* cotangents of d_nodes (J,3) / local_rot (J,4) / global_trans (3,)
                                                from other consumers (regularisers, the projection loss), or NULL */
  float *dL_dlocal_rot, *dL_dglobal_trans;   /* out (J,4), (3,) */
  float* pose_workspace;                     /* riggs_pose_mlp_backward_workspace_floats */
  float* pose_flat_grads;                    /* the PoseMLP's parameter gradients, flat, in parameter order */
} riggs_frame_grads;

int riggs_frame_forward(const riggs_frame* frame, riggs_stream stream);
int riggs_frame_backward(const riggs_frame* frame, const riggs_frame_grads* grads, riggs_stream stream);

/* =====================================================================
 * Gaussian optimizer.
 *   GaussianModel.training_setup  scene/gaussian_model.py:197-221 builds torch.optim.Adam(l, lr=0.0, eps=1e-15)
 *   with ONE parameter tensor per group and per-group learning rates; train_rig.py:527 steps it.
 * riggs_adam_step applies that update (plain Adam: no weight decay, no amsgrad; torch's single-tensor
 * operation order) to up to 32 parameter tensors in ONE launch.  All pointer arrays are HOST arrays of
 * DEVICE pointers (16-byte aligned tensors of numel[k] floats); lr[k] is the group's current learning
 * rate, step[k] the step count AFTER this update (>= 1); exp_avg / exp_avg_sq are updated in place.
 *   add_densification_stats       scene/gaussian_model.py:516-518
 *   max_radii2D update            train_rig.py:333-335
 * riggs_densify_stats: for every i with update_filter[i] (bytes, torch.bool layout):
 *   xyz_gradient_accum[i] += ||viewspace_grad[i, :2]||, denom[i] += 1, and if max_radii2D and radii are
 *   non-NULL max_radii2D[i] = max(max_radii2D[i], radii[i]).  viewspace_grad is (N,3).
 * ===================================================================== */
int riggs_adam_step(int32_t n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, const double* lr, const int64_t* step, double beta1,
                    double beta2, double eps, riggs_stream stream);
/* riggs_adam_step for a trainer that issues every call eagerly and never looks at the frame's status words (an unmodified
 * train_rig.py): an element whose gradient is NaN or Inf — what a frame poisoned by a lost PoseMLP hand-off produces — keeps its
 * parameter and both moments, and *nonfinite_count (device u32, not NULL, never reset by the library) is incremented per such
 * element; every other element is updated exactly as by riggs_adam_step.  (The step counts of such a trainer live on the host:
 * a whole-step gate would leave them ahead of the device.  Inside a hipGraph use riggs_adam_step_gated.) */
int riggs_adam_step_guarded(int32_t n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                            float* const* exp_avg_sq, const int64_t* numel, const double* lr, const int64_t* step, double beta1,
                            double beta2, double eps, uint32_t* nonfinite_count, riggs_stream stream);
/* The same update with the step counts (and optionally the learning rates) in DEVICE memory, the layout of
 * torch.optim.Adam(capturable=True): step_dev[k] / lr_dev[k] are HOST arrays of DEVICE pointers to 0-dim float tensors;
 * step_dev[k][0] holds the count AFTER this update (the caller increments it on the stream beforehand); lr_dev may be
 * NULL or hold NULL entries (then lr[k], a host value baked into the launch, is used).  Safe to capture in a hipGraph. */
int riggs_adam_step_capturable(int32_t n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, const double* lr,
                               const float* const* step_dev, const float* const* lr_dev, double beta1, double beta2,
                               double eps, riggs_stream stream);
/* ---- A frame's "valid" gate.  Every way a frame of this library can go wrong WITHOUT the host noticing in time leaves a
 * non-zero device word behind: the sticky status word of the one-launch PoseMLP kernels (a workgroup hand-off timed out, the
 * pose was poisoned with NaN: sync_state[riggs_pose_mlp_status_word()]), the rasterizer's counters[1] (bit 0: the instance
 * arena overflowed, the lists were truncated; bit 1: the depth sort's in-launch barrier timed out), the gradient-row
 * exchange's status[1] (a segment overflowed or some rank's frame was invalid: nothing was unpacked).  A host that steps its
 * optimizer inside the same hipGraph cannot look at them first — so the kernels that CONSUME a frame's gradients take the
 * words themselves: when (word[i][0] & mask[i]) != 0 for any i < n they leave every output untouched.
 *   riggs_adam_step_gated         = riggs_adam_step_capturable that also advances the step counts (step_dev[k][0] += 1, as a
 *                                   first launch) — parameters, both moments and the counts are bit-identical to before the
 *                                   call when the gate is set, and *skipped (device u32, may be NULL) is incremented instead.
 *   riggs_grad_rows_pack_gated    = riggs_grad_rows_pack that marks the segment "frame invalid" (rows needed = 0xFFFFFFFF)
 *                                   instead of packing: every rank's riggs_grad_rows_unpack then skips (gradients untouched) and
 *                                   raises bit 1 of status[1] — which the ranks' optimizers take as THEIR gate, so that all
 *                                   replicas skip the step together and stay bit-identical.
 *   riggs_gate_flag               writes 1.0f (gate set) or 0.0f into `flag` — a spare float INSIDE the buffer of a dense
 *                                   gradient all-reduce (sum or average), issued in front of that collective: afterwards the
 *                                   slot is non-zero on every rank when SOME rank's frame was invalid (the NaN of a poisoned
 *                                   frame spreads to every rank's gradients through the sum; so does this flag), and serves as
 *                                   a gate word (mask 0x7FFFFFFF) of every rank's optimizer.
 */
#define RIGGS_GATE_MAX 4
typedef struct riggs_gate {
    int32_t n;                                  /* words in use (0: never gated) */
    int32_t reserved;
    const uint32_t* word[RIGGS_GATE_MAX];       /* device pointers */
    uint32_t mask[RIGGS_GATE_MAX];
} riggs_gate;
int riggs_gate_flag(const riggs_gate* gate, float* flag, riggs_stream stream);
/* (advance_steps = 0: the counts were advanced by riggs_adam_steps_advance_gated — one launch for up to 128 of them, what a
 * host with more than 32 parameter tensors issues once in front of its riggs_adam_step_gated calls; `skipped` is counted there) */
int riggs_adam_steps_advance_gated(int32_t n_steps, float* const* step_dev, const riggs_gate* gate, uint32_t* skipped,
                                   riggs_stream stream);
int riggs_adam_step_gated(int32_t n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                          float* const* exp_avg_sq, const int64_t* numel, const double* lr, float* const* step_dev,
                          const float* const* lr_dev, double beta1, double beta2, double eps, const riggs_gate* gate,
                          uint32_t* skipped, int32_t advance_steps, riggs_stream stream);
/* The gated pair with the bias corrections evaluated ONCE: riggs_adam_steps_advance_coef advances the counts like
 * riggs_adam_steps_advance_gated and writes coef[2 k] = 1 - beta1^t, coef[2 k + 1] = sqrt(1 - beta2^t) of the new counts (double
 * precision, one thread per count; coef: 2 n_steps device floats); riggs_adam_step_gated_coef = riggs_adam_step_gated
 * (advance_steps = 0) reading them (coef: the 2 n_groups floats of ITS tensors) instead of evaluating two double-precision pow()
 * in the prologue of each of its 16 384 workgroups — a tenth of the update's time.  All counts of one call share the betas. */
int riggs_adam_steps_advance_coef(int32_t n_steps, float* const* step_dev, const riggs_gate* gate, uint32_t* skipped, double beta1,
                                  double beta2, float* coef, riggs_stream stream);
int riggs_adam_step_gated_coef(int32_t n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, const double* lr, float* const* step_dev,
                               const float* const* lr_dev, double beta1, double beta2, double eps, const riggs_gate* gate,
                               const float* coef, riggs_stream stream);
int riggs_densify_stats(int32_t num_points, const float* viewspace_grad, const uint8_t* update_filter,
                        const int32_t* radii, float* xyz_gradient_accum, float* denom, float* max_radii2D,
                        riggs_stream stream);

/* =====================================================================
 * Gradient-row exchange of the frame-sharded step (the reference is single-GPU:
 * utils/general_utils.py:207 pins cuda:0, so this has no reference counterpart — it is what the data-parallel
 * caller of riggs_raster_backward binds instead of an all-reduce over all N rows).
 * riggs_raster_backward leaves in its workspace which Gaussians received a gradient this frame (the others' gradients
 * are exactly zero in every tensor).  riggs_grad_rows_pack copies those rows — (Gaussian index, row of every tensor in
 * `grads`, times `scale`) — in ascending Gaussian order into `segment`:
 *   32-bit words [0] rows stored = min(needed, capacity), [1] rows needed, [2] N, [3] row_floats,
 *   [4 .. 4 + ceil(N/256)] first row of every block of 256 Gaussians, then (16-byte aligned) `capacity` rows of
 *   row_floats = riggs_grad_rows_row_floats(...) floats (index bits first, zero padded to a multiple of 4).
 * The caller all-gathers the segments of all ranks (equal riggs_grad_rows_segment_bytes) and calls
 * riggs_grad_rows_unpack: per Gaussian the rows are combined IN RANK ORDER — first occurrence overwrites, later ones are
 * added; rows in no segment are left as they are (zero on every rank) — without atomics, so every rank obtains the same
 * bits.  `grads` are HOST arrays of DEVICE pointers to (N, widths[k]) row-major float tensors (<= 8).  `status` (8 words;
 * [4] = THIS call's flag — [1]'s bits for this call alone, rewritten every call: the word a step's optimizers gate on; [0..3])
 * is STICKY — only ever raised by the kernel, cleared by whoever reads it: [0] = the largest number of rows a segment
 * needed, [1] != 0 when in some call a segment overflowed `capacity` or did not match (N, row_floats) (bit 0) or was marked
 * "frame invalid" by riggs_grad_rows_pack_gated (bit 1): in that call NOTHING
 * was unpacked (the gradients kept their local values) and the caller has to exchange that step densely — or, when it polls
 * only every k steps and the optimizers have already stepped on un-averaged gradients, re-synchronise the replicas; [2] = calls
 * since it was cleared, [3] = the call (1-based) that failed first.
 * `backward_workspace` is the workspace the last riggs_raster_backward of these N Gaussians used, on the same stream. */
int32_t riggs_grad_rows_row_floats(int32_t n_tensors, const int32_t* widths);
size_t riggs_grad_rows_segment_bytes(int32_t num_points, int32_t row_floats, int32_t capacity);
int riggs_grad_rows_pack(int32_t num_points, const void* backward_workspace, int32_t n_tensors, const float* const* grads,
                         const int32_t* widths, float scale, int32_t capacity, void* segment, riggs_stream stream);
int riggs_grad_rows_pack_gated(int32_t num_points, const void* backward_workspace, int32_t n_tensors, const float* const* grads,
                               const int32_t* widths, float scale, int32_t capacity, void* segment, const riggs_gate* gate,
                               riggs_stream stream);
/* backward_workspace (may be NULL): when given, the rows written are recorded in it as "rows that hold a gradient", which
 * keeps cfg.sparse_zero of the next riggs_raster_backward valid (rows other ranks touched are zeroed there when due). */
int riggs_grad_rows_unpack(int32_t num_points, int32_t world, int32_t capacity, const void* segments, int32_t n_tensors,
                           float* const* grads, const int32_t* widths, uint32_t* status, void* backward_workspace,
                           riggs_stream stream);
/* Test / measurement aid (multi-GPU readiness on one GPU): n_cus workgroups that each claim a whole compute unit's LDS — so they
 * sit on n_cus distinct CUs, as a collective library's channel kernels do beside a rank's frame — and spin until the host raises
 * *stop_flag (a DEVICE word, raised by a fill on another stream: pinned host memory is not coherent for a running kernel by default) or max_ms have passed (bounded).  *started (device u32, zeroed by the
 * caller) counts the workgroups that are resident.  Launch it on a stream of its own. */
int riggs_debug_pin_cus(int32_t n_cus, const int32_t* stop_flag, uint32_t max_ms, uint32_t* started, riggs_stream stream);

/* =====================================================================
 * Image loss: utils/loss_utils.py:17-18 (l1_loss), :33-77 (ssim, 11x11 Gaussian window,
 * sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2, mean over all elements), used at train_rig.py:508-509.
 * image / gt are (C, H, W) float32.  forward writes out3 = {l1 = mean |image - gt|, ssim = mean ssim_map,
 * (1 - lambda_dssim) l1 + lambda_dssim (1 - ssim)} (device; the third is the trainer's loss_img, train_rig.py:509) and
 * keeps three derivative maps + per-workgroup partial sums in `state` (riggs_l1_ssim_state_floats floats); backward
 * takes the upstream gradients of the three scalars as DEVICE scalars (NULL = 0) and writes dL/dimage (C, H, W).
 * ===================================================================== */
size_t riggs_l1_ssim_state_floats(int32_t C, int32_t H, int32_t W);
int riggs_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float* image, const float* gt, float lambda_dssim,
                          float* state, float* out3, riggs_stream stream);
int riggs_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float* image, const float* gt, const float* state,
                           float lambda_dssim, const float* g_l1, const float* g_ssim, const float* g_loss,
                           float* dL_dimage, riggs_stream stream);

/* =====================================================================
 * Skeleton projection loss: TrainRig.cal_skeleton_loss, train_rig.py:309-314, =
 * sampling_skeleton_points (:264-276) -> project_nodes_to_2d_elements (utils/other_utils.py:101-127) ->
 * pytorch3d.loss.chamfer_distance(x, y, norm=1) (third-party; restated from its published definition).
 * S line parameters t (the host evaluates the reference's int(max_len / (sum_len / 512)) and linspace) on each of the
 * J-1 bones of the posed joints d_nodes (J, 3); world_view_transform is the camera's (4, 4) matrix as the reference
 * stores it (row-vector convention, DEVICE pointer); fx, fy, cx, cy as other_utils.py:107-117 computes them; thinned is
 * the frame's (M, 2) silhouette-skeleton pixels as (row, col).  forward writes loss2 = {loss, weight * loss} (device;
 * `weight` is the trainer's robust per-frame weight of train_rig.py:465-467 as a DEVICE scalar, NULL = 1) and keeps the
 * nearest-neighbour keys in `state` (riggs_skeleton_projection_state_floats floats, 8-byte aligned); backward takes the
 * upstream gradients of the two scalars as DEVICE scalars (NULL = 0) and writes dL/d(d_nodes) (J, 3).  Deterministic
 * (the only atomics are 64-bit max and LDS integer adds).  pixel_count: optional DEVICE int32 (1 <= count <= M): the number
 * of valid rows of `thinned`, M then being the buffer's capacity — one captured graph serves frames of any pixel count.
 * ===================================================================== */
size_t riggs_skeleton_projection_state_floats(int32_t J, int32_t S, int32_t M);
int riggs_skeleton_projection_forward(int32_t J, int32_t S, int32_t M, const int32_t* parents, const float* d_nodes,
                                      const float* t, const float* world_view_transform, float fx, float fy, float cx,
                                      float cy, const float* thinned, const int32_t* pixel_count, const float* weight, float* state,
                                      float* loss2, riggs_stream stream);
int riggs_skeleton_projection_backward(int32_t J, int32_t S, int32_t M, const int32_t* parents, const float* d_nodes,
                                       const float* t, const float* world_view_transform, float fx, float fy, float cx,
                                       float cy, const float* thinned, const int32_t* pixel_count, const float* weight,
                                       float* state, const float* g_loss, const float* g_weighted, float* grad_nodes,
                                       riggs_stream stream);

/* =====================================================================
 * Stage-1 control-node deformation, per-Gaussian part: ControlNodeWarp.cal_nn_weight
 * (utils/time_utils.py:934-964) + the blend of ControlNodeWarp.forward (:1138-1191) for skinning = False,
 * node_trans_bias = None, pred_opacity = pred_color = False.  pytorch3d.ops.knn_points (third-party) is restated as
 * "K smallest squared distances, ascending, ties to the lowest index".
 * x (N, 3); feature (N, feat_stride), its first `hyper` columns are the hyper coordinates (NULL and hyper = 0: xyz only);
 * motion_mask (N) or NULL (= 1); nodes (M, node_stride) = xyz then hyper coordinates; node_radius_log = _node_radius,
 * node_weight_logit = _node_weight (NULL: with_node_weight = False); node_trans / node_rot / node_scale / local_rot are the
 * node network's d_xyz (M, 3), d_rotation (M, 4), d_scaling (M, 3), local_rotation (M, 4; raw, (1,0,0,0) is added inside).
 * flags: 1 = local_frame, 2 = d_rot_as_res.  K <= 8, hyper <= 11, M (3 + hyper rounded up to 4) floats <= 128 KB.
 * forward writes d_xyz, d_rotation, d_scaling and cal_nn_weight's nn_idx (int32), nn_weight, nn_dist (N, K).
 * backward takes upstream gradients (each may be NULL = 0) and writes the gradients of feature (N, feat_stride; columns
 * beyond `hyper` zero), motion_mask, the node attributes, _node_radius, _node_weight and the nodes' hyper coordinates
 * (M, hyper); xyz of Gaussians and nodes are detached in the reference (:944, :947-949, :1152).  workspace:
 * riggs_cnode_backward_workspace_floats floats, 16-byte aligned.
 * ===================================================================== */
int riggs_cnode_backward_blocks(int32_t N, int32_t M, int32_t hyper);
size_t riggs_cnode_backward_workspace_floats(int32_t N, int32_t M, int32_t K, int32_t hyper);
int riggs_cnode_forward(int32_t N, int32_t M, int32_t K, int32_t hyper, int32_t feat_stride, int32_t node_stride, int32_t flags,
                        const float* x, const float* feature, const float* motion_mask, const float* nodes,
                        const float* node_radius_log, const float* node_weight_logit, const float* node_trans,
                        const float* node_rot, const float* node_scale, const float* local_rot, float* d_xyz, float* d_rot,
                        float* d_scale, int32_t* nn_idx, float* nn_weight, float* nn_dist, riggs_stream stream);
int riggs_cnode_backward(int32_t N, int32_t M, int32_t K, int32_t hyper, int32_t feat_stride, int32_t node_stride, int32_t flags,
                         const float* x, const float* feature, const float* motion_mask, const float* nodes,
                         const float* node_radius_log, const float* node_weight_logit, const float* node_trans,
                         const float* node_rot, const float* node_scale, const float* local_rot, const int32_t* nn_idx,
                         const float* nn_dist, const float* g_xyz, const float* g_rot, const float* g_scale,
                         float* g_feature, float* g_motion_mask, float* g_node_trans, float* g_node_rot, float* g_node_scale,
                         float* g_local_rot, float* g_node_radius_log, float* g_node_weight_logit, float* g_nodes_hyper,
                         float* workspace, riggs_stream stream);

/* =====================================================================
 * simple_knn._C.distCUDA2 (scene/gaussian_model.py:20,170): mean squared distance to the 3
 * nearest neighbours.  points (P,3) -> out (P,).  workspace: riggs_knn_workspace_bytes(P).
 * ===================================================================== */
size_t riggs_knn_workspace_bytes(int32_t num_points);
int riggs_dist2_knn3(int32_t num_points, const float* points, float* out, void* workspace, riggs_stream stream);
/* the exact all-pairs search riggs_dist2_knn3 uses below 2048 points, callable at any size: the grid search's test oracle */
int riggs_dist2_knn3_bruteforce(int32_t num_points, const float* points, float* out, riggs_stream stream);

/* =====================================================================
 * Densification / pruning of the Gaussian cloud on the device (SURVEY.md §2 row 6 "next") — scene/gaussian_model.py:
 * densify_and_prune :500-514, densify_and_clone :475-498, densify_and_split :440-473, prune_points :373-392 and the optimizer
 * surgery around them (:338-417), called every densification_interval iterations at train_rig.py:359-365.
 *   riggs_densify_select   the three predicates of ONE densify_and_prune call as byte flags (3, N): [0] the old row survives
 *                          (not split, not pruned), [1] a clone of it is made and survives, [2] its split children survive
 *                          (grads = accum / denom with NaN -> 0; clone: |grad| >= thr and max scale <= dense_limit; split: grad
 *                          >= thr and max scale > dense_limit; prune: sigmoid(opacity) < min_opacity or, with world_limit >= 0,
 *                          max scale > world_limit — the children with their scale / child_div; the reference's screen-size
 *                          test never fires there: densification_postfix has zeroed max_radii2D by then)
 *   riggs_compact_indices  ascending indices of the set flags + their number (device int32), three launches, no atomics
 *   riggs_rows_gather      dst[t][m, :] = src[t][plan[m], :] for up to 32 tensors in ONE launch; plan[m] < 0 marks a NEW row made
 *                          from source ~plan[m]: tensors with zero_new[t] (the Adam moments) get zeros there
 *   riggs_split_children   the children's position R(q_parent) (z * scale_parent) + xyz_parent and log(scale_parent / child_div)
 *                          (z: unit normals, (n_children, 3); child j has parent parents[j % n_parents]: copy-major like .repeat(N, 1))
 * ===================================================================== */
int riggs_densify_select(int32_t num_points, int32_t scaling_columns, const float* xyz_gradient_accum, const float* denom,
                         const float* scaling, const float* opacity, float grad_threshold, float dense_limit, float min_opacity,
                         float world_limit, float child_div, uint8_t* flags, riggs_stream stream);
size_t riggs_compact_workspace_bytes(int32_t num_points);
int riggs_compact_indices(int32_t num_points, const uint8_t* flags, int32_t* out_indices, int32_t* count, void* workspace,
                          riggs_stream stream);
int riggs_rows_gather(int32_t n_out, const int32_t* plan, int32_t n_tensors, const float* const* src, float* const* dst,
                      const int32_t* row_floats, const uint8_t* zero_new, riggs_stream stream);
int riggs_split_children(int32_t n_children, int32_t n_parents, int32_t scaling_columns, const int32_t* parents,
                         const float* unit_normals, const float* xyz, const float* scaling, const float* rotation, float child_div,
                         float* new_xyz, float* new_scaling, riggs_stream stream);

/* =====================================================================
 * Dual-quaternion blending of rigid transforms — utils/dual_quaternion.py: QT2DQ :135-143, DQ2QT :146-165,
 * DQBlending :168-179 (what these two entry points compute), interpolate :182-187 and transformation_blending :190-197
 * (host compositions of them: riggs_amd/dual_quaternion.py).  The reference never calls the module on the skeleton path
 * (SURVEY.md §0.3); BASELINE.json's north_star names it, so it is here as an operator of its own.
 *   shared != 0: ONE set of K <= 1024 transforms for all rows — q (K, 4) raw quaternions (w, x, y, z), t (K, 3),
 *                weights (N, K): skinning.   shared == 0: every row has its own K <= 8 — q (N, K, 4), t (N, K, 3), weights (N, K).
 *   norm_over_nodes: QT2DQ normalises with torch.nn.functional.normalize(q), whose default axis is dim=1 — the quaternion
 *                axis of a 2-D q (0 here), the NODE axis of a 3-D q (1 here: every component is divided by its norm over the K
 *                nodes).  Both are the reference's results for the respective input rank.
 *   out_mode 0: out_rot (N, 9) row-major rotation matrix (rot_as_q=False), 1: out_rot (N, 4) through matrix_to_quaternion
 *                (rot_as_q=True; no sign standardisation), 2: out_rot (N, 16) = [R | t; 0 0 0 1] and out_t unused.
 * The dual part of a node is standardize_quaternion((0, t) * q) / 2 as in the reference (its sign does not follow q's).
 * Backward: cotangents in the layout of the outputs (g_t NULL = zeros; ignored with out_mode 2); dL_dq / dL_dt in the shape
 * of q / t, dL_dweights (N, K) or NULL; deterministic (no atomics).  workspace: riggs_dqb_backward_workspace_floats floats.
 * ===================================================================== */
int riggs_dqb_forward(int32_t num_rows, int32_t K, int32_t shared, int32_t norm_over_nodes, int32_t out_mode, const float* q,
                      const float* t, const float* weights, float* out_rot, float* out_t, riggs_stream stream);
size_t riggs_dqb_backward_workspace_floats(int32_t num_rows, int32_t K, int32_t shared);
int riggs_dqb_backward(int32_t num_rows, int32_t K, int32_t shared, int32_t norm_over_nodes, int32_t out_mode, const float* q,
                       const float* t, const float* weights, const float* g_rot, const float* g_t, float* dL_dq, float* dL_dt,
                       float* dL_dweights, float* workspace, riggs_stream stream);

/* =====================================================================
 * Kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * `mask` has bit i set to time stage i; 0 disables (default: no events, no overhead).
 * riggs_prof_read synchronises on the recorded events and returns the sum / count since the
 * last riggs_prof_reset.  Stage ids: riggs_prof_name(i) for i in [0, riggs_prof_count()).
 * ===================================================================== */
/* debugging aid: per-wave statistics of the forward compositing kernel (8 u64 per wave, 4 waves per work item — a work
 * item is one pixel block of one tile = one workgroup, in launch order: {100 MHz ticks of the compositing loop, rounds,
 * survivors | hardware id << 32, steps, steps with a contribution, list length | tile << 32 | wide << 63, start tick, 0}),
 * followed (at word n_items * 32; n_items = riggs_raster_set_trace_items, default 8 per tile) by 4 u64 per chunk of the
 * compositing backward ({start, end, hardware id, workgroup << 32 | tile << 16 | chunk}); tools/fwd_trace.py, bwd_trace.py.
 * (How many tiles the forward may composite with 32 lanes per pixel and from which walk depth: riggs_set_option.)
 * NULL disables */
int riggs_raster_set_trace(void* dev_u64);
int riggs_raster_set_trace_items(uint64_t n_items);
/* =====================================================================
 * Per-Gaussian MLP heads on the matrix cores: WeightMLP / DeformMLP of
 * skeleton_utils/network_utils.py:6-112 as one fused launch per direction — 16-bit operands, fp32 accumulation
 * (v_mfma_f32_32x32x16_bf16 / _f16).  Every "bf16" buffer below holds the format the `fp16` argument selects, the same in
 * all five calls: 0 = bfloat16 (fp32's range, 8 significand bits), 1 = IEEE half (11 significand bits: parameter gradients
 * within 1-2 % of the fp32 arithmetic instead of 3-11 %, same rate; its range is the caller's business — see g_scale).
 * x_emb_bf16 (N rounded up to 128, in_pad) bf16, zero padded -> depth x [Linear(256) + ReLU], the embedding
 * re-concatenated IN FRONT of the hidden vector after layer `skip` (network_utils.py:58-61, 103-106) -> Linear(out_ch <= 32).
 * weights_bf16[l]: the 256 x K_l weights (K_0 = in_pad, K_{skip+1} = in_pad + 256 with the embedding columns first, else 256;
 * in_pad = in_ch rounded up to 64, padding columns zero) in the kernels' FRAGMENT-MAJOR order — [neuron tile T of 32][K-step s
 * of 16][lane 0..63][8 values], value j of a lane = W[32 T + (lane & 31)][16 s + 8 (lane >> 5) + j]: the 1 KB a wave loads per
 * (tile, step) is one contiguous run; w_out_bf16: the head alike, one tile of 32 outputs (>= out_ch zero) x 16 steps.
 * riggs_mlp_pack produces all of them from the fp32 masters; no caller needs to know the order.
 * acts_bf16 (depth, N, 256) receives the post-ReLU activations (operand of the weight gradients) and relu_masks
 * (depth, ceil(N / riggs_mlp_rows_per_workgroup()), 256) x 16 bytes their signs in the kernels' accumulator layout, for
 * riggs_mlp_backward (both NULL for inference).
 * n_rows_dev (riggs_mlp_forward / _backward / _wgrad; may be NULL): a device int32 — only rows < min(*n_rows_dev, N) exist; N
 * stays the row stride of acts_bf16 / dpre_bf16 and sizes the grid (the rest of the workgroups leave at once).  This is how the
 * row-sparse backward runs inside a hipGraph on the rows riggs_mlp_live_rows compacted, their number known to the device only.
 * Opt-in on the host side (riggs_amd.mlp): the reference computes these MLPs in fp32.
 * ===================================================================== */
/* Output epilogue of riggs_mlp_forward (host struct, may be NULL = the head's plain value), applied by the launch that holds the
 * value anyway instead of elementwise launches behind it:
 *   sigmoid:  out = sigmoid(head)  — WeightMLP (network_utils.py:107); the backward then multiplies the cotangent by
 *             out (1 - out) (riggs_mlp_live_rows / riggs_mlp_cotangent: sigmoid_out);
 *   res_base / res_mask / res_out:  res_out[n][c] = res_base[n][c] + out[n][c] * res_mask[n]  ((N, out_ch), (N) or NULL = 1,
 *             (N, out_ch)) — DeformMLP: the template offsets join the blended translation in front of the motion mask
 *             (skeleton_warp.py:152-161); the backward's counterpart is riggs_mlp_cotangent's g_rows / row_mask. */
struct riggs_mlp_epilogue {
  int32_t sigmoid;
  int32_t reserved;
  const float* res_base;
  const float* res_mask;
  float* res_out;
};
int riggs_mlp_forward(int32_t N, int32_t in_ch, int32_t out_ch, int32_t depth, int32_t skip,
                      const void* const* weights_bf16, const float* const* biases, const void* w_out_bf16,
                      const float* b_out, const void* x_emb_bf16, void* acts_bf16, void* relu_masks, float* out,
                      const int32_t* n_rows_dev, const struct riggs_mlp_epilogue* epilogue, int32_t fp16, riggs_stream stream);
/* Data-gradient pass of the same MLP: g_out (N, out_ch) = dL/d(output) -> dpre_bf16 (depth, N, 256) = dL/d(pre-activation)
 * of every hidden layer, the operand of the weight gradients  dW_l = dpre_l^T · input_l ,  db_l = sum_n dpre_l
 * (riggs_mlp_wgrad).  weights_t_bf16[l] (l >= 1): W_l[:, hidden part]^T (rows = the units of layer
 * l - 1), 256 x 256 values, fragment-major as above; w_out_t_bf16: W_out^T (rows = hidden units, 32 padded outputs as
 * inputs), fragment-major.  No gradient w.r.t. x_emb (detached in the reference).
 * g_scale (device scalar, may be NULL = 1): g_out is multiplied by it on load, so dpre and db_partial come out scaled by it —
 * with fp16 the caller passes a power of two that lifts max|g_out| to ~2^10 (loss-scaled gradients: a per-pixel-averaged
 * loss leaves |g_out| ~ 1e-7, below half precision's normal range) and divides the parameter gradients by it. */
int riggs_mlp_backward(int32_t N, int32_t out_ch, int32_t depth, int32_t skip, const void* const* weights_t_bf16,
                       const void* w_out_t_bf16, const float* g_out, const float* g_scale, const void* relu_masks,
                       void* dpre_bf16, float* db_partial, const int32_t* n_rows_dev, int32_t fp16, riggs_stream stream);
/* Row-sparse backward, step 1: the rows of g_out (N, out_ch) that hold a non-zero — the Gaussians the render gave a gradient;
 * the WeightMLP has no other cotangent: its regulariser is dead code at train_rig.py:433-444, and the skinning backward writes
 * exact zeros for the rest — in ascending order: live_idx (N) their indices, live_count (device int32) their number M,
 * x_live_bf16 ((N rounded up to 128), in_pad) the gathered rows of x_emb_bf16 (zero-filled up to the next multiple of 128
 * behind row M), g_live (N, out_ch) the gathered rows of g_out.  Two launches, no atomics (deterministic).  A zero row
 * contributes zero to every data gradient, bias sum and weight product, so riggs_mlp_forward (activations of the live rows
 * only: the first forward then stores none), riggs_mlp_backward and riggs_mlp_wgrad on (x_live, g_live, n_rows_dev =
 * live_count) return the parameter gradients of the dense pass.  workspace: riggs_mlp_live_rows_workspace_bytes(N), 8-byte
 * aligned.
 * sigmoid_out (N, out_ch; may be NULL): the head's output went through riggs_mlp_epilogue.sigmoid — the cotangent of the head is
 * g_out * s (1 - s), which is what is tested for a non-zero and gathered into g_live (no sigmoid_backward launch in front).
 * scale (device float; may be NULL): receives the fp16 gradient scale of riggs_mlp_grad_scale over that cotangent — per-block
 * maxima from the first launch, reduced by the second: no launch and no atomic of its own. */
size_t riggs_mlp_live_rows_workspace_bytes(int32_t N);
int riggs_mlp_live_rows(int32_t N, int32_t out_ch, int32_t in_ch, const float* g_out, const float* sigmoid_out, const void* x_emb_bf16,
                        void* workspace, int32_t* live_idx, int32_t* live_count, void* x_live_bf16, float* g_live, float* scale,
                        riggs_stream stream);
/* The g_scale of the fp16 format from the gradient itself: scale[0] = 2^floor(log2(1024 / max|g|)) (max|g| clamped at 1e-30), two
 * launches, no host synchronisation.  zero_word: a device u32 that is ZERO on entry (the caller clears it once) and zero again
 * behind the call. */
int riggs_mlp_grad_scale(int64_t n, const float* g, float* scale, uint32_t* zero_word, riggs_stream stream);
/* riggs_mlp_grad_scale with an L2 regulariser on the MLP's OUTPUT folded in — the template offsets' term of the stage-2
 * objective, train_rig.py:446-454: lambda * mean(template_offsets^2) over all Gaussians (x1e3 on the template frame) —
 *   g_eff = g + coef[0] * out   (n floats each; coef: a device scalar = 2 lambda / n, refreshed by the host between replays),
 * scale[0] as riggs_mlp_grad_scale from max|g_eff|, mean_sq[0] (may be NULL) = mean(out^2), the value the reference logs.
 * The same two launches; partials512: 512 floats of scratch.  The data-gradient and weight-gradient passes then read g_eff. */
int riggs_mlp_l2_grad_scale(int64_t n, const float* g, const float* out, const float* coef, float* g_eff, float* scale,
                            uint32_t* zero_word, float* partials512, float* mean_sq, riggs_stream stream);
/* The general form (the same two launches): the cotangent the data-gradient pass reads, for an (N, out_ch) head,
 *   g_eff = (g + g_rows * row_mask[row]) * [s (1 - s)] + l2_coef[0] * l2_out
 * g, g_rows (N, out_ch): at least one; row_mask (N; NULL = 1; needs g_rows): the motion mask that multiplied the head's output on
 * its way into res_out (riggs_mlp_epilogue); sigmoid_out (N, out_ch; NULL = no factor): the head's sigmoid-ed output;
 * l2_out / l2_coef (both or neither): the L2 term above.  scale / zero_word / partials512 / mean_sq as above. */
int riggs_mlp_cotangent(int32_t N, int32_t out_ch, const float* g, const float* g_rows, const float* row_mask,
                        const float* sigmoid_out, const float* l2_out, const float* l2_coef, float* g_eff, float* scale,
                        uint32_t* zero_word, float* partials512, float* mean_sq, riggs_stream stream);
/* db_partial: (ceil(N / riggs_mlp_rows_per_workgroup()), depth, 256) fp32 — per-workgroup column sums of dpre; the bias
 * gradients are their sum over the first axis.  May be NULL when riggs_mlp_wgrad follows (it sums the columns itself). */
int32_t riggs_mlp_rows_per_workgroup(void);
/* The parameter gradients of one MLP from what the two passes left in memory, in three launches: every product
 *   dW_l = dpre_l^T · input_l   (input_0 = x_emb, input_l = acts_{l-1}; layer skip + 1: [x_emb | acts_skip]),
 *   dW_out = (g_out · g_scale)^T · acts_{depth-1},   db_l = sum_n dpre_l,   db_out = sum_n g_out · g_scale
 * streamed once by one workgroup per CU (split over the Gaussians, fp32 accumulators, LDS-direct loads — mlp_wgrad.hip), the
 * split partials summed, divided by g_scale (NULL = 1) and written to the fp32 tensors torch.autograd hands back:
 * grad_weights[l] (256, K_true) row-major with K_true = in_ch / in_ch + 256 (layer skip + 1) / 256, grad_biases[l] (256),
 * grad_w_out (out_ch, 256), grad_b_out (out_ch).  x_emb_bf16 / acts_bf16 / dpre_bf16: the buffers of riggs_mlp_forward /
 * riggs_mlp_backward, same format.  workspace: riggs_mlp_wgrad_workspace_bytes bytes, 256-byte aligned, contents irrelevant
 * (depends on N, the shape and the device's CU count). */
size_t riggs_mlp_wgrad_workspace_bytes(int32_t N, int32_t in_ch, int32_t depth, int32_t skip);
int riggs_mlp_wgrad(int32_t N, int32_t in_ch, int32_t out_ch, int32_t depth, int32_t skip, const void* x_emb_bf16,
                    const void* acts_bf16, const void* dpre_bf16, const float* g_out, const float* g_scale, void* workspace,
                    size_t workspace_bytes, float* const* grad_weights, float* const* grad_biases, float* grad_w_out,
                    float* grad_b_out, const int32_t* n_rows_dev, int32_t fp16, riggs_stream stream);
/* riggs_mlp_wgrad for an MLP whose input ends in tail_ch values that are THE SAME FOR EVERY ROW (DeformMLP: the pose vector,
 * /root/reference/skeleton_utils/skeleton_warp.py:152 `pose = local_rot.detach().reshape(-1)[None].expand(N, -1)`): the masters'
 * weights of the two layers that read the input have in_ch + tail_ch input columns, the packed operands and x_emb_bf16 only the
 * first in_ch (riggs_mlp_pack_tail); the gradient of the tail's columns is (bias gradient) x tail — written by the launch that
 * sums the partials.  tail: tail_ch floats on the device.  tail_ch = 0: riggs_mlp_wgrad. */
int riggs_mlp_wgrad_tail(int32_t N, int32_t in_ch, int32_t tail_ch, const float* tail, int32_t out_ch, int32_t depth, int32_t skip,
                         const void* x_emb_bf16, const void* acts_bf16, const void* dpre_bf16, const float* g_out, const float* g_scale,
                         void* workspace, size_t workspace_bytes, float* const* grad_weights, float* const* grad_biases,
                         float* grad_w_out, float* grad_b_out, const int32_t* n_rows_dev, int32_t fp16, riggs_stream stream);
/* fp32 master weights ((256, K_true) row-major per layer, (out_ch, 256) for the head) -> every bf16 operand the two kernels
 * read (layouts above), in one launch. */
int riggs_mlp_pack(int32_t in_ch, int32_t out_ch, int32_t depth, int32_t skip, const float* const* weights, const float* w_out,
                   void* const* weights_bf16, void* const* weights_t_bf16, void* w_out_bf16, void* w_out_t_bf16,
                   int32_t fp16, riggs_stream stream);
/* riggs_mlp_pack for masters with a constant input tail (riggs_mlp_wgrad_tail): K_true = in_ch + tail_ch (+ 256 in layer skip + 1);
 * the tail's columns are left out of the packed operands. */
int riggs_mlp_pack_tail(int32_t in_ch, int32_t tail_ch, int32_t out_ch, int32_t depth, int32_t skip, const float* const* weights,
                        const float* w_out, void* const* weights_bf16, void* const* weights_t_bf16, void* w_out_bf16,
                        void* w_out_t_bf16, int32_t fp16, riggs_stream stream);
/* ... and what they contribute: bias_eff (2, 256) = [b_first + W_first[:, in_ch : in_ch + tail_ch] tail,
 * b_skip + W_skip[:, in_ch : in_ch + tail_ch] tail] in fp32 — the biases riggs_mlp_forward takes for layer 0 and layer skip + 1
 * (network_utils.py:46-51: h = cat([x_emb, t_emb]) enters both).  One launch per forward: the tail changes with every frame. */
int riggs_mlp_tail_bias(int32_t in_ch, int32_t tail_ch, const float* w_first, const float* b_first, const float* w_skip,
                        const float* b_skip, const float* tail, float* bias_eff, riggs_stream stream);
/* The kernels' input operand from positions: row n = [x_n, sin(2^k x_n), cos(2^k x_n) for k < multires, tail (n_tail floats,
 * the same for every row: DeformMLP's pose), 0 ...] as bf16, (N rounded up to 128) x (width rounded up to 64)
 * (utils/time_utils.py:208-256 get_embedder + the concatenation of network_utils.py:40-46). */
int riggs_mlp_embed(int32_t N, int32_t multires, int32_t n_tail, const float* x, const float* tail, void* out_bf16,
                    int32_t fp16, riggs_stream stream);
/* self-test of the MFMA fragment layouts mlp.hip assumes: writes D = A B for A = [I_16; 0] and an asymmetric B */
int riggs_mlp_layout_probe(float* out32x32, riggs_stream stream);

int riggs_prof_count(void);
const char* riggs_prof_name(int32_t id);
int riggs_prof_enable(uint32_t mask);
int riggs_prof_reset(void);
int riggs_prof_read(int32_t id, float* total_ms, int32_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* RIGGS_HIP_H */
