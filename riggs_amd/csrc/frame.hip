// One host transition per direction for the whole frame (include/riggs_hip.h: riggs_frame_forward / riggs_frame_backward):
// PoseMLP -> forward kinematics + skinning -> fused render glue + preprocess -> depth sort -> tile sort -> compositing, and
// back.  Nothing is computed here that the entry points it strings together do not compute: it removes the host's share of an
// eagerly issued frame (an unmodified trainer crosses the ctypes boundary seven times per frame and pays ~0.3 ms of Python
// for it; train_rig.py:535-554), it is not another kernel path.
#include "common.h"

using namespace riggs;

__global__ void frame_add3_kernel(float* __restrict__ dst, const float* __restrict__ src) { dst[threadIdx.x] += src[threadIdx.x]; }

extern "C" {

int riggs_frame_forward(const riggs_frame* f, riggs_stream s) {
  RIGGS_REQUIRE(f != nullptr, "riggs_frame_forward: frame is NULL");
  const int N = f->cfg.num_points;
  int rc = riggs_pose_mlp_forward(f->depth, f->width, f->multires, f->skip, f->n_rot, f->weights, f->biases, f->W_rot, f->b_rot, f->W_tr,
                                  f->b_tr, f->t, f->rot_bias4, f->sync_state, f->acts, f->local_rot, f->global_trans, s);
  if (rc) return rc;
  rc = riggs_lbs_forward_fk(N, f->num_joints, f->K, f->xyz, f->joints, f->parents, f->node_radius_log, f->local_rot, f->global_trans,
                            f->motion_mask, f->weight_mod, f->transforms, f->node_rot, f->d_nodes, f->d_xyz, f->d_rotation, nullptr, s);
  if (rc) return rc;
  RIGGS_REQUIRE(f->cfg.glue != 0, "riggs_frame_forward: the frame entry takes the RAW Gaussian parameters (cfg.glue = 1)");
  rc = riggs_raster_preprocess(&f->cfg, f->xyz, f->features_dc, f->features_rest, nullptr, f->opacity, f->scaling, f->rotation, nullptr,
                               f->d_xyz, f->d_rotation, f->d_scaling, f->geom, f->radii, f->counters, s);
  if (rc) return rc;
  return riggs_raster_render(&f->cfg, f->geom, f->binning, f->instance_capacity, f->binning_bytes, f->image_state, f->out_color,
                             f->out_depth, f->out_alpha, f->counters, s);
}

int riggs_frame_backward(const riggs_frame* f, const riggs_frame_grads* g, riggs_stream s) {
  RIGGS_REQUIRE(f != nullptr && g != nullptr, "riggs_frame_backward: NULL argument");
  const int N = f->cfg.num_points;
  // rasterizer: dL/d_xyz = dL/dd_xyz and dL/d_rotation = dL/dd_rotation (the residuals are added to the raw parameters)
  int rc = riggs_raster_backward(&f->cfg, f->xyz, f->features_dc, f->features_rest, nullptr, f->opacity, f->scaling, f->rotation, nullptr,
                                 f->d_xyz, f->d_rotation, f->d_scaling, f->radii, f->geom, f->binning, f->instance_capacity,
                                 f->image_state, f->counters, g->dL_dcolor, g->dL_ddepth, g->dL_dalpha, g->raster_workspace, g->dL_dxyz,
                                 g->dL_dmeans2D, g->dL_dfeatures_dc, nullptr, g->dL_dopacity, g->dL_dscaling, g->dL_drotation, nullptr,
                                 g->dL_dd_scaling, g->dL_dfeatures_rest, s);
  if (rc) return rc;
  rc = riggs_lbs_backward(N, f->num_joints, f->K, f->xyz, f->joints, f->parents, f->node_radius_log, f->transforms, f->node_rot,
                          f->global_trans, f->motion_mask, f->weight_mod, g->dL_dxyz, g->dL_drotation, g->dL_dtransforms,
                          g->dL_dnode_radius_log, g->dL_dglobal_trans_skinning, g->dL_dmotion_mask, g->dL_dweight_mod, g->lbs_workspace, s);
  if (rc) return rc;
  if (g->g_global_trans) hipLaunchKernelGGL(frame_add3_kernel, dim3(1), dim3(3), 0, (hipStream_t)s, g->dL_dglobal_trans_skinning, g->g_global_trans);
  return riggs_pose_mlp_backward_fk(f->depth, f->width, f->multires, f->skip, f->n_rot, f->weights, f->biases, f->W_rot, f->b_rot, f->W_tr,
                                    f->b_tr, f->acts, f->num_joints, f->local_rot, f->joints, f->parents, f->transforms, g->dL_dtransforms,
                                    g->dL_dd_nodes, g->g_local_rot, g->dL_dglobal_trans_skinning, g->dL_dlocal_rot, g->dL_dglobal_trans,
                                    nullptr, nullptr, g->pose_workspace, g->pose_flat_grads, f->sync_state, s);
}

}  // extern "C"
