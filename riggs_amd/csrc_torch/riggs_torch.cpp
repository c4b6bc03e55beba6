// PyTorch-ROCm front-end of libriggs_hip.so (north_star: "exposed ... via a PyTorch-ROCm C++/HIP extension").
//
// The two autograd nodes an unmodified train_rig.py runs per iteration — SkeletonWarp.forward (train_rig.py:411 -> skeleton
// step) and render() (gaussian_renderer/__init__.py:133-141) — with the node, the argument marshalling and the output
// allocations in C++: as Python autograd.Functions over ctypes (riggs_amd/skeleton.py: _PoseDeform, riggs_amd/render.py:
// _FusedGlueRaster — which stay, as the fallback and as what the C-ABI tests exercise) an eagerly issued frame is bound by the
// host, its backward by the hop of every Python node through the autograd engine's device thread.  Nothing is computed here:
// every launch is a call of the C ABI (include/riggs_hip.h) on torch's current stream, on tensors torch owns.
//
//   torch.ops.riggs.pose_deform(...)  = riggs_pose_mlp_forward -> riggs_lbs_forward_fk | riggs_lbs_backward -> riggs_pose_mlp_backward_fk
//   torch.ops.riggs.glue_raster(...)  = riggs_raster_preprocess -> riggs_raster_render  | riggs_raster_backward
//
// The arena policy (how large the instance arena is, the asynchronous read-back of the instance count, overflow reporting) stays
// with riggs_amd.rasterizer.RasterArena: the op is handed the arena and returns the frame's counters.
#include <torch/autograd.h>
#include <torch/library.h>

#include <c10/hip/HIPStream.h>

#include <cstring>
#include <vector>

#include "../../include/riggs_hip.h"

namespace {

using at::Tensor;
using c10::optional;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

inline void* ptr(const Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
inline void* ptr(const optional<Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr; }

inline void check(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed (rc=", rc, "): ", riggs_last_error());
}

inline riggs_stream current_stream(const Tensor& t) {
  return (riggs_stream)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

// the product path is GPU-only and fp32: the same refusals as riggs_amd._lib.require_cuda_f32
inline Tensor f32(const char* name, const Tensor& t) {
  TORCH_CHECK(t.is_cuda(), name, " must be a CUDA(HIP) tensor — the product path is GPU-only");
  TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32, got ", t.scalar_type());
  return t.is_contiguous() ? t : t.contiguous();
}
inline Tensor f32(const char* name, const optional<Tensor>& t) { return (t.has_value() && t->defined()) ? f32(name, *t) : Tensor(); }
inline Tensor opt(const optional<Tensor>& t) { return (t.has_value() && t->defined()) ? *t : Tensor(); }

// ------------------------------------------------------------------------------------------------------------------
// SkeletonWarp.forward(x, t, motion_mask) as one node (riggs_amd/skeleton.py: _PoseDeform is the Python twin, argument for
// argument): PoseMLP (utils/time_utils.py:208-256, skeleton_utils/network_utils.py:115-150), forward kinematics
// (skeleton_warp.py:242-273) + skinning (:41-76, 130-172) forward; skinning backward, reverse chain sweep + PoseMLP backward.
// ------------------------------------------------------------------------------------------------------------------
struct PoseDeformFn : public torch::autograd::Function<PoseDeformFn> {
  // inputs: t, rot_bias?, sync?, rho, mask?, x, joints, parents, weight_mod?, fixed_coef?, fixed_loss?, bone_table?, params...
  static variable_list forward(AutogradContext* ctx, const Tensor& t_in, const optional<Tensor>& rot_bias, const optional<Tensor>& sync,
                               const Tensor& rho_in, const optional<Tensor>& mask, const Tensor& x_in, const Tensor& joints,
                               const Tensor& parents, const optional<Tensor>& weight_mod_in, const optional<Tensor>& fixed_coef,
                               const optional<Tensor>& fixed_loss, const optional<Tensor>& bone_table, int64_t K, int64_t depth,
                               int64_t width, int64_t multires, int64_t skip, at::TensorList params_in) {
    ctx->set_materialize_grads(false);
    TORCH_CHECK((int64_t)params_in.size() == 2 * depth + 4, "pose_deform: 2 * depth + 4 PoseMLP parameters expected");
    std::vector<Tensor> params;
    params.reserve(params_in.size());
    for (const auto& p : params_in) params.push_back(f32("PoseMLP parameter", p));
    const Tensor x = f32("x", x_in), rho = f32("_node_radius", rho_in), t = f32("t", t_in);
    const int64_t N = x.size(0), J = joints.size(0);
    const int64_t n_rot = params[2 * depth].size(0);
    Tensor weight_mod = f32("skinning weight offsets", weight_mod_in);
    if (weight_mod.defined()) TORCH_CHECK(weight_mod.dim() == 2 && weight_mod.size(0) == N && weight_mod.size(1) == J - 1, "skinning weight offsets must be (N, J-1)");
    Tensor mflat;
    if (mask.has_value() && mask->defined()) {
      mflat = f32("motion_mask", mask->reshape({-1}));
      TORCH_CHECK(mflat.numel() == N, "motion_mask must hold one value per point");
    }
    const auto fo = x.options();
    const int64_t n_acts = (int64_t)riggs_pose_mlp_acts_floats((int32_t)depth, (int32_t)width, (int32_t)multires);
    const int64_t o_small = (n_acts + 63) & ~(int64_t)63;
    // (two allocations for the eight outputs / saved arrays, every piece on a 256-byte boundary — as the Python twin)
    Tensor sbuf = at::empty({o_small + J * 23 + 4}, fo);
    Tensor acts = sbuf.narrow(0, 0, n_acts), small = sbuf.narrow(0, o_small, J * 23 + 4);
    Tensor local_rot = small.narrow(0, 0, J * 4).view({J, 4}), transforms = small.narrow(0, J * 4, J * 12).view({J, 12});
    Tensor node_rot = small.narrow(0, J * 16, J * 4).view({J, 4}), d_nodes = small.narrow(0, J * 20, J * 3).view({J, 3});
    Tensor global_trans = small.narrow(0, J * 23, 3);
    const int64_t o_rot = (3 * N + 63) & ~(int64_t)63;
    Tensor dbuf = at::empty({o_rot + 4 * N}, fo);
    Tensor d_xyz = dbuf.narrow(0, 0, 3 * N).view({N, 3}), d_rot = dbuf.narrow(0, o_rot, 4 * N).view({N, 4});
    std::vector<const float*> Wp(depth), bp(depth);
    for (int64_t l = 0; l < depth; l++) { Wp[l] = params[2 * l].data_ptr<float>(); bp[l] = params[2 * l + 1].data_ptr<float>(); }
    const Tensor* h = &params[2 * depth];
    riggs_stream st = current_stream(x);
    check(riggs_pose_mlp_forward((int32_t)depth, (int32_t)width, (int32_t)multires, (int32_t)skip, (int32_t)n_rot, Wp.data(), bp.data(),
                                 h[0].data_ptr<float>(), h[1].data_ptr<float>(), h[2].data_ptr<float>(), h[3].data_ptr<float>(),
                                 t.data_ptr<float>(), (const float*)ptr(rot_bias), ptr(sync), acts.data_ptr<float>(),
                                 local_rot.data_ptr<float>(), global_trans.data_ptr<float>(), st),
          "riggs_pose_mlp_forward");
    check(riggs_lbs_forward_fk((int32_t)N, (int32_t)J, (int32_t)K, x.data_ptr<float>(), joints.data_ptr<float>(), parents.data_ptr<int32_t>(),
                               rho.data_ptr<float>(), local_rot.data_ptr<float>(), global_trans.data_ptr<float>(), (const float*)ptr(mflat),
                               (const float*)ptr(weight_mod), transforms.data_ptr<float>(), node_rot.data_ptr<float>(),
                               d_nodes.data_ptr<float>(), d_xyz.data_ptr<float>(), d_rot.data_ptr<float>(), ptr(bone_table), st),
          "riggs_lbs_forward_fk");
    variable_list saved = {acts, local_rot, global_trans, rho, mflat, x, joints, parents, transforms, node_rot, weight_mod,
                           opt(fixed_coef), opt(fixed_loss), opt(sync)};
    for (auto& p : params) saved.push_back(p);
    ctx->save_for_backward(saved);
    ctx->saved_data["cfg"] = std::vector<int64_t>{depth, width, multires, skip, n_rot, K};
    ctx->saved_data["need_mask"] = mask.has_value() && mask->defined() && mask->requires_grad();
    ctx->saved_data["mask_shape"] = (mask.has_value() && mask->defined()) ? mask->sizes().vec() : std::vector<int64_t>{};
    ctx->mark_non_differentiable({node_rot});
    return {d_xyz, d_rot, d_nodes, local_rot, global_trans, transforms, node_rot};
  }

  static variable_list backward(AutogradContext* ctx, variable_list g) {
    const auto s = ctx->get_saved_variables();
    const Tensor &acts = s[0], &local_rot = s[1], &global_trans = s[2], &rho = s[3], &mflat = s[4], &x = s[5], &joints = s[6],
                 &parents = s[7], &transforms = s[8], &node_rot = s[9], &weight_mod = s[10], &fixed_coef = s[11], &fixed_loss = s[12],
                 &sync = s[13];
    const auto cfg = ctx->saved_data["cfg"].toIntVector();
    const int64_t depth = cfg[0], width = cfg[1], multires = cfg[2], skip = cfg[3], n_rot = cfg[4], K = cfg[5];
    std::vector<Tensor> params(s.begin() + 14, s.end());
    const int64_t N = x.size(0), J = joints.size(0);
    const auto fo = x.options();
    Tensor g_xyz = g[0].defined() ? g[0].contiguous() : at::zeros({N, 3}, fo);
    Tensor g_rot = g[1].defined() ? g[1].contiguous() : at::zeros({N, 4}, fo);
    Tensor bsmall = at::empty({J * 16 + 12}, fo);  // (dG | dq | dgt | dgt_total: one allocation)
    Tensor dG = bsmall.narrow(0, 0, J * 12).view({J, 12}), dq = bsmall.narrow(0, J * 12, J * 4).view({J, 4});
    Tensor dgt = bsmall.narrow(0, J * 16, 3), dgt_total = bsmall.narrow(0, J * 16 + 4, 3);
    Tensor drho = at::empty({J}, fo);
    const bool need_mask = mflat.defined() && ctx->saved_data["need_mask"].toBool();
    Tensor dmask = need_mask ? at::empty({N}, fo) : Tensor();
    Tensor dmod = weight_mod.defined() ? at::empty({N, J - 1}, fo) : Tensor();
    riggs_stream st = current_stream(x);
    Tensor ws = at::empty({(int64_t)riggs_lbs_backward_workspace_bytes((int32_t)N, (int32_t)J)}, fo.dtype(at::kByte));
    check(riggs_lbs_backward((int32_t)N, (int32_t)J, (int32_t)K, x.data_ptr<float>(), joints.data_ptr<float>(), parents.data_ptr<int32_t>(),
                             rho.data_ptr<float>(), transforms.data_ptr<float>(), node_rot.data_ptr<float>(), global_trans.data_ptr<float>(),
                             (const float*)ptr(mflat), (const float*)ptr(weight_mod), g_xyz.data_ptr<float>(), g_rot.data_ptr<float>(),
                             dG.data_ptr<float>(), drho.data_ptr<float>(), dgt.data_ptr<float>(), (float*)ptr(dmask), (float*)ptr(dmod),
                             ws.data_ptr(), st),
          "riggs_lbs_backward");
    if (g[5].defined()) dG = at::add(dG, g[5]);
    if (g[4].defined()) dgt = at::add(dgt, g[4].reshape({-1}));
    Tensor gn = g[2].defined() ? g[2].contiguous() : Tensor();
    Tensor gq = g[3].defined() ? g[3].contiguous() : Tensor();
    int64_t total = 0;
    for (auto& p : params) total += p.numel();
    Tensor flat = at::empty({total}, fo);
    Tensor dzs = at::empty({(int64_t)riggs_pose_mlp_backward_workspace_floats((int32_t)depth, (int32_t)width, (int32_t)multires)}, fo);
    std::vector<const float*> Wp(depth), bp(depth);
    for (int64_t l = 0; l < depth; l++) { Wp[l] = params[2 * l].data_ptr<float>(); bp[l] = params[2 * l + 1].data_ptr<float>(); }
    const Tensor* h = &params[2 * depth];
    check(riggs_pose_mlp_backward_fk((int32_t)depth, (int32_t)width, (int32_t)multires, (int32_t)skip, (int32_t)n_rot, Wp.data(), bp.data(),
                                     h[0].data_ptr<float>(), h[1].data_ptr<float>(), h[2].data_ptr<float>(), h[3].data_ptr<float>(),
                                     acts.data_ptr<float>(), (int32_t)J, local_rot.data_ptr<float>(), joints.data_ptr<float>(),
                                     parents.data_ptr<int32_t>(), transforms.data_ptr<float>(), dG.data_ptr<float>(), (const float*)ptr(gn),
                                     (const float*)ptr(gq), dgt.data_ptr<float>(), dq.data_ptr<float>(), dgt_total.data_ptr<float>(),
                                     (const float*)ptr(fixed_coef), (float*)ptr(fixed_loss), dzs.data_ptr<float>(), flat.data_ptr<float>(),
                                     ptr(sync), st),
          "riggs_pose_mlp_backward_fk");
    // inputs: t, rot_bias, sync, rho, mask, x, joints, parents, weight_mod, fixed_coef, fixed_loss, bone_table, K, depth, width,
    // multires, skip, params...
    variable_list out(17 + params.size());
    out[3] = drho;
    if (need_mask) out[4] = dmask.reshape(ctx->saved_data["mask_shape"].toIntVector());
    out[8] = dmod;
    int64_t off = 0;
    for (size_t i = 0; i < params.size(); i++) {
      const int64_t n = params[i].numel();
      Tensor gi = flat.narrow(0, off, n);
      out[17 + i] = params[i].dim() == 1 ? gi : gi.view(params[i].sizes());
      off += n;
    }
    return out;
  }
};

std::vector<Tensor> pose_deform(const Tensor& t, const optional<Tensor>& rot_bias, const optional<Tensor>& sync, const Tensor& rho,
                                const optional<Tensor>& mask, const Tensor& x, const Tensor& joints, const Tensor& parents,
                                const optional<Tensor>& weight_mod, const optional<Tensor>& fixed_coef, const optional<Tensor>& fixed_loss,
                                const optional<Tensor>& bone_table, int64_t K, int64_t depth, int64_t width, int64_t multires,
                                int64_t skip, std::vector<Tensor> params) {
  return PoseDeformFn::apply(t, rot_bias, sync, rho, mask, x, joints, parents, weight_mod, fixed_coef, fixed_loss, bone_table, K, depth,
                             width, multires, skip, at::TensorList(params));
}

// ------------------------------------------------------------------------------------------------------------------
// render()'s default branch as one node over the RAW Gaussian parameters (riggs_amd/render.py: _FusedGlueRaster is the Python twin):
// render glue (gaussian_renderer/__init__.py:74-92) + rasterizer forward (:133-141) | both backwards.
// ------------------------------------------------------------------------------------------------------------------
struct RasterCfgHost {
  riggs_raster_cfg c;
};

riggs_raster_cfg make_cfg(int64_t N, int64_t M, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier, int64_t sh_degree,
                          const Tensor& bg, const Tensor& view, const Tensor& proj, const Tensor& campos, bool debug, bool isotropic,
                          bool tight_lists) {
  riggs_raster_cfg c;
  std::memset(&c, 0, sizeof(c));
  c.num_points = (int32_t)N; c.sh_degree = (int32_t)sh_degree; c.sh_coeffs = (int32_t)M;
  c.image_height = (int32_t)H; c.image_width = (int32_t)W;
  c.tanfovx = (float)tanfovx; c.tanfovy = (float)tanfovy; c.scale_modifier = (float)scale_modifier;
  c.bg = bg.data_ptr<float>(); c.viewmatrix = view.data_ptr<float>(); c.projmatrix = proj.data_ptr<float>(); c.campos = campos.data_ptr<float>();
  c.debug = debug ? 1 : 0; c.glue = 1; c.isotropic = isotropic ? 1 : 0; c.deterministic = 0; c.sparse_zero = 0;
  c.tight_lists = tight_lists ? 1 : 0;
  return c;
}

struct GlueRasterFn : public torch::autograd::Function<GlueRasterFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& xyz_in, const Tensor& means2D, const Tensor& f_dc_in,
                               const Tensor& f_rest_in, const Tensor& opacity_in, const Tensor& scaling_in, const Tensor& rotation_in,
                               const optional<Tensor>& d_xyz_in, const optional<Tensor>& d_rot_in, const optional<Tensor>& d_scaling_in,
                               const Tensor& bg_in, const Tensor& view_in, const Tensor& proj_in, const Tensor& campos_in,
                               const Tensor& binning, const Tensor& workspace, int64_t cap, int64_t H, int64_t W, double tanfovx,
                               double tanfovy, double scale_modifier, int64_t sh_degree, bool debug, bool isotropic, bool tight_lists) {
    ctx->set_materialize_grads(false);
    const Tensor xyz = f32("_xyz", xyz_in), f_dc = f32("_features_dc", f_dc_in), f_rest = f32("_features_rest", f_rest_in);
    const Tensor opacity = f32("_opacity", opacity_in), scaling = f32("_scaling", scaling_in), rotation = f32("_rotation", rotation_in);
    const Tensor d_xyz = f32("d_xyz", d_xyz_in), d_rot = f32("d_rotation", d_rot_in), d_scaling = f32("d_scaling", d_scaling_in);
    const int64_t N = xyz.size(0);
    TORCH_CHECK(xyz.dim() == 2 && xyz.size(1) == 3, "_xyz must be (N, 3)");
    TORCH_CHECK(f_dc.dim() == 3 && f_dc.size(0) == N && f_dc.size(1) == 1 && f_dc.size(2) == 3, "_features_dc must be (N, 1, 3)");
    TORCH_CHECK(f_rest.dim() == 3 && f_rest.size(0) == N && f_rest.size(2) == 3, "_features_rest must be (N, M - 1, 3)");
    TORCH_CHECK(opacity.numel() == N && rotation.numel() == 4 * N && scaling.numel() == (isotropic ? N : 3 * N), "Gaussian parameter shapes");
    TORCH_CHECK(!d_xyz.defined() || d_xyz.numel() == 3 * N, "d_xyz must be (N, 3)");
    TORCH_CHECK(!d_rot.defined() || d_rot.numel() == 4 * N, "d_rotation must be (N, 4)");
    TORCH_CHECK(!d_scaling.defined() || d_scaling.numel() == 3 * N, "d_scaling must be (N, 3)");
    const int64_t M = 1 + f_rest.size(1);
    const Tensor bg = f32("bg", bg_in.reshape({-1})), view = f32("viewmatrix", view_in), proj = f32("projmatrix", proj_in);
    const Tensor campos = f32("campos", campos_in.reshape({-1}));
    TORCH_CHECK(bg.numel() == 3 && view.numel() == 16 && proj.numel() == 16 && campos.numel() == 3, "camera tensors");
    riggs_raster_cfg cfg = make_cfg(N, M, H, W, tanfovx, tanfovy, scale_modifier, sh_degree, bg, view, proj, campos, debug, isotropic, tight_lists);
    const auto fo = xyz.options();
    const auto bo = fo.dtype(at::kByte), io = fo.dtype(at::kInt);
    Tensor geom = at::empty({(int64_t)riggs_raster_geom_bytes((int32_t)N)}, bo);
    Tensor img = at::empty({(int64_t)riggs_raster_image_bytes((int32_t)H, (int32_t)W)}, bo);
    Tensor radii = at::empty({N}, io), counters = at::empty({4}, io);
    Tensor color = at::empty({3, H, W}, fo), depth = at::empty({1, H, W}, fo), alpha = at::empty({1, H, W}, fo);
    riggs_stream st = current_stream(xyz);
    check(riggs_raster_preprocess(&cfg, xyz.data_ptr<float>(), f_dc.data_ptr<float>(), f_rest.data_ptr<float>(), nullptr,
                                  opacity.data_ptr<float>(), scaling.data_ptr<float>(), rotation.data_ptr<float>(), nullptr,
                                  (const float*)ptr(d_xyz), (const float*)ptr(d_rot), (const float*)ptr(d_scaling), geom.data_ptr(),
                                  radii.data_ptr<int32_t>(), (uint32_t*)counters.data_ptr<int32_t>(), st),
          "riggs_raster_preprocess");
    check(riggs_raster_render(&cfg, geom.data_ptr(), binning.data_ptr(), cap, (size_t)binning.numel(), img.data_ptr(), color.data_ptr<float>(),
                              depth.data_ptr<float>(), alpha.data_ptr<float>(), (uint32_t*)counters.data_ptr<int32_t>(), st),
          "riggs_raster_render");
    ctx->save_for_backward({xyz, f_dc, f_rest, opacity, scaling, rotation, d_xyz, d_rot, d_scaling, bg, view, proj, campos, geom, img, binning,
                            radii, counters, workspace});
    ctx->saved_data["i"] = std::vector<int64_t>{N, M, H, W, sh_degree, cap, debug ? 1 : 0, isotropic ? 1 : 0, tight_lists ? 1 : 0,
                                                (d_scaling_in.has_value() && d_scaling_in->defined() && d_scaling_in->requires_grad()) ? 1 : 0};
    ctx->saved_data["f"] = std::vector<double>{tanfovx, tanfovy, scale_modifier};
    ctx->mark_non_differentiable({radii, counters});
    return {color, radii, depth, alpha, counters};
  }

  static variable_list backward(AutogradContext* ctx, variable_list g) {
    const auto s = ctx->get_saved_variables();
    const Tensor &xyz = s[0], &f_dc = s[1], &f_rest = s[2], &opacity = s[3], &scaling = s[4], &rotation = s[5], &d_xyz = s[6], &d_rot = s[7],
                 &d_scaling = s[8], &bg = s[9], &view = s[10], &proj = s[11], &campos = s[12], &geom = s[13], &img = s[14], &binning = s[15],
                 &radii = s[16], &counters = s[17], &ws = s[18];
    const auto iv = ctx->saved_data["i"].toIntVector();
    const auto fv = ctx->saved_data["f"].toDoubleVector();
    const int64_t N = iv[0], M = iv[1], H = iv[2], W = iv[3], cap = iv[5];
    const bool iso = iv[7] != 0, need_ds = d_scaling.defined() && iv[9] != 0;
    riggs_raster_cfg cfg = make_cfg(N, M, H, W, fv[0], fv[1], fv[2], iv[4], bg, view, proj, campos, iv[6] != 0, iso, iv[8] != 0);
    const auto fo = xyz.options();
    Tensor g_means3D = at::empty({N, 3}, fo), g_means2D = at::empty({N, 3}, fo), g_dc = at::empty({N, 1, 3}, fo);
    Tensor g_rest = at::empty({N, M - 1, 3}, fo), g_opac = at::empty({N, 1}, fo), g_scales = at::empty({N, iso ? 1 : 3}, fo);
    Tensor g_rots = at::empty({N, 4}, fo);
    Tensor g_ds = need_ds ? at::empty({N, 3}, fo) : Tensor();
    Tensor gc = g[0].defined() ? f32("grad_color", g[0]) : at::zeros({3, H, W}, fo);  // (a loss on depth / alpha only)
    Tensor gd = g[2].defined() ? f32("grad_depth", g[2]) : Tensor();
    Tensor ga = g[3].defined() ? f32("grad_alpha", g[3]) : Tensor();
    riggs_stream st = current_stream(xyz);
    check(riggs_raster_backward(&cfg, xyz.data_ptr<float>(), f_dc.data_ptr<float>(), f_rest.data_ptr<float>(), nullptr, opacity.data_ptr<float>(),
                                scaling.data_ptr<float>(), rotation.data_ptr<float>(), nullptr, (const float*)ptr(d_xyz), (const float*)ptr(d_rot),
                                (const float*)ptr(d_scaling), radii.data_ptr<int32_t>(), geom.data_ptr(), binning.data_ptr(), cap, img.data_ptr(),
                                (const uint32_t*)counters.data_ptr<int32_t>(), gc.data_ptr<float>(), (const float*)ptr(gd), (const float*)ptr(ga),
                                ws.data_ptr(), g_means3D.data_ptr<float>(), g_means2D.data_ptr<float>(), g_dc.data_ptr<float>(), nullptr,
                                g_opac.data_ptr<float>(), g_scales.data_ptr<float>(), g_rots.data_ptr<float>(), nullptr, (float*)ptr(g_ds),
                                g_rest.data_ptr<float>(), st),
          "riggs_raster_backward");
    // dL/d(d_xyz) == dL/dxyz and dL/d(d_rotation) == dL/d_rotation: the residual branches get an alias (a second tensor object
    // on the same storage) so that AccumulateGrad adopts the parameter gradients instead of cloning them
    variable_list out(26);
    out[0] = g_means3D; out[1] = g_means2D; out[2] = g_dc; out[3] = g_rest; out[4] = g_opac; out[5] = g_scales; out[6] = g_rots;
    if (d_xyz.defined()) out[7] = g_means3D.detach();
    if (d_rot.defined()) out[8] = g_rots.detach();
    out[9] = g_ds;
    return out;
  }
};

std::vector<Tensor> glue_raster(const Tensor& xyz, const Tensor& means2D, const Tensor& f_dc, const Tensor& f_rest, const Tensor& opacity,
                                const Tensor& scaling, const Tensor& rotation, const optional<Tensor>& d_xyz, const optional<Tensor>& d_rot,
                                const optional<Tensor>& d_scaling, const Tensor& bg, const Tensor& view, const Tensor& proj,
                                const Tensor& campos, const Tensor& binning, const Tensor& workspace, int64_t cap, int64_t H, int64_t W,
                                double tanfovx, double tanfovy, double scale_modifier, int64_t sh_degree, bool debug, bool isotropic,
                                bool tight_lists) {
  return GlueRasterFn::apply(xyz, means2D, f_dc, f_rest, opacity, scaling, rotation, d_xyz, d_rot, d_scaling, bg, view, proj, campos, binning,
                             workspace, cap, H, W, tanfovx, tanfovy, scale_modifier, sh_degree, debug, isotropic, tight_lists);
}

int64_t abi_version() { return (int64_t)riggs_version(); }

}  // namespace

TORCH_LIBRARY(riggs, m) {
  m.def("pose_deform", &pose_deform);
  m.def("glue_raster", &glue_raster);
  m.def("abi_version", &abi_version);
}
