import sys, torch, ctypes as C
sys.path.insert(0, "/root/repo")
from riggs_amd import _lib as L, synth
from riggs_amd.skeleton import SkeletonWarp
lib = L.lib()
for J in (24, 64):
    sc = synth.make_scene(300000, J, 7)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8, use_skinning_weight_mlp=False, use_template_offsets=False).cuda()
    sw._node_radius.data = sc["node_radius"].cuda()
    x = sc["xyz"].cuda(); t = torch.tensor(0.41, device="cuda")
    w = torch.randn(x.shape[0], 3, device="cuda")
    for fused in (False, True):
        def step():
            for p in sw.parameters(): p.grad = None
            dv = sw(x, t, None) if fused else sw.deform_by_pose(x, sw.get_pose_info(sw.expand_time(t)), None)
            ((dv["d_xyz"] * w).sum() + dv["d_nodes"].sum()).backward()
        for _ in range(5): step()
        torch.cuda.synchronize()
        lib.riggs_prof_reset(); lib.riggs_prof_enable(0xFFFFFFFF)
        for _ in range(30): step()
        torch.cuda.synchronize(); lib.riggs_prof_enable(0)
        tot, cnt = C.c_float(), C.c_int32(); out = {}
        lib.riggs_prof_name.restype = C.c_char_p
        for i in range(lib.riggs_prof_count()):
            L.check(lib.riggs_prof_read(i, C.byref(tot), C.byref(cnt)), "r")
            if cnt.value: out[lib.riggs_prof_name(i).decode()] = round(1e3 * tot.value / cnt.value, 1)
        print("J", J, "fused" if fused else "separate", out, "sum", round(sum(out.values()), 1))
