"""In-kernel stamps of the resident BatchNorm backward (development aid): where a launch spends its time."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "jck-generation_amd")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import gpu_util as G
B = 256
ws = torch.zeros(G.lib.jck_grid_sync_bytes() // 4, dtype=torch.int32, device="cuda")
st = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
G.lib.jck_debug_bnres_stamps(st)
flush = torch.empty(300 << 20, dtype=torch.uint8, device="cuda")
G.lib.jck_tune(b"bn_res", 2)
GROUPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for name, hw, c in (("L1", 32, 64), ("L2", 16, 128), ("L3", 8, 256), ("L4", 4, 512)):
    rows, groups = B * hw * hw, GROUPS
    y = (torch.randn(groups, rows, c, device="cuda") * 1.5 + 0.3).to(torch.bfloat16)
    ga = torch.randn(groups, rows, c, device="cuda").to(torch.bfloat16)
    aux = torch.cat([torch.ones(groups, c), torch.zeros(groups, c), torch.full((groups, c), 0.3), torch.full((groups, c), 0.66)], 1).cuda().contiguous()
    sums = torch.zeros(groups * G.lib.jck_bn_bwd_ws_floats(c), device="cuda")
    gy = torch.empty_like(ga)
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    for cold in (1, 0):
        for it in range(3):
            if cold: flush.fill_(1)
            torch.cuda.synchronize()
            G.lib.jck_bn_act_bwd_res(0, ga, y, aux, 0.2, sums, gy, dg, db, rows, c, groups, groups, ws, G.cur_stream())
            torch.cuda.synchronize()
        s = st.view(256, 8).cpu().double() / 100.0          # us
        t0 = s[:, 0].min()
        names = ["start", "phase1 done", "row stored", "barrier passed", "coefs ready", "stores drained"]
        print(f"{name} cold={cold}: " + " | ".join(f"{n}: med {float((s[:, i] - t0).median()):.1f} min {float((s[:, i] - t0).min()):.1f} max {float((s[:, i] - t0).max()):.1f}" for i, n in enumerate(names)), flush=True)
