"""A/B micro-benchmark: BatchNorm backward as reduce + sums + apply (three launches) vs the resident one-launch form, at the
layer shapes of a batch-256 DCGAN step (development / profiling aid).  usage: python tools/mb_bn.py [B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "jck-generation_amd")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import gpu_util as G
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G.lib.jck_tune(b"bn_res", 2)
ws = torch.zeros(G.lib.jck_grid_sync_bytes() // 4, dtype=torch.int32, device="cuda")
flush = torch.empty(300 << 20, dtype=torch.uint8, device="cuda")
for name, hw, c in (("L1", 32, 64), ("L2", 16, 128), ("L3", 8, 256), ("L4", 4, 512)):
    for groups in (1, 3):
        rows = B * hw * hw
        y = (torch.randn(groups, rows, c, device="cuda") * 1.5 + 0.3).to(torch.bfloat16)
        ga = torch.randn(groups, rows, c, device="cuda").to(torch.bfloat16)
        aux = torch.cat([torch.ones(groups, c), torch.zeros(groups, c), torch.full((groups, c), 0.3), torch.full((groups, c), 0.66)], 1).cuda().contiguous()
        sums = torch.zeros(groups * G.lib.jck_bn_bwd_ws_floats(c), device="cuda")
        gy = torch.empty_like(ga)
        dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        fns = {"3-launch": lambda: G.lib.jck_bn_act_bwd_grouped(0, ga, y, aux, 0.2, sums, gy, dg, db, rows, c, groups, groups, G.cur_stream()),
               "resident": lambda: G.lib.jck_bn_act_bwd_res(0, ga, y, aux, 0.2, sums, gy, dg, db, rows, c, groups, groups, ws, G.cur_stream())}
        res = {k: [] for k in fns}
        for rnd in range(5):
            for k, fn in fns.items():
                fn(); torch.cuda.synchronize()
                ts = []
                for _ in range(6):
                    flush.fill_(1)                   # cold caches: in the step the tensors were written a few launches earlier
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                res[k].append(sorted(ts)[len(ts) // 2])
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        mb = groups * rows * c * 2 / 1e6
        print(f"{name} C={c} rows={rows} groups={groups} ({mb:.1f} MB/tensor): 3-launch {med['3-launch']:.1f} us | resident {med['resident']:.1f} us "
              f"= {3 * mb / med['resident'] / 1e3 * 1e0:.2f} TB/s algorithmic | speedup {med['3-launch'] / med['resident']:.2f}", flush=True)
print("grid sync error word:", G.lib.jck_grid_sync_error(ws))
