"""A/B micro-benchmark of single conv products in ONE process (jck_tune switches the variant between interleaved rounds).
usage: python tools/mb2.py <knob> <valA> <valB> <shape>[,<shape>...] [B ...]      (development / profiling aid)"""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "jck-generation_amd")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import gpu_util as G
knob, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
shapes = sys.argv[4].split(",")
Bs = [int(x) for x in sys.argv[5:]] or [256]
CFG = {"down2": ("down", 32, 64, 128), "down3": ("down", 16, 128, 256), "down4": ("down", 8, 256, 512),
       "up2": ("up", 4, 512, 256), "up3": ("up", 8, 256, 128), "up4": ("up", 16, 128, 64),
       "wg2": ("wg", 32, 64, 128), "wg3": ("wg", 16, 128, 256), "wg4": ("wg", 8, 256, 512), "wg1": ("wg", 64, 64, 128), "down1": ("down", 64, 3, 64)}
prec = 0
def make(which, B):
    cfg = CFG[which]; kind = cfg[0]
    torch.manual_seed(0)
    if kind == "down":
        _, hb, cb, cs = cfg
        x = torch.randn(B, hb, hb, 4 if cb == 3 else cb, device="cuda").to(torch.bfloat16)
        if cb == 3:
            x[..., 3] = 0
        w = G.pack_down(torch.randn(cs, cb, 4, 4) * 0.05, prec)
        out = torch.empty(B, hb // 2, hb // 2, cs, dtype=torch.bfloat16, device="cuda")
        stats, slots = G.stats_buf(B * (hb // 2) ** 2, cs)
        fn = lambda: G.lib.jck_conv_down(prec, x, w, out, stats, ctypes.byref(slots), B, hb, hb, cb, cs, G.cur_stream())
        return fn, 2 * B * (hb // 2) ** 2 * cs * 16 * cb, out, lambda: out.fill_(float('nan'))
    if kind == "up":
        _, hs, cs, cb = cfg
        x = torch.randn(B, hs, hs, cs, device="cuda").to(torch.bfloat16)
        w = G.pack_up(torch.randn(cs, cb, 4, 4) * 0.05, prec)
        out = torch.empty(B, 2 * hs, 2 * hs, cb, dtype=torch.bfloat16, device="cuda")
        stats, slots = G.stats_buf(B * 4 * hs * hs, cb)
        fn = lambda: G.lib.jck_conv_up(prec, x, w, out, stats, ctypes.byref(slots), 0, B, hs, hs, cs, cb, G.cur_stream())
        return fn, 2 * B * hs * hs * 4 * cb * 4 * cs, out, lambda: out.fill_(float('nan'))
    _, hb, cb, cs = cfg
    big = torch.randn(B, hb, hb, cb, device="cuda").to(torch.bfloat16)
    small = torch.randn(B, hb // 2, hb // 2, cs, device="cuda").to(torch.bfloat16)
    nb = G.lib.jck_conv_wgrad_ws_bytes(B, hb, hb, cb, cs)
    ws = torch.empty(nb // 4, device="cuda")
    grad = torch.zeros(cs, cb, 4, 4, device="cuda")
    fn = lambda: G.lib.jck_conv_wgrad(prec, small, big, ws, nb, grad, 0, B, hb, hb, cb, cs, G.cur_stream())
    return fn, 2 * B * (hb // 2) ** 2 * cs * 16 * cb, grad, lambda: (ws.fill_(float('nan')), grad.fill_(float('nan')))
for which in shapes:
    for B in Bs:
        fn, flops, out, poison = make(which, B)
        res = {va: [], vb: []}
        outs = {}
        for rnd in range(5):
            for v in (va, vb):
                G.lib.jck_tune(knob.encode(), v)
                poison()
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): fn()
                e1.record(); torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / 10)
                outs[v] = out.float().clone()
        med = {v: sorted(r)[len(r) // 2] for v, r in res.items()}
        d = (outs[va] - outs[vb]).abs().max().item() / (outs[va].abs().max().item() + 1e-30)
        print(f"{which} B={B}: {knob}={va}: {med[va]*1e3:.1f} us {flops/med[va]/1e9:.0f} TF | {knob}={vb}: {med[vb]*1e3:.1f} us {flops/med[vb]/1e9:.0f} TF | "
              f"speedup {med[va]/med[vb]:.3f} | max rel diff of results {d:.2e}", flush=True)
