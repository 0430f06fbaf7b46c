"""Micro-benchmark of single conv products at the B=256 shapes (development / profiling aid)."""
import ctypes, sys, os, time
sys.path.insert(0, '/root/repo/jck-generation_amd'); sys.path.insert(0, '/root/repo/tests')
import torch
import gpu_util as G
which = sys.argv[1] if len(sys.argv) > 1 else "down3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prec = 0
B = int(os.environ.get("MB_B", "256"))
cfg = {"down1": ("down", 64, 3, 64), "up1": ("up", 32, 64, 3), "wg1": ("wg", 64, 3, 64), "down2": ("down", 32, 64, 128), "down3": ("down", 16, 128, 256), "down4": ("down", 8, 256, 512),
       "up2": ("up", 4, 512, 256), "up3": ("up", 8, 256, 128), "up4": ("up", 16, 128, 64),
       "wg2": ("wg", 32, 64, 128), "wg3": ("wg", 16, 128, 256), "wg4": ("wg", 8, 256, 512)}[which]
kind = cfg[0]
torch.manual_seed(0)
if kind == "down":
    _, hb, cb, cs = cfg
    x = torch.randn(B, hb, hb, 4 if cb == 3 else cb, device="cuda").to(torch.bfloat16)
    w = G.pack_down(torch.randn(cs, cb, 4, 4) * 0.05, prec)
    out = torch.empty(B, hb // 2, hb // 2, cs, dtype=torch.bfloat16, device="cuda")
    stats, slots = G.stats_buf(B * (hb // 2) ** 2, cs)
    fn = lambda: G.lib.jck_conv_down(prec, x, w, out, stats, ctypes.byref(slots), B, hb, hb, cb, cs, G.cur_stream())
    flops = 2 * B * (hb // 2) ** 2 * cs * 16 * cb
elif kind == "up":
    _, hs, cs, cb = cfg
    x = torch.randn(B, hs, hs, cs, device="cuda").to(torch.bfloat16)
    w = G.pack_up(torch.randn(cs, cb, 4, 4) * 0.05, prec)
    out = torch.empty(B, 2 * hs, 2 * hs, 4 if cb == 3 else cb, dtype=torch.bfloat16, device="cuda")
    stats, slots = G.stats_buf(B * 4 * hs * hs, 4 if cb == 3 else cb)
    if cb == 3:
        fn = lambda: G.lib.jck_conv_up(prec, x, w, out, None, None, 1, B, hs, hs, cs, cb, G.cur_stream())
    else:
        fn = lambda: G.lib.jck_conv_up(prec, x, w, out, stats, ctypes.byref(slots), 0, B, hs, hs, cs, cb, G.cur_stream())
    flops = 2 * B * hs * hs * 4 * cb * 4 * cs
else:
    _, hb, cb, cs = cfg
    big = torch.randn(B, hb, hb, 4 if cb == 3 else cb, device="cuda").to(torch.bfloat16)
    small = torch.randn(B, hb // 2, hb // 2, cs, device="cuda").to(torch.bfloat16)
    nb = G.lib.jck_conv_wgrad_ws_bytes(B, hb, hb, cb, cs)
    ws = torch.empty(nb // 4, device="cuda")
    grad = torch.zeros(cs, cb, 4, 4, device="cuda")
    fn = lambda: G.lib.jck_conv_wgrad(prec, small, big, ws, nb, grad, 0, B, hb, hb, cb, cs, G.cur_stream())
    flops = 2 * B * (hb // 2) ** 2 * cs * 16 * cb
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"B={B} {which}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s")
