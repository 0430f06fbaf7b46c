"""Development probe: per-role split of a wave-specialised weight-gradient k-step (JCK_WGRAD_STAMP=1, JCK_WGRAD_WS=1)."""
import ctypes, os, sys
os.environ["JCK_WGRAD_STAMP"] = "1"
sys.path.insert(0, '/root/repo/jck-generation_amd'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import torch
import gpu_util as G
for B, hb, cb, cs in ((512, 16, 128, 256), (512, 32, 64, 128), (512, 8, 256, 512), (256, 16, 128, 256)):
    big = torch.randn(B, hb, hb, cb, device="cuda").to(torch.bfloat16)
    small = torch.randn(B, hb // 2, hb // 2, cs, device="cuda").to(torch.bfloat16)
    nb = G.lib.jck_conv_wgrad_ws_bytes(B, hb, hb, cb, cs)
    ws = torch.empty(nb // 4, device="cuda")
    grad = torch.zeros(cs, cb, 4, 4, device="cuda")
    for _ in range(3):
        G.lib.jck_conv_wgrad(0, small, big, ws, nb, grad, 0, B, hb, hb, cb, cs, G.cur_stream())
    torch.cuda.synchronize()
    n = 256 * 8 * 4
    buf = (ctypes.c_ulonglong * n)()
    G.lib.jck_debug_wgrad_stamps(buf, n)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 4).astype(np.float64)
    print(f"B={B} hb={hb} cb={cb} cs={cs}: kernel ticks mean {a[..., 3].mean():.0f}")
    for role, sl in (("consumer", slice(0, 4)), ("loader", slice(4, 8))):
        r = a[:, sl, :]
        fr = r[..., :3].mean((0, 1)) / r[..., 3].mean()
        print(f"   {role:8s} wait+barrier {fr[0]:.3f}  issue {fr[1]:.3f}  lds+mfma {fr[2]:.3f}   (kernel {r[..., 3].mean():.0f} ticks)")
