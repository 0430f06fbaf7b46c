"""Development probe: where a weight-gradient workgroup spends its k-steps (JCK_WGRAD_STAMP=1)."""
import ctypes, os, sys
os.environ["JCK_WGRAD_STAMP"] = "1"
sys.path.insert(0, '/root/repo/jck-generation_amd'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import torch
import gpu_util as G
B = int(os.environ.get("MB_B", "512"))
hb, cb, cs = 16, 128, 256
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
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 4)[:, :4, :].astype(np.float64)
nk = B * 64 // 256 // 64 * (256 // 256)  # informational only
print("per wave totals (s_memtime ticks), mean over 256 WGs x 4 waves:")
for i, nm in enumerate(("wait+barrier", "dma issue", "lds+mfma", "kernel")):
    print(f"  {nm:14s} mean {a[..., i].mean():10.0f}  min {a[..., i].min():10.0f}  max {a[..., i].max():10.0f}")
print("fractions of kernel:", (a[..., :3].mean((0, 1)) / a[..., 3].mean()).round(3))
