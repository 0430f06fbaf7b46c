import ctypes, sys
sys.path.insert(0, '/root/repo/jck-generation_amd'); sys.path.insert(0, '/root/repo/tests')
import torch
import gpu_util as G
prec = 0
def run(B, hb, cb, cs, reps=20):
    x = torch.randn(B, hb, hb, cb, device="cuda").to(torch.bfloat16)
    w = G.pack_down(torch.randn(cs, cb, 4, 4) * 0.05, prec)
    out = torch.empty(B, hb // 2, hb // 2, cs, dtype=torch.bfloat16, device="cuda")
    stats, slots = G.stats_buf(B * (hb // 2) ** 2, cs)
    fn = lambda: G.lib.jck_conv_down(prec, x, w, out, stats, ctypes.byref(slots), B, hb, hb, cb, cs, G.cur_stream())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2 * B * (hb // 2) ** 2 * cs * 16 * cb
    print(f"B={B} hb={hb} cb={cb} cs={cs}: M={B*(hb//2)**2} K={16*cb} ksteps={16*cb//64}  {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TF", flush=True)
for cb in (64, 128, 256, 512):
    run(256, 16, cb, 256)
for B in (64, 128, 256, 512, 1024, 2048):
    run(B, 16, 128, 256)
for B in (256, 1024):
    run(B, 32, 64, 128)
