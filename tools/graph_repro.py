"""development aid: bisect a hipGraph replay crash (u8 input + tail engine + sampling)"""
import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jck-generation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from hipgan.engine import DcganEngine, DeviceBatch
from oracle.gan_oracle import build_params
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
torch.manual_seed(12345)
g, d = build_params("dcgan")
eng = DcganEngine(batch=16, prec="bf16")
eng.load_state(g, d)
tail = DcganEngine(batch=8, share=eng)
big = DcganEngine(batch=64, share=eng)
data = (torch.rand(40, 3, 32, 32) * 255).to(torch.uint8).cuda()
z = torch.randn(64, 100, 1, 1, device="cuda")
side = torch.cuda.Stream()
step = 0
for epoch in range(3):
    perm = torch.randperm(40).cuda()
    for i, (e, n) in enumerate(((eng, 16), (eng, 16), (tail, 8))):
        if mode == "notail" and e is tail:
            continue
        idx = perm[i * 16:i * 16 + n]
        real = DeviceBatch(data, idx) if mode != "f32" else DeviceBatch(data, idx).materialize()
        e.step_async(real, None, 2e-4)
        step += 1
        print("step", step, "B", n, "graphs", len(e._graph_cache), flush=True)
        if mode not in ("nosample",) and step in (1, 5):
            eng.join()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                img = big.sample(z)
            torch.cuda.current_stream().wait_stream(side)
            print("sampled", float(img.abs().mean()), flush=True)
torch.cuda.synchronize()
print("REPRO-OK", eng.scalars())
