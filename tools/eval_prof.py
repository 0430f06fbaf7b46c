"""Measurement of the evaluation branch (development aid; reference train/dcgan_trainer.py:198-212):
  * one pass of the metric network (InceptionV3Hip, seeded random weights - the fine-tuned ones exist nowhere offline) over the
    reference's 64 fixed-noise images: ms, images/sec, achieved TFLOP/s of the fp32 convolution chain;
  * the training step (DCGAN, batch 256, bf16) alone and while such a pass runs on a side stream.
Run under rocprofv3 for the per-kernel table:  rocprofv3 --kernel-trace --stats -d gpurun_out/eval -o ev -- python3 tools/eval_prof.py --only-eval
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jck-generation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def conv_flops(n_images):
    """2 * MACs of every convolution of the chain at 299x299 input (walks the same layer table the HIP chain walks)."""
    from inception import BLOCKS, STEM, _pair

    def out_hw(h, w, k, s, p):
        return (h + 2 * p[0] - k[0]) // s[0] + 1, (w + 2 * p[1] - k[1]) // s[1] + 1
    h = w = 299
    c, macs = 3, 0
    for op in STEM:
        if op[0] == "max":
            h, w = out_hw(h, w, _pair(op[1]), _pair(op[2]), _pair(op[3]))
            continue
        k, s, p = _pair(op[2]), _pair(op[3]), _pair(op[4])
        h, w = out_hw(h, w, k, s, p)
        macs += h * w * op[1] * c * k[0] * k[1]
        c = op[1]
    for _, branches in BLOCKS:
        cin, hin, win, total = c, h, w, 0
        for ops in branches.values():
            cb, hb, wb = cin, hin, win
            for op in ops:
                if op[0] in ("max", "avg"):
                    hb, wb = out_hw(hb, wb, _pair(op[1]), _pair(op[2]), _pair(op[3]))
                elif op[0] == "split":
                    for sub in op[1:]:
                        k = _pair(sub[2])
                        macs += hb * wb * sub[1] * cb * k[0] * k[1]
                    cb = sum(sub[1] for sub in op[1:])
                else:
                    k, s, p = _pair(op[2]), _pair(op[3]), _pair(op[4])
                    hb, wb = out_hw(hb, wb, k, s, p)
                    macs += hb * wb * op[1] * cb * k[0] * k[1]
                    cb = op[1]
            total += cb
            h, w = hb, wb
        c = total
    macs += c * 100
    return 2.0 * macs * n_images


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only-eval", action="store_true")
    ap.add_argument("--images", type=int, default=64)
    a = ap.parse_args()
    from inception import InceptionV3Hip
    from oracle.inception_oracle import random_state_dict
    dev = torch.device("cuda", 0)
    net = InceptionV3Hip(random_state_dict(0), device=dev)
    x = torch.rand(a.images, 3, 299, 299, device=dev)
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for _ in range(reps):
        net(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = conv_flops(a.images)
    out = {"eval": {"images": a.images, "ms_per_pass": round(ms, 3), "images_per_sec": round(a.images / ms * 1e3, 1),
                    "gflop_per_pass": round(fl / 1e9, 2), "achieved_tflops_fp32": round(fl / (ms * 1e-3) / 1e12, 2),
                    "fp32_mfma_peak_tflops": 157.3}}
    if not a.only_eval:
        from hipgan.engine import DcganEngine
        from model import DCGAN
        B = 256
        eng = DcganEngine(batch=B, prec="bf16", device=dev)
        torch.manual_seed(12345)
        g, d = DCGAN.Generator(), DCGAN.Discriminator()
        g.apply(DCGAN.weights_init)
        d.apply(DCGAN.weights_init)
        eng.load_state(g.state_dict(), d.state_dict())
        gen = torch.Generator(device=dev).manual_seed(1)
        batches = [torch.rand(B, 3, 64, 64, device=dev, generator=gen) * 2 - 1 for _ in range(4)]
        side = torch.cuda.Stream(device=dev)

        def steps(n, with_eval):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if with_eval:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(with_eval):
                        net(x)
            for i in range(n):
                eng.step_async(batches[i % 4], None, 2e-4)
            eng.join()
            torch.cuda.current_stream().synchronize()
            t_train = time.perf_counter() - t0
            torch.cuda.synchronize()
            return t_train / n * 1e3, (time.perf_counter() - t0) * 1e3
        steps(10, 0)
        alone, _ = steps(40, 0)
        n_eval = max(1, int(40 * alone / ms))            # as many metric passes as fit beside the 40 steps
        beside, total = steps(40, n_eval)
        out["training_step"] = {"alone_ms": round(alone, 4), "beside_metric_network_ms": round(beside, 4),
                                "slowdown": round(beside / alone, 3), "metric_passes_beside": n_eval,
                                "both_finished_ms": round(total, 2), "serial_would_be_ms": round(40 * alone + n_eval * ms, 2)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
