#!/usr/bin/env python3
"""Headline benchmark: images/sec of the full DCGAN G+D training step (train/dcgan_trainer.py:155-189 of the
reference), 64x64x3 synthetic images, batch 256 per GPU, on the MI355X-native HIP path.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; data parallel = replicate, shard the batch, all-reduce D and G gradients with RCCL
(torch.distributed "nccl").  Rank 0 prints ONE JSON line.
A "step" = one pass of the hot path over one synthetic batch already resident in HBM, including the RNG draws
(instance noise x2, z, GP alpha), four D passes, one G pass, five backward chains, two Adam steps.  Every kernel is
launched from the host on two HIP streams (JCK_GRAPH=1 replays the step from captured hipGraphs instead: one host call per
step segment, but a captured step is one stream - `launch_mode` and `host_enqueue_ms_per_step` say which ran).
With one GPU the same line carries `secondary.cgan`: BASELINE.json configs[3] (CGAN, 10-class labels, batch 256, the
penalty back-propagated) measured the same way in the same process.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "jck-generation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0              # same guide: HBM3E peak 8.0 TB/s (6.29 TB/s measured for a float4 copy)
# algorithmic MACs per image: D forward, G forward and their pieces
D_FWD, D_CONV1, G_FWD, G_CONV1 = 103_817_216, 3_145_728, 104_628_224, 819_200


def step_flops_per_image_128():
    """The same step on the 128x128 plan (one more stride-2 stage, 1024 channels at the 4x4 end): D forward
    = conv1 3->64 @64x64 + 64->128 @32x32 + ... + 512->1024 @4x4 + the 16384-wide head; G the mirror image."""
    d = [(64 * 64, 64, 48), (32 * 32, 128, 1024), (16 * 16, 256, 2048), (8 * 8, 512, 4096), (4 * 4, 1024, 8192)]
    d_fwd = sum(px * co * k for px, co, k in d) + 16 * 1024
    d_c1 = d[0][0] * d[0][1] * d[0][2]
    g_c1 = 100 * 16 * 1024
    g_fwd = g_c1 + sum(px * co * k for px, co, k in d[1:]) + d_c1          # ConvT layers mirror D's convs
    mac = 4 * d_fwd + g_fwd + 2 * (d_fwd + (d_fwd - d_c1)) + d_fwd + d_fwd + g_fwd + (g_fwd - g_c1)
    return 2 * mac


def step_flops_per_image(dead_wgrad=False):
    """FLOPs of the work the engine performs per image and step (2 FLOP per MAC)."""
    mac = 4 * D_FWD + G_FWD                              # forward passes
    mac += 2 * (D_FWD + (D_FWD - D_CONV1))               # D backward for real and fake: wgrad + dgrad without conv1
    mac += D_FWD                                         # gradient-penalty pass: full dgrad chain
    mac += D_FWD                                         # G phase: dgrad through D to the image
    if dead_wgrad:
        mac += D_FWD                                     # the reference also forms D's (dead) weight gradients here
    mac += G_FWD + (G_FWD - G_CONV1)                     # G backward: wgrad + dgrad without conv1
    return 2 * mac


# profiler variant label (csrc/ops.hip PROF_NAMES) -> substring of the kernel symbol rocprofv3 reports
_SYMBOL = {"igemm<bf16,128,256>": "_kernel<128, 256,", "igemm<bf16,128,128>": "igemm_dma_kernel<128, 128,",
           "igemm<bf16,128,64>": "_kernel<128, 64,", "igemm<bf16,64,128>": "_kernel<64, 128,",      # igemm_dma_kernel / igemm_dma_persist_kernel
           "wgrad<bf16,128,128>": "wgrad_dma_kernel<3, 4, false, true, 1,",
           "wgrad<bf16,256,128>": "wgrad_dma_kernel<3, 4, false, true, 2,",
           "wgrad<bf16,64,64,img>": "wgrad_kernel<PrecBf16, 64, 64, 2>", "img_down<bf16>": "img_down_kernel", "img_up<bf16>": "img_up_kernel",
           "bn_act_fwd": "bn_act_fwd_kernel<", "bn_bwd_resident": "bn_bwd_res_kernel<"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/rNN_traffic.json: rocprofv3 --pmc FETCH_SIZE
    and --pmc WRITE_SIZE in separate runs of this command, FETCH_SIZE doubled as the gfx950 guide prescribes).  rocprofv3
    cannot run inside the timed process, so the number is read back from the newest committed profile; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    pat = _SYMBOL.get(kernel)
    if not files or not pat:
        return None
    with open(files[-1]) as f:
        tr = json.load(f)
    hits = [v for name, v in tr.items() if pat in name]           # a template kernel may appear as several instantiations
    n = sum(v["launches"] for v in hits)
    if not n:
        return None
    rd = sum(v["read_bytes"] * v["launches"] for v in hits) / n
    wr = sum(v["write_bytes"] * v["launches"] for v in hits) / n
    return {"bytes_per_launch": round(rd + wr), "read": round(rd), "write": round(wr), "source": os.path.basename(files[-1])}


def _host_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box hands a one-GPU job 16
    of its 128 hardware threads; timing the oracle on all 128 oversubscribes the share and reads 2-3x too slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return n, model


def cpu_baseline(batch, seconds_budget=30.0):
    """The CPU oracle (a restatement of the reference's step, pinned to it by tests/golden) timed on the host cores as
    BASELINE.md section 3 prescribes: a thread sweep (1, half, all of the CPUs this job may use) picks the thread count, then
    3 warm-up + up to 10 timed steps at the headline batch with it (bounded by `seconds_budget`)."""
    import torch
    from oracle.gan_oracle import GanOracle
    avail, model = _host_cpus()
    g = torch.Generator().manual_seed(2024)
    real = torch.rand(batch, 3, 64, 64, generator=g) * 2 - 1
    saved = torch.get_num_threads()
    sweep, small = {}, real[:64]
    for n in sorted({1, max(1, avail // 2), avail}):
        torch.set_num_threads(n)
        torch.manual_seed(0)
        orc = GanOracle("dcgan", lr=2e-4, seed=12345)
        orc.step(small)                                  # warm-up (allocator, thread pool)
        t0 = time.time()
        k = 0
        while k < 3 and (k < 1 or time.time() - t0 < 4.0):
            orc.step(small)
            k += 1
        sweep[n] = round(64 * k / (time.time() - t0), 1)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    torch.manual_seed(0)
    orc = GanOracle("dcgan", lr=2e-4, seed=12345)
    warm = 0
    t0 = time.time()
    while warm < 3 and (warm < 1 or time.time() - t0 < seconds_budget / 3):
        orc.step(real)
        warm += 1
    t0 = time.time()
    n = 0
    while n < 10 and (n < 2 or time.time() - t0 < seconds_budget):
        orc.step(real)
        n += 1
    dt = time.time() - t0
    torch.set_num_threads(saved)
    return {"value": round(batch * n / dt, 2), "unit": "images/sec", "cores": best, "kind": "port",
            "s_per_step": round(dt / n, 3), "cpu_model": model, "cpus_available": avail, "cpus_in_machine": os.cpu_count(),
            "thread_sweep_images_per_sec_batch64": {str(k): v for k, v in sweep.items()},
            "sample": f"{n} full G+D steps at batch {batch} after {warm} warm-up ({dt:.1f} s) with {best} threads (the fastest of "
                      f"the sweep at batch 64) on {model}, {avail} CPUs available to the job; torch fp32 CPU oracle "
                      f"(oracle/gan_oracle.py), anomaly detection off"}


def measure(a, model, world, rank, dev, dist):
    """Warm-up, the timed region, the per-launch profile of the MFMA kernels; -> dict (rank 0 fills the roofline)."""
    import torch
    from hipgan import lib
    from hipgan.dist import GradReducer
    from hipgan.engine import CganEngine, DcganEngine, DeviceBatch
    from model import CGAN, DCGAN
    B = a.batch
    cgan = model == "cgan"
    M = CGAN if cgan else DCGAN
    S = a.size if not cgan else 64
    eng = (CganEngine if cgan else DcganEngine)(batch=B, prec=a.prec, device=dev, **({"image_size": S} if S != 64 else {}))
    torch.manual_seed(12345)                            # change_randomseed.py:1 - identical weights on every rank
    net_g, net_d = (M.Generator(S), M.Discriminator(S)) if S != 64 else (M.Generator(), M.Discriminator())
    net_g.apply(M.weights_init)
    net_d.apply(M.weights_init)
    eng.load_state(net_g.state_dict(), net_d.state_dict())
    gen = torch.Generator(device=dev).manual_seed(2024 + rank)
    eng.set_noise_seed(2024 + rank)                      # in-kernel instance noise: its own stream per rank
    batches = [torch.rand(B, 3, S, S, device=dev, generator=gen) * 2 - 1 for _ in range(4)]
    if a.input == "u8":
        data = (torch.rand(50000, 3, 32, 32, device=dev, generator=gen) * 255).to(torch.uint8)        # CIFAR-sized
        batches = [DeviceBatch(data, torch.randint(0, 50000, (B,), device=dev, generator=gen)) for _ in range(4)]
    # CGAN: 10-class synthetic labels as one-hot int64 [B,100] (classes 0-9 of the reference's 100-wide encoding)
    labels = [torch.nn.functional.one_hot(torch.randint(0, 10, (B,), device=dev, generator=gen), 100).to(torch.int64)
              for _ in range(4)] if cgan else None
    # JCK_BENCH_FORCE_DDP=1 (test hook, tests/test_bench_multirank_gpu.py): the data-parallel control flow with ONE rank over a
    # real RCCL communicator - everything of the N > 1 path except a second device
    force_ddp = world == 1 and os.environ.get("JCK_BENCH_FORCE_DDP") == "1" and dist is not None
    red = GradReducer(world, force=force_ddp) if (world > 1 or force_ddp) else None
    overlap = {"on": os.environ.get("JCK_BENCH_DDP_OVERLAP", "1") != "0"}      # D split + D(real) prefetch
    exposed = {"d": [], "g": [], "on": False}

    def reducer(tag):
        """red.start whose wait is bracketed by HIP events while `exposed` is on: the time the compute stream is blocked before
        Adam(D) / Adam(G) = the part of the all-reduce that no compute hid."""
        def start(flat, **kw_):
            w = red.start(flat, **kw_)
            if w is None or not exposed["on"]:
                return w

            def wait():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                w()
                e1.record()
                exposed[tag].append((e0, e1))
            return wait
        return start

    count = [0]
    use_red = {"on": True}

    def one_step(_i):
        # device-side RNG inside the step like the reference: noise=None draws n1 | z | n2, alpha (and the dropout masks)
        i = count[0]
        count[0] += 1
        kw = {}
        if red and use_red["on"]:
            # data parallel: SUM all-reduce of both gradient arenas; the next batch is announced so that the forward half
            # of its D(real) pass runs under G's all-reduce
            kw = dict(reduce_d=reducer("d"), reduce_g=reducer("g"), grad_scale=1.0 / world,
                      next_real=None if (cgan or not overlap["on"]) else batches[(i + 1) % 4])
        if not kw and not cgan and overlap["on"]:
            # one GPU: the next batch is announced all the same (as the trainer does): the forward half of its D(real) pass runs beside
            # this step's Adam(G) + repack, where the second stream is idle - every step still does one D(real) forward
            kw = dict(next_real=batches[(i + 1) % 4])
        eng.step_async(batches[i % 4], None, 2e-4, generator=gen, labels=labels[i % 4] if cgan else None, **kw)

    # every rank must hold the same parameters (the all-reduced gradients went through the same Adam); on a mismatch the guard
    # re-broadcasts rank 0's state and switches the engine to the plain schedule (hipgan/dist.py - the trainers run the same guard)
    from hipgan.dist import ReplicaGuard
    guard = ReplicaGuard(eng, world if dist else 1)

    ddp = {}
    if not overlap["on"]:
        eng.ddp_overlap = False
    for i in range(a.warmup):
        one_step(i)
    torch.cuda.synchronize()
    if red:
        ddp["mode"] = ("D all-reduce in two pieces under D's own backward + G all-reduce under the next batch's D(real) forward"
                       if overlap["on"] and not cgan else "one all-reduce per network, waited for before its Adam")
        # phases issued while an all-reduce is in flight never launch a grid-barrier kernel when the collective has peers to wait
        # for: RCCL's kernel holds CUs (include/jckgan.h JCK_PHASE_NO_RESIDENT)
        ddp["bn_backward_under_collectives"] = ("three-launch form (no grid barrier while RCCL holds CUs)"
                                                if eng.collective_world() > 1 else "resident form (one rank: the collective waits for nobody)")
        torch.cuda.synchronize()
        ddp["replicas_in_sync_after_warmup"] = guard.check()
        if not ddp["replicas_in_sync_after_warmup"] and overlap["on"]:
            # the overlapped schedule has only ever run against one device (tests): if the first multi-device run disagrees,
            # measure the plain schedule instead of a wrong one - and say so
            overlap["on"] = False
            for i in range(max(2, a.warmup)):
                one_step(i)
            ddp["mode"] = "one all-reduce per network, waited for before its Adam (overlapped schedule failed the replica check)"
            torch.cuda.synchronize()
            ddp["replicas_in_sync_after_fallback"] = guard.in_sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        one_step(i)
    t_host = time.perf_counter() - t0                   # host time to enqueue the timed steps (diagnostic)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    scal = eng.scalars()
    ms = dt / a.steps * 1e3
    fl = step_flops_per_image() * B
    if S == 128:
        fl = step_flops_per_image_128() * B
    if cgan:   # + v-chain (D forward-like + direct wgrads) and the reverse sweep (wgrad + dgrad) of the penalty, + linear1
        fl += 2 * B * (D_FWD + D_FWD + D_FWD + (D_FWD - D_CONV1))
    res = {"value": round(world * B * a.steps / dt, 1), "ms_per_step": round(ms, 4),
           "host_enqueue_ms_per_step": round(t_host / a.steps * 1e3, 4),
           "launch_mode": "hipGraph replay" if eng.graphs and eng._graph_cache else "eager launches",
           "losses_last_step": {k: round(v, 5) for k, v in scal.items() if k in ("loss_d", "loss_g", "gp")},
           "step_mfma": {"flops_per_step": fl, "achieved_tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                         "frac_of_peak": round(fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}}
    if red:
        # what the collectives cost (every rank runs these legs; after the timed region, which stays exactly K steps):
        #   comm_exposed_ms   HIP events around the waits in front of Adam(D) / Adam(G), mean over 5 steps and max over ranks
        #   n1_equivalent_ms  the same step with the reducers off (what one GPU of this node does alone); DESIGN.md section 6's
        #                     efficiency estimate is n1_equivalent_ms / ms_per_step
        exposed["on"] = True
        for i in range(5):
            one_step(i)
        torch.cuda.synchronize()
        exposed["on"] = False
        ce = torch.tensor([sum(e0.elapsed_time(e1) for e0, e1 in exposed[t]) / 5.0 for t in ("d", "g")], device=dev, dtype=torch.float64)
        if dist and world > 1:
            dist.all_reduce(ce, op=dist.ReduceOp.MAX)
        ddp["comm_exposed_ms"] = {"d": round(float(ce[0]), 4), "g": round(float(ce[1]), 4)}
        torch.cuda.synchronize()
        ddp["replicas_in_sync_at_end"] = guard.in_sync()
        use_red["on"] = False
        for i in range(2):
            one_step(i)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(10):
            one_step(i)
        torch.cuda.synchronize()
        tn1 = torch.tensor([(time.perf_counter() - t1) / 10 * 1e3], device=dev, dtype=torch.float64)
        if dist and world > 1:
            dist.all_reduce(tn1, op=dist.ReduceOp.MAX)
        use_red["on"] = True
        ddp["n1_equivalent_ms"] = round(float(tn1[0]), 4)
        ddp["rccl_world"] = dist.get_world_size() if dist else 1
        ddp["backend"] = dist.get_backend() if dist else None
        # who runs the all-reduce: torch.distributed's ProcessGroupNCCL (default) or the library's own RCCL communicator behind the
        # C ABI (JCK_RCCL_NATIVE=1: include/jckgan.h jck_comm_*)
        ddp["collective"] = "jck_comm_* (RCCL behind the C ABI)" if getattr(red, "_comm", None) is not None else "torch.distributed.all_reduce"
        ddp["message_bytes"] = {"d": int(eng.arenas["d_grads"].numel()) * 4, "g": int(eng.arenas["g_grads"].numel()) * 4}
        res["ddp"] = ddp
    if not a.no_roofline:
        # every rank runs the three extra steps (they contain the gradient all-reduce); rank 0 times its launches with HIP
        # events on the launch streams - which needs every launch to come from the host, so these steps are not replayed
        eng.graphs = False
        torch.cuda.synchronize()
        if rank == 0:
            lib.jck_prof_enable(1)
        for i in range(3):
            one_step(i)
        torch.cuda.synchronize()
    if rank == 0 and not a.no_roofline:
        import ctypes as C
        lib.jck_prof_enable(0)
        cap = 64
        names, cnt, msv, flv = (C.c_char_p * cap)(), (C.c_int * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
        byv, stv = (C.c_double * cap)(), (C.c_void_p * cap)()
        n = lib.jck_prof_collect(cap, names, cnt, msv, flv, byv, stv)
        main_stream = torch.cuda.current_stream().cuda_stream
        # one record per (kernel, HIP stream the launches were issued on) - from the launch records themselves: the
        # engine's second stream carries the weight-gradient products beside the dgrad / BatchNorm chain and, since round 4, D(real)'s
        # forward beside G's
        def make_row(name, count, ms, fl, by, stream):
            r = {"kernel": name, "launches_per_step": count / 3, "avg_ms": ms / count, "ms_per_step": ms / 3, "stream": stream}
            if by > 0:
                r["bound"], r["gbs"] = "hbm", by / (ms * 1e-3) / 1e9
            else:
                r["bound"], r["tflops"] = "mfma", fl / (ms * 1e-3) / 1e12
            return r
        per_stream, total = [], {}
        for i in range(n):
            on_main = (stv[i] or 0) == main_stream
            per_stream.append(make_row(names[i].decode(), cnt[i], msv[i], flv[i], byv[i], "main" if on_main else "side (beside the main stream)"))
            t = total.setdefault(names[i].decode(), [0, 0.0, 0.0, 0.0, set()])
            t[0] += cnt[i]; t[1] += msv[i]; t[2] += flv[i]; t[3] += byv[i]; t[4].add(on_main)
        rows = [make_row(k, t[0], t[1], t[2], t[3], "main" if t[4] == {True} else ("side (beside the main stream)" if t[4] == {False} else "main + side"))
                for k, t in total.items()]
        rows.sort(key=lambda r: -r["ms_per_step"])
        per_stream.sort(key=lambda r: -r["ms_per_step"])
        method = "HIP events around every launch on the stream it is launched on, 3 extra (eagerly launched) steps after the timed region"

        def mfma_obj(d, selection):
            tr = pmc_traffic(d["kernel"])
            return {"bound": "mfma", "kernel": d["kernel"], "achieved": round(d["tflops"], 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(d["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4),
                    "traffic": (tr or {}).get("bytes_per_launch"), "traffic_detail": tr,      # HBM bytes per launch (PMC)
                    "avg_launch_ms": round(d["avg_ms"], 5), "launches_per_step": d["launches_per_step"], "ms_per_step": round(d["ms_per_step"], 5),
                    "stream": d["stream"], "selection": selection, "method": method}
        mf = [r for r in rows if r["bound"] == "mfma"]
        hb = [r for r in rows if r["bound"] == "hbm"]
        if mf:
            # (1) the kernel with the largest total time in the step, whatever stream it runs on; (2) the largest one on the
            # step's critical (main) stream - the weight gradients overlap that chain on the second stream, so their launch
            # durations are those of a kernel sharing the chip; (3) the largest HBM-bound kernel (the BatchNorm side of the step)
            res["roofline"] = mfma_obj(mf[0], "largest total time per step among the MFMA kernels, any stream")
            crit = [r for r in per_stream if r["bound"] == "mfma" and r["stream"] == "main"] or mf
            res["roofline_critical"] = mfma_obj(crit[0], "largest total time per step among the MFMA kernels' launches on the step's main stream "
                                                         "(launches of the same kernel on the second stream are not counted)")
        if hb:
            d = hb[0]
            tr = pmc_traffic(d["kernel"])
            res["roofline_hbm"] = {"bound": "hbm", "kernel": d["kernel"], "achieved": round(d["gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(d["gbs"] / HBM_PEAK_GBS, 4), "traffic": (tr or {}).get("bytes_per_launch"), "traffic_detail": tr,
                                   "algorithmic_bytes_per_launch": round(d["gbs"] * 1e9 * d["avg_ms"] * 1e-3),
                                   "avg_launch_ms": round(d["avg_ms"], 5), "launches_per_step": d["launches_per_step"],
                                   "ms_per_step": round(d["ms_per_step"], 5), "stream": d["stream"],
                                   "selection": "largest total time per step among the streaming BatchNorm kernels; algorithmic bytes = "
                                                "forward: read y + write a; backward: read g_a and y + write g_y",
                                   "method": method}
        if rows:
            res["kernels"] = [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]
            res["mfma_kernels_ms_per_step"] = round(sum(r["ms_per_step"] for r in mf), 4)
            res["bn_kernels_ms_per_step"] = round(sum(r["ms_per_step"] for r in hb), 4)
    del eng
    torch.cuda.empty_cache()
    return res


def self_launch(n):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks as child processes of this one
    (same arguments; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would), pass rank 0's JSON line
    through, return the worst exit status.  The parent makes no HIP call (nothing here imports torch): a process that has
    initialised the GPU must not be the one that starts or replaces others on this pool."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:                       # a free port on the loopback interface
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if out:
        sys.stdout.write(out)
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, status): {bad}", file=sys.stderr)
        return max(abs(c) for _, c in bad) or 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--prec", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--size", type=int, default=64, choices=[64, 128],
                    help="64 = the reference's nets (headline); 128 = BASELINE.json configs[4]'s topology (use --batch 128)")
    ap.add_argument("--model", default="dcgan", choices=["dcgan", "cgan"],
                    help="dcgan = the headline config (BASELINE.json configs[1]); cgan = configs[3] (label-concat path, 10 classes)")
    ap.add_argument("--input", default="tensor", choices=["tensor", "u8"],
                    help="tensor: fp32 NCHW batches resident in HBM (default); u8: index batches into a uint8 32x32 dataset "
                         "resident in HBM, Resize/ToTensor/Normalize done inside the step (the training pipeline's form)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the CGAN (configs[3]) measurement of the one-GPU run")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process starts the N ranks itself and never touches the GPU
        sys.exit(self_launch(a.gpus))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world} (launch one rank per GPU: torch.distributed.run "
                 f"--nproc-per-node {a.gpus}, or run `python bench.py --gpus {a.gpus}` without WORLD_SIZE set)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback on the product path)"
    # test hooks (tests/test_bench_multirank_gpu.py rehearses the N > 1 control flow on a one-GPU box): every rank on
    # cuda:0 and gloo instead of RCCL (RCCL needs one device per rank).  Never set by the driver.
    if os.environ.get("JCK_BENCH_ONE_GPU") == "1":
        local = 0
        os.environ.setdefault("JCK_BN_RES", "0")      # ranks share one GPU: the resident (grid-barrier) launches need a chip each
    backend = os.environ.get("JCK_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("JCK_BENCH_FORCE_DDP") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))     # the one-rank rehearsal hook only
        else:
            os.environ.setdefault("MASTER_PORT", "29500")      # every rank must agree: a fixed default (torchrun sets it anyway)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    B = a.batch
    cgan = a.model == "cgan"
    res = measure(a, a.model, world, rank, dev, dist)
    out = {"metric": "images/sec (G+D step) DCGAN 64x64 bs256" if not cgan else "images/sec (G+D step) CGAN 64x64 bs256",
           "value": res.pop("value"), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "host_enqueue_ms_per_step": res.pop("host_enqueue_ms_per_step"), "ms_per_step": res.pop("ms_per_step"),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.prec,
           "data": "synthetic" if a.input == "tensor" else "synthetic uint8 dataset in HBM, transformed in the step",
           "config": {"workload": (f"DCGAN 64x64x3 synthetic, batch {B} per GPU, full G+D step incl. GP pass, RNG and Adam "
                                   f"(BASELINE.json configs[1]{'/[2]' if world > 1 else ''})") if not cgan else
                                  (f"CGAN 64x64x3, 10-class synthetic one-hot labels, batch {B} per GPU, full step incl. the "
                                   f"back-propagated gradient penalty (BASELINE.json configs[3])"),
                      "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                      "dead_D_wgrad_in_G_phase": "skipped (never observable: zeroed at train/dcgan_trainer.py:155)"}}
    out.update(res)
    if a.size == 128:
        out["metric"] = "images/sec (G+D step) DCGAN 128x128"
        out["config"]["workload"] = (f"DCGAN 128x128x3 synthetic (one more stride-2 stage, 1024 channels at 4x4), batch {B} per GPU, "
                                     f"full G+D step (BASELINE.json configs[4] without the Inception/FID evaluation)")
    if world == 1 and not cgan and not a.no_secondary and a.size == 64:
        # BASELINE.json configs[3] in the same process and the same way, so that its throughput is a driver-run number too
        sec = measure(a, "cgan", world, rank, dev, dist)
        sec = {"metric": "images/sec (G+D step) CGAN 64x64 bs256", "unit": "images/sec", "dtype": a.prec,
               "config": {"workload": f"CGAN 64x64x3, 10-class synthetic one-hot labels, batch {B}, full step incl. the "
                                      f"back-propagated gradient penalty (BASELINE.json configs[3])"}, **sec}
        sec.pop("kernels", None)
        out["secondary"] = {"cgan": sec}
        if a.prec == "bf16":
            # the headline config on the PARITY path (exact-fp32 MFMA, fp32 storage): the precision that meets the north star's
            # 1e-3 per-step criterion (tests/test_step_gpu.py), driver-run every round beside the bf16 number
            import copy
            a32 = copy.copy(a)
            a32.prec, a32.steps, a32.warmup, a32.no_roofline = "f32", min(a.steps, 10), min(a.warmup, 3), True
            s32 = measure(a32, "dcgan", world, rank, dev, dist)
            s32.pop("kernels", None)
            out["secondary"]["f32"] = {"metric": "images/sec (G+D step) DCGAN 64x64 bs256", "unit": "images/sec", "dtype": "f32",
                                       "steps": a32.steps, "warmup": a32.warmup,
                                       "config": {"workload": f"DCGAN 64x64x3 synthetic, batch {B}, the headline step on the parity path "
                                                              f"(fp32 storage, v_mfma_f32_16x16x4_f32: losses within 1e-3 of the CPU oracle per step)"},
                                       **s32}
            # BASELINE.json configs[4]'s per-GPU workload (128x128 topology, batch 128; without the Inception / FID evaluation, which
            # profiles/r03_eval.json measures): a driver-run number for the one config that is otherwise exercised by tests only
            a128 = copy.copy(a)
            a128.size, a128.batch, a128.steps, a128.warmup, a128.no_roofline = 128, 128, min(a.steps, 20), min(a.warmup, 5), True
            s128 = measure(a128, "dcgan", world, rank, dev, dist)
            s128.pop("kernels", None)
            out["secondary"]["size128"] = {"metric": "images/sec (G+D step) DCGAN 128x128 bs128", "unit": "images/sec", "dtype": "bf16",
                                           "steps": a128.steps, "warmup": a128.warmup,
                                           "config": {"workload": "DCGAN 128x128x3 synthetic (one more stride-2 stage, 1024 channels at 4x4), batch 128, "
                                                                  "full G+D step (BASELINE.json configs[4] per GPU, without the Inception/FID evaluation)"},
                                           **s128}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
