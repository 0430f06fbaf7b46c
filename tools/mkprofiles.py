"""Turns the rocprofv3 output of tools/prof.sh (gpurun_out/prof) into the committed round artifacts under profiles/:
rNN_bench_kernel_stats.csv, rNN_bench_domain_stats.csv, rNN_traffic.json (HBM bytes per launch from the separate
FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as the gfx950 guide prescribes, KB -> bytes) and rNN_summary.md.
Usage: python tools/mkprofiles.py r01 [bench.json ...]   (development aid; reads only gpurun_out/ and profiles/)"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def short(name):
    return name.split("(")[0].strip()


def counters(path):
    """{kernel: {counter: [values per dispatch]}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def cgan_summary(tag):
    """profiles/rNN_cgan_summary.md + rNN_cgan_kernel_stats.csv from tools/prof_cgan.sh's output (gpurun_out/prof_cgan), if present."""
    src = os.path.join(ROOT, "gpurun_out", "prof_cgan")
    stats = os.path.join(src, "bench_kernel_stats.csv")
    if not os.path.exists(stats):
        return
    shutil.copy(stats, os.path.join(DST, f"{tag}_cgan_kernel_stats.csv"))
    line = [l for l in open(os.path.join(src, "bench_stdout.log")).read().splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    n = b["steps"] + b["warmup"]
    rows = list(csv.DictReader(open(stats)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / n / 1e6
    calls = sum(int(r["Calls"]) for r in rows) / n

    def part(keys):
        return sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in keys)) / n / 1e6
    gemm = part(("igemm", "wgrad_dma", "wgrad_kernel", "img_down", "img_up"))
    bn = part(("bn_", "bn2_"))
    md = [f"# CGAN bench (eager launches, two HIP streams: the default), rocprofv3 --kernel-trace --stats, {n} traced steps ({tag})", "",
          f"Sum of kernel durations {tot:.3f} ms/step over {calls:.0f} launches/step: GEMM-shaped {gemm:.2f}, BatchNorm {bn:.2f}, "
          f"everything else {tot - gemm - bn:.2f} ms (the weight gradients overlap the main stream, so the sum exceeds the wall time); "
          f"the profiled run reported {b['ms_per_step']} ms/step = {b['value']} images/sec.", "",
          "| ms/step | launches/step | avg us | kernel |", "|---|---|---|---|"]
    for r in rows[:40]:
        c, t = int(r["Calls"]), float(r["TotalDurationNs"])
        md.append(f"| {t / n / 1e6:.4f} | {c / n:.1f} | {t / c / 1e3:.2f} | `{r['Name'][:100]}` |")
    open(os.path.join(DST, f"{tag}_cgan_summary.md"), "w").write("\n".join(md) + "\n")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    for f in ("bench_kernel_stats.csv", "bench_domain_stats.csv"):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, f"{tag}_{f}"))
    stdout = open(os.path.join(SRC, "bench_stdout.log")).read()
    line = [l for l in stdout.splitlines() if l.startswith("{")][-1]
    prof_bench = json.loads(line)
    nsteps = prof_bench["steps"] + prof_bench["warmup"]
    rows = list(csv.DictReader(open(os.path.join(SRC, "bench_kernel_stats.csv"))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    fetch = counters(os.path.join(SRC, "pmc_fetch_counter_collection.csv"))
    write = counters(os.path.join(SRC, "pmc_write_counter_collection.csv"))
    traffic = {}
    for k, d in fetch.items():
        if "FETCH_SIZE" not in d or k not in write:
            continue
        fs, ws = d["FETCH_SIZE"], write[k].get("WRITE_SIZE", [0.0])
        traffic[k] = {"launches": len(fs), "fetch_size_kb_avg": sum(fs) / len(fs), "write_size_kb_avg": sum(ws) / len(ws),
                      "read_bytes": sum(fs) / len(fs) * 2 * 1024, "write_bytes": sum(ws) / len(ws) * 1024}
    traffic = dict(sorted(traffic.items(), key=lambda kv: -(kv[1]["read_bytes"] + kv[1]["write_bytes"]) * kv[1]["launches"]))
    json.dump(traffic, open(os.path.join(DST, f"{tag}_traffic.json"), "w"), indent=1)
    sq = counters(os.path.join(SRC, "pmc_sq_counter_collection.csv"))
    mf = counters(os.path.join(SRC, "pmc_mfma_counter_collection.csv"))
    md = [f"# Round {tag} rocprofv3 summary (MI355X, `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps "
          f"{prof_bench['steps']} --warmup {prof_bench['warmup']} --no-cpu-baseline --no-roofline`)", "",
          f"{nsteps} traced steps. Sum of kernel durations {tot / nsteps / 1e6:.3f} ms/step (weight gradients run on a second "
          f"HIP stream beside the dgrad chain, so the sum can exceed the wall time); the profiled run itself reported "
          f"{prof_bench['ms_per_step']} ms/step, the un-profiled bench is in `profiles/{tag}_bench.json`.", "",
          "| ms/step | launches/step | avg us | kernel |", "|---|---|---|---|"]

    def part(keys):
        return sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in keys)) / nsteps / 1e6
    md.insert(3, f"By class (ms/step): gather-GEMMs {part(('igemm',)):.3f}, weight gradients {part(('wgrad',)):.3f} (second stream), "
                 f"BatchNorm backward {part(('bn_bwd',)):.3f} (resident form {part(('bn_bwd_res',)):.3f}), BatchNorm forward "
                 f"{part(('bn_act_fwd', 'bn_finalize', 'bn_fwd_fused')):.3f}, image-side convs {part(('img_down', 'img_up')):.3f}; "
                 f"{sum(int(r['Calls']) for r in rows) / nsteps:.0f} launches/step.")
    md.insert(4, "")
    for r in rows[:40]:
        n, t = int(r["Calls"]), float(r["TotalDurationNs"])
        md.append(f"| {t / nsteps / 1e6:.4f} | {n / nsteps:.1f} | {t / n / 1e3:.2f} | `{r['Name'][:100]}` |")
    md += ["", "## HBM traffic per launch from PMC (separate passes; FETCH_SIZE doubled per the gfx950 correction, KB -> bytes)", "",
           "| kernel | launches | FETCH_SIZE avg (KB) | read bytes (x2 x1024) | WRITE_SIZE avg (KB) | write bytes |", "|---|---|---|---|---|---|"]
    for k, v in list(traffic.items())[:24]:
        md.append(f"| `{k}` | {v['launches']} | {v['fetch_size_kb_avg']:.0f} | {v['read_bytes'] / 1e6:.2f} MB | "
                  f"{v['write_size_kb_avg']:.0f} | {v['write_bytes'] / 1e6:.2f} MB |")
    cols = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_MFMA"]
    cols2 = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"]
    md += ["", "## SQ counters, averages per launch", "",
           "| kernel | " + " | ".join(c.replace("SQ_", "") for c in cols + cols2) + " |", "|" + "---|" * (len(cols + cols2) + 1)]
    for k in sq:
        if not any(s in k for s in ("igemm", "wgrad", "img_", "bn_")):
            continue
        vals = [sq[k].get(c, []) for c in cols] + [mf.get(k, {}).get(c, []) for c in cols2]
        md.append(f"| `{k}` | " + " | ".join(f"{sum(v) / len(v):.3g}" if v else "-" for v in vals) + " |")
    open(os.path.join(DST, f"{tag}_summary.md"), "w").write("\n".join(md) + "\n")
    # one step as a timeline: start offset, duration, HIP stream, kernel (from the kernel trace of the stats run)
    import re
    tr = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], r["Kernel_Name"], r["Grid_Size_X"])
          for r in csv.DictReader(open(os.path.join(SRC, "bench_kernel_trace.csv")))]
    tr.sort()
    marks = [e[0] for e in tr if "step_tail" in e[3] or "pack_tail" in e[3]]
    if len(marks) >= 3:
        a, b = marks[-3], marks[-2]
        lines = [f"# one traced step of `bench.py` ({(b - a) / 1e3:.1f} us under rocprofv3): start us | duration us | HIP stream | kernel | grid"]
        for st, en, sid, name, grid in tr:
            if a < st <= b:
                nm = re.sub(r"\(.*", "", name.replace("void ", "")).replace("unsigned short", "bf16")[:60]
                lines.append(f"{(st - a) / 1e3:8.1f} {(en - st) / 1e3:7.1f} s{sid} {nm} g{grid}")
        open(os.path.join(DST, f"{tag}_timeline.txt"), "w").write("\n".join(lines) + "\n")
    for extra in sys.argv[2:]:
        base = os.path.basename(extra)
        shutil.copy(extra, os.path.join(DST, base if base.startswith(tag + "_") else f"{tag}_{base}"))
    cgan_summary(tag)
    print("wrote profiles/%s_*" % tag)


if __name__ == "__main__":
    main()
