#!/usr/bin/env python3
"""Records tests/golden/bf16_error_table.json: the measured distance of the bf16 fast path (HIP, MI355X) from the fp32 CPU
oracle and from the bf16-storage emulation, per gradient tensor and per logged scalar, for DCGAN and CGAN at batch 8 / 64 /
256 (see tests/bf16_error.py).  Run on a GPU box from the repo root:

    python tests/golden/make_bf16_error_table.py [out.json]

`tests/test_bf16_envelope.py` holds the path to <= 2x these values.  Nothing of /root/reference is read."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "jck-generation_amd"), os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

CASES = [("dcgan", 8, 2), ("dcgan", 64, 2), ("dcgan", 256, 1), ("cgan", 8, 2), ("cgan", 64, 2), ("cgan", 256, 1)]


def main():
    import torch
    import bf16_error as be
    out = {"_about": "relative L2 (gradient tensors) / relative error (scalars) of one teacher-forced step; hip = JCK_PREC_BF16 engine, "
                     "ref = fp32 CPU oracle, emu = oracle with bf16 storage emulation (DCGAN only); max over the steps of a case",
           "device": torch.cuda.get_device_name(0), "cases": {}}
    for family, B, steps in CASES:
        rows = be.measure(family, B, steps)
        case = {"steps": steps, "scalars": {}, "d_grads": {}, "g_grads": {}}
        for group in ("scalars", "d_grads", "g_grads"):
            for k in rows[0][group]:
                case[group][k] = {m: max(r[group][k][m] for r in rows) for m in rows[0][group][k]}
        out["cases"][f"{family}_B{B}"] = case
        print(family, B, {m: f"{be.worst(rows, 'd_grads', m):.3e}" for m in rows[0]['d_grads']['conv1.weight']},
              {m: f"{be.worst(rows, 'g_grads', m):.3e}" for m in rows[0]['g_grads']['conv1.weight']},
              {k: f"{v['hip_vs_ref']:.2e}" for k, v in case["scalars"].items() if k in ("loss_d", "loss_g", "gp")}, flush=True)
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "bf16_error_table.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    # markdown rendering for profiles/
    md = [f"# bf16 fast path: measured error per tensor ({out['device']})", "",
          "One teacher-forced step from identical state; relative L2 for gradients, relative error for scalars; max over the "
          "steps of a case.  hip = engine (JCK_PREC_BF16), ref = fp32 CPU oracle, emu = oracle with bf16 storage emulation "
          "(oracle/bf16_emu.py).  Test limits = 2x `hip_vs_ref` / `hip_vs_emu` (tests/test_bf16_envelope.py).", ""]
    for name, case in out["cases"].items():
        md += [f"## {name} ({case['steps']} step(s))", "", "| tensor | hip_vs_ref | emu_vs_ref | hip_vs_emu |", "|---|---|---|---|"]
        for group in ("scalars", "d_grads", "g_grads"):
            for k, v in case[group].items():
                f3 = lambda m: f"{v[m]:.2e}" if m in v else "-"
                md.append(f"| {group}:{k} | {f3('hip_vs_ref')} | {f3('emu_vs_ref')} | {f3('hip_vs_emu')} |")
        md.append("")
    with open(os.path.splitext(dst)[0] + ".md", "w") as f:
        f.write("\n".join(md))


if __name__ == "__main__":
    main()
