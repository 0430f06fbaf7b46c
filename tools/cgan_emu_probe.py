"""development aid: CGAN distances hip / emu / ref per group (limits of tests/test_bf16_envelope.py come from here)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "jck-generation_amd")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bf16_error as be
for fam in ("cgan",):
    for B, steps in ((8, 2), (64, 2), (256, 1)):
        t = time.time()
        rows = be.measure(fam, B, steps)
        out = {}
        for group in ("scalars", "d_grads", "g_grads"):
            for key in ("hip_vs_ref", "emu_vs_ref", "hip_vs_emu"):
                out[f"{group}:{key}"] = round(be.worst(rows, group, key), 4)
        print(fam, B, f"{time.time() - t:.1f}s", out, flush=True)
        worst_t = {}
        for group in ("d_grads", "g_grads"):
            for k in rows[0][group]:
                h = max(r[group][k]["hip_vs_ref"] for r in rows); e = max(r[group][k]["emu_vs_ref"] for r in rows); he = max(r[group][k]["hip_vs_emu"] for r in rows)
                worst_t[f"{group[0]}:{k}"] = (round(h, 3), round(e, 3), round(he, 3), round(h / max(e, 1e-9), 2))
        print("   per tensor (hip_vs_ref, emu_vs_ref, hip_vs_emu, ratio):", worst_t, flush=True)
