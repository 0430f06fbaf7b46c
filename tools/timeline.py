"""One traced step of a rocprofv3 kernel trace as a timeline (development aid): python tools/timeline.py <kernel_trace.csv> [out.txt]
start offset us | duration us | HIP stream | kernel | grid, between the last-but-two and last-but-one end-of-step (pack_tail) launches."""
import csv
import re
import sys


def main():
    tr = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], r["Kernel_Name"], r["Grid_Size_X"])
          for r in csv.DictReader(open(sys.argv[1]))]
    tr.sort()
    marks = [e[0] for e in tr if "step_tail" in e[3] or "pack_tail" in e[3]]
    a, b = marks[-3], marks[-2]
    lines = [f"# one traced step ({(b - a) / 1e3:.1f} us under rocprofv3): start us | duration us | HIP stream | kernel | grid"]
    for st, en, sid, name, grid in tr:
        if a < st <= b:
            nm = re.sub(r"\(.*", "", name.replace("void ", "")).replace("unsigned short", "bf16")[:60]
            lines.append(f"{(st - a) / 1e3:8.1f} {(en - st) / 1e3:7.1f} s{sid} {nm} g{grid}")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    else:
        sys.stdout.write(out)


if __name__ == "__main__":
    main()
