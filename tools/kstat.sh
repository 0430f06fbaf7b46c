#!/bin/bash
# quick per-kernel time table of the default bench under rocprofv3 (development aid): tools/kstat.sh <tag> [ENV=val ...]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -o b -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/ks_$tag/stdout.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/ks_$tag/**/b_kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
n=25
tot=0
for r in rows[:28]:
    t=float(r["TotalDurationNs"])/n/1000; tot+=t
    print(f'{t:8.1f} us/step  {int(r["Calls"])/n:5.1f}/step  avg {float(r["AverageNs"])/1000:7.1f} us  {r["Name"][:100]}')
print("sum of listed", round(tot,1))
PY
tail -1 gpurun_out/ks_$tag/stdout.log | cut -c1-200
