#!/bin/bash
# per-kernel time table of the bench with extra bench arguments under rocprofv3 (development aid): tools/kstat_args.sh <tag> <bench args...>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary "$@" > gpurun_out/ks_$tag/stdout.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/ks_$tag/**/b_kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
n=13
tot=0
for r in rows[:30]:
    t=float(r["TotalDurationNs"])/n/1000; tot+=t
    print(f'{t:8.1f} us/step  {int(r["Calls"])/n:5.1f}/step  avg {float(r["AverageNs"])/1000:7.1f} us  {r["Name"][:110]}')
print("sum of listed", round(tot,1))
PY
tail -1 gpurun_out/ks_$tag/stdout.log | cut -c1-200
