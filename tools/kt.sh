#!/bin/bash
# usage (GPU box): bash tools/kt.sh <tag> <script + args>   -- rocprofv3 kernel trace, per-kernel average of our kernels
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$tag -- python3 "$@" > $out/kt_$tag.log 2>&1
tail -1 $out/kt_$tag.log
python3 - $out/kt_$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for row in csv.DictReader(open(f)):
    n = row["Name"]
    if n.startswith("k_") or n.startswith("void k_"):
        if int(row["Calls"]) >= 4: print("   ", n.split("(")[0][:48], row["Calls"], "avg us", round(float(row["AverageNs"]) / 1e3, 2), "min", round(float(row["MinNs"]) / 1e3, 2))
PY
