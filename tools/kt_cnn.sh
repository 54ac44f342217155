#!/bin/bash
# per-kernel medians of one CNN update / forward under rocprofv3:  bash tools/kt_cnn.sh <tag> [cnn_probe args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_$tag -- python3 $GRAFT_REPO_ROOT/tools/cnn_probe.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/kt_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections, re
import subprocess
_dem = {}
def dem(n):      # rocprofv3 leaves names with the bf16 type (DF16b) mangled: demangle them as `half`, then rename
    if not n.startswith("_Z"): return n
    if n not in _dem:
        try: _dem[n] = subprocess.run(["c++filt", n.replace("DF16b", "Dh")], capture_output=True, text=True).stdout.strip().replace("half", "bf16") or n
        except Exception: _dem[n] = n
    return _dem[n]
f = sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/kt_$tag/*/*kernel_trace.csv"))[-1]
acc = collections.defaultdict(list); order = []
for r in csv.DictReader(open(f)):
    n = dem(r["Kernel_Name"])
    if "k_cnn" in n or "k_td" in n or "k_conv" in n:
        n = re.sub(r"\(.*", "", n).replace("void ", "")[:60]
        if n not in acc: order.append(n)
        acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k in order:
    v = sorted(acc[k]); per = len(v) / 23.0; tot += v[len(v)//2] * per
    print("$tag", k, "x%.1f" % per, "median us", round(v[len(v)//2], 1))
print("$tag", "sum of kernel medians per step", round(tot, 1))
PY
