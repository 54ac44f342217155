#!/bin/bash
# kernels of ONE iteration of the configs[4] loop (4 vector env steps + 1 update), in launch order:  bash tools/kt_loop.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ktl_$tag -- python3 $GRAFT_REPO_ROOT/tools/diag/cnn_loop_trace.py > $GRAFT_REPO_ROOT/gpurun_out/ktl_$tag.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/ktl_$tag.log
python3 - <<PY
import csv, glob, re, subprocess
f = sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/ktl_$tag/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def dem(n):
    if n.startswith("_Z"): n = subprocess.run(["c++filt", n.replace("DF16b", "Dh")], capture_output=True, text=True).stdout.strip().replace("half", "bf16")
    return re.sub(r"\(.*", "", n).replace("void ", "")[:56]
upd = [i for i, r in enumerate(rows) if "k_cnn_adam" in r["Kernel_Name"]]
a, b = upd[-3] + 1, upd[-2] + 1                      # one iteration: after an update's last kernel up to and including the next one's
tot = 0.0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    print("$tag", f"{d:7.1f}", dem(r["Kernel_Name"]))
print("$tag", "launches", b - a, "sum of kernel times us", round(tot, 1))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ktl_$tag
