#!/bin/bash
# usage (GPU box): bash tools/prof_pmc_mfma.sh <tag> <program args...>
# one rocprofv3 --pmc pass with the LDS counters (conflicts, unaligned replays, active cycles) + MFMA busy (8 SQ slots + GRBM), program directly after `--`;
# prints and stores per-kernel sums: MFMA busy cycles, wave cycles, waits, LDS conflicts.
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmcl_$tag -- python3 "$@" > $out/pmcl_$tag.log 2>&1
echo "rc=$?"; tail -2 $out/pmcl_$tag.log
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys, collections
import subprocess
_dem = {}
def dem(n):      # rocprofv3 leaves names with the bf16 type (DF16b) mangled: demangle them as `half`, then rename
    if not n.startswith("_Z"): return n
    if n not in _dem:
        try: _dem[n] = subprocess.run(["c++filt", n.replace("DF16b", "Dh")], capture_output=True, text=True).stdout.strip().replace("half", "bf16") or n
        except Exception: _dem[n] = n
    return _dem[n]
root, tag = sys.argv[1], sys.argv[2]
files = glob.glob(f"{root}/pmcl_{tag}/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(files[0])):
    n = dem(row["Kernel_Name"]).split("(")[0].replace("void ", "")
    if not n.startswith("k_"): continue
    acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, d in acc.items():
    r = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    r["launches"] = len(next(iter(d.values())))
    g = r.get("GRBM_GUI_ACTIVE", 0)
    if g:   # GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles; 1024 SIMDs
        r["mfma_busy_frac"] = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g / 8 * 1024)
    res[k] = r
json.dump(res, open(f"{root}/pmcl_{tag}.json", "w"), indent=1)
for k, r in sorted(res.items()):
    print(k[:44], {c: (round(v, 4) if isinstance(v, float) and v < 10 else int(v)) for c, v in r.items()})
PY
