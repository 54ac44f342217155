#!/bin/bash
# usage (GPU box): bash tools/prof_pmc.sh <tag>     -- two separate PMC passes (FETCH_SIZE, WRITE_SIZE) as
# MI355X_MICROARCH.md prescribes (TCC has 4 slots: the two do not fit one pass), kernel-trace only otherwise.
tag=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --profile-steps 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$c.err
  echo "$c rc=$?"
done
python3 - "$GRAFT_REPO_ROOT/gpurun_out" "$tag" <<'PY'
import csv, glob, json, sys, collections
root, tag = sys.argv[1], sys.argv[2]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{root}/pmc_{tag}_{c}/*/*counter_collection.csv")
    if not files:
        print("no counter file for", c); continue
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if row.get("Counter_Name") != c: continue
        n = row["Kernel_Name"]
        if not (n.startswith("k_") or n.startswith("void k_")): continue
        key = n.split("(")[0].replace("void ", "")
        if "qnet_fwd" in key: key += "/grid" + row.get("Grid_Size", row.get("Grid_Size_X", "?"))
        acc[key].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v)//4:]                        # skip setup / prefill launches
        out.setdefault(k, {})[c + "_KB_per_launch_median"] = sorted(v)[len(v)//2]
        out[k]["launches"] = len(v)
json.dump(out, open(f"{root}/pmc_{tag}.json", "w"), indent=1)
for k, v in sorted(out.items()): print(k, v)
PY
