#!/bin/bash
# usage (GPU box): bash tools/prof_pmc_sample.sh <tag> [log2 capacity = 20] ["batch log2s"]  -- FETCH_SIZE and WRITE_SIZE of k_per_sample2 per batch size, in
# separate rocprofv3 --pmc passes (TCC has 4 slots: the two do not fit one pass), program directly after `--`.
tag=$1; log2n=${2:-20}; sizes=${3:-"10 16 18 20 22"}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmcs_${tag}_$c -- python3 $GRAFT_REPO_ROOT/tools/per_sample_probe.py --plain 6 --log2n $log2n --log2 $sizes > /dev/null 2> $out/pmcs_${tag}_$c.err
  echo "$c rc=$?"
done
python3 - "$out" "$tag" "$log2n" "$sizes" <<'PY'
import csv, glob, json, sys
root, tag = sys.argv[1], sys.argv[2]
sizes, per = [int(x) for x in sys.argv[4].split()], 6            # the probe's launch plan: `per` launches of each batch size, in this order
L = int(sys.argv[3])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{root}/pmcs_{tag}_{c}/*/*counter_collection.csv")
    if not files:
        print("no counter file for", c); continue
    rows = [r for r in csv.DictReader(open(files[0])) if r.get("Counter_Name") == c and "k_per_sample2" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    assert len(rows) == per * len(sizes), (len(rows), per * len(sizes))
    for i, lb in enumerate(sizes):
        v = sorted(float(r["Counter_Value"]) for r in rows[i * per + 1:(i + 1) * per])      # first launch of a size: cold
        rec = res.setdefault(f"k_per_sample2/B{1 << lb}", {"launches": len(v), "grid_threads": int(rows[i * per]["Grid_Size"])})
        rec[c + "_KB_per_launch_median"] = v[len(v) // 2]
for k, rec in res.items():
    B = int(k.split("/B")[1])
    rec["algorithmic_KB"] = (4 * L + 16 * 8 + 26) * B / 1024.0; rec["log2_capacity"] = L
    rec["fetch_correction"] = 1.0       # taken as reported: see DESIGN.md (gather of 32-byte rows, not a wide coalesced stream)
json.dump(res, open(f"{root}/pmcs_{tag}.json", "w"), indent=1)
for k, v in res.items(): print(k, v)
PY
