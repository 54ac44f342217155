#!/bin/bash
# usage (GPU box): bash tools/prof_pmc_write.sh <tag>  -- FETCH_SIZE and WRITE_SIZE of the sorted priority write-back kernels
# (k_per_write_seg, k_per_top_seg) per batch size, in separate rocprofv3 --pmc passes, program directly after `--`.
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmcw_${tag}_$c -- python3 $GRAFT_REPO_ROOT/tools/per_write_probe.py --plain 6 --log2 10 16 18 20 > /dev/null 2> $out/pmcw_${tag}_$c.err
  echo "$c rc=$?"
done
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys
root, tag = sys.argv[1], sys.argv[2]
sizes, per = [10, 16, 18, 20], 6            # the probe's launch plan: `per` calls of each batch size, in this order
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{root}/pmcw_{tag}_{c}/*/*counter_collection.csv")
    if not files:
        print("no counter file for", c); continue
    for kern in ("k_per_write_seg", "k_per_top_seg"):
        rows = [r for r in csv.DictReader(open(files[0])) if r.get("Counter_Name") == c and kern in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        if len(rows) != per * len(sizes):
            print(kern, c, "unexpected launch count", len(rows)); continue
        for i, lb in enumerate(sizes):
            v = sorted(float(r["Counter_Value"]) for r in rows[i * per + 1:(i + 1) * per])
            res.setdefault(f"{kern}/B{1 << lb}", {"launches": len(v)})[c + "_KB_per_launch_median"] = v[len(v) // 2]
for k, rec in res.items():
    B = int(k.split("/B")[1])
    rec["algorithmic_KB"] = (8 * 20 + 12) * B / 1024.0 if "write_seg" in k else None
json.dump(res, open(f"{root}/pmcw_{tag}.json", "w"), indent=1)
for k, v in sorted(res.items()): print(k, v)
PY
