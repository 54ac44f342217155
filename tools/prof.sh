#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools_prof.sh <tag> [bench args]
# kernel-trace profile of bench.py; prints the per-kernel summary of our kernels
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --profile-steps 5 "$@" > $GRAFT_REPO_ROOT/gpurun_out/bench_$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_$tag.err
echo "rc=$?"
python3 - "$out" "$GRAFT_REPO_ROOT/gpurun_out/bench_$tag.json" <<'PY'
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for row in csv.DictReader(open(f)):
    n = row['Name']
    if n.startswith('k_') or n.startswith('void k_'):
        print(f"{n.split('(')[0][:30]:30s} calls {row['Calls']:>6s} avg {float(row['AverageNs'])/1e3:8.2f}us min {float(row['MinNs'])/1e3:7.2f} max {float(row['MaxNs'])/1e3:8.2f} pct {float(row['Percentage']):.2f}")
try:
    d = json.load(open(sys.argv[2]))
    print('steps/s', round(d['value'], 1), 'update-only/s', round(d['update_only_per_sec'], 1), 'actor env-steps/s', round(d['actor_only_env_steps_per_sec']))
except Exception as e:
    print('bench json unreadable:', e)
PY
