#!/bin/bash
# usage (GPU box, via gpurun): bash tools/gpu_round.sh <tag> [pytest -k expr]
# GPU parity suite, then a short bench run; logs under gpurun_out/
tag=$1; sel=$2
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ -n "$sel" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$sel" > $out/pytest_$tag.log 2>&1
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_$tag.log 2>&1
fi
rc=$?
tail -15 $out/pytest_$tag.log
echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --steps 400 --warmup 40 > $out/bench_$tag.json 2> $out/bench_$tag.err
rc=$?
echo "bench rc=$rc"
tail -3 $out/bench_$tag.err
python - "$out/bench_$tag.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "min", "max", "repeats", "update_only_per_sec", "actor_only_env_steps_per_sec")})
print("wall", d["wall_clock"])
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_us", "kernel")})
for k, v in d["kernels"].items():
    print(" ", k, round(v["avg_us"], 2), "us", round(v["achieved"], 2), v["unit"], round(v["frac"], 4), {x: round(v[x]) for x in ("updates_per_sec", "env_steps_per_sec") if x in v} or "")
print("bf16", d.get("bf16", {}).get("value"), "obs_wrapper_d9", d.get("obs_wrapper_d9", {}).get("value"))
cb = d.get("cpu_baseline", {})
print("cpu", cb.get("value"), cb.get("cores"), cb.get("one_thread", {}).get("value"), cb.get("torch_cpu", {}).get("value"))
PY
