#!/usr/bin/env python3
"""Copy the judged summaries of a gpurun profiling session into profiles/ (tracked):
    python tools/collect_profiles.py <tag> <round>      e.g.  r01final r01
Inputs under gpurun_out/: prof_<tag>/ (rocprofv3 --kernel-trace --stats of bench.py), pmc_<tag>.json
(tools/prof_pmc.sh), bench_<tag>.json, sweep_*.json."""
import csv
import glob
import os
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)
f = glob.glob(f"gpurun_out/prof_{tag}/*/*kernel_stats.csv")[0]
with open(f"profiles/{rnd}_kernel_stats.csv", "w", newline="") as o:
    w = csv.writer(o)
    cols = ["Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
    w.writerow(["Name"] + cols)
    for r in csv.DictReader(open(f)):
        w.writerow([r["Name"].split("(")[0][:100]] + [r[k] for k in cols])
f = glob.glob(f"gpurun_out/prof_{tag}/*/*kernel_trace.csv")[0]
tr = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_", "void k_"))]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
# two inner-loop iterations of the f32 loop inside a graph: k_actor (4 vector env steps + leaves + next batch's PER draw),
# k_qnet_fwd (gather + 3 passes), k_bwd_rows, k_dw (+ Adam + priority write-back)
i0 = len(tr) // 3
while "k_actor" not in tr[i0]["Kernel_Name"]:
    i0 += 1
with open(f"profiles/{rnd}_kernel_trace_one_step.csv", "w", newline="") as o:
    w = csv.writer(o)
    w.writerow(["kernel", "start_ns_rel", "duration_ns", "grid", "workgroup", "lds_bytes"])
    t0 = int(tr[i0]["Start_Timestamp"])
    for r in tr[i0:i0 + 8]:
        w.writerow([r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]) - t0,
                    int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"]])
import json
pmc = json.load(open(f"gpurun_out/pmc_{tag}.json")) if os.path.exists(f"gpurun_out/pmc_{tag}.json") else {}
if os.path.exists(f"gpurun_out/pmcs_{tag}.json"):          # the stand-alone sampler, one record per batch size
    pmc.update(json.load(open(f"gpurun_out/pmcs_{tag}.json")))
json.dump(pmc, open(f"profiles/{rnd}_pmc.json", "w"), indent=1)
mf = {}
for part in ("bench", "big", "big16", "big16f", "bench16", "cnn", "cnnf32"):                               # MFMA / wave-state counter passes
    f2 = f"gpurun_out/pmcm_{tag}_{part}.json"
    if os.path.exists(f2):
        mf.update({k: v for k, v in json.load(open(f2)).items() if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0})
if mf:
    json.dump(mf, open(f"profiles/{rnd}_pmc_mfma.json", "w"), indent=1)
tk = {}
for f2 in (f"gpurun_out/pmcm_{tag}_trunk.json", f"gpurun_out/pmcl_{tag}_trunk.json"):          # the CNN trunk kernel: wave-state + LDS counters, B = 2048
    if os.path.exists(f2):
        for k, v in json.load(open(f2)).items():
            if k.startswith("k_cnn_trunk"): tk.setdefault(k, {}).update(v)
if tk:
    json.dump(tk, open(f"profiles/{rnd}_pmc_trunk.json", "w"), indent=1)
for src, dst in ((f"gpurun_out/bench_{tag}.json", f"profiles/{rnd}_bench_under_rocprof.json"), (f"gpurun_out/cnn_{tag}.txt", f"profiles/{rnd}_cnn_kernels.txt"), (f"gpurun_out/cnn_sweep_{tag}.json", f"profiles/{rnd}_cnn_sweep.json"), (f"gpurun_out/cnn_sweep_{tag}_layerwise.json", f"profiles/{rnd}_cnn_sweep_layerwise_conv.json"),
                 (f"gpurun_out/probe_{tag}.json", f"profiles/{rnd}_per_sample_probe.json"), (f"gpurun_out/wprobe_{tag}.json", f"profiles/{rnd}_per_write_probe.json"),
                 (f"gpurun_out/pmcw_{tag}.json", f"profiles/{rnd}_pmc_per_write.json")):
    if os.path.exists(src):
        shutil.copy(src, dst)
for s in glob.glob(f"gpurun_out/sweep_{tag}*.json"):
    shutil.copy(s, f"profiles/{rnd}_{os.path.basename(s)}")
print(open(f"profiles/{rnd}_kernel_trace_one_step.csv").read())
