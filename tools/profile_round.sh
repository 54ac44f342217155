#!/bin/bash
# usage (GPU box): bash tools/profile_round.sh <tag>   -- every rocprofv3 pass whose summary is committed under profiles/:
#   kernel-trace + stats of bench.py; FETCH_SIZE / WRITE_SIZE of the bench kernels and of the stand-alone sampler (separate
#   passes); one MFMA / wave-state counter pass for the bench kernels, the large-batch kernels and the CNN update (bf16, f32);
#   per-kernel medians of the CNN forward / update (kernel trace); batch sweeps.
tag=$1
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out
bash $R/tools/prof.sh $tag --steps 400 --warmup 40 2>&1 | tail -12
bash $R/tools/prof_pmc.sh $tag 2>&1 | tail -8
bash $R/tools/prof_pmc_sample.sh $tag 2>&1 | tail -7
bash $R/tools/prof_pmc_mfma.sh ${tag}_bench $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --profile-steps 2 2>&1 | grep "^k_\|rc=" | cut -c1-330
bash $R/tools/prof_pmc_mfma.sh ${tag}_big $R/tools/big_probe.py --mode update --log2 17 --reps 6 2>&1 | grep "^k_big\|rc=" | cut -c1-330
# r03: the bf16 MLP kernels -- the 64-row path (k_big_fwd16 / k_big_rows16_bwd / k_big_dw16 / k_big_reduce<true>) and the 16-row bench path
# (k_qnet_fwd16, k_dw16, the bf16 k_actor)
bash $R/tools/prof_pmc_mfma.sh ${tag}_big16 $R/tools/big_probe.py --mode update --log2 17 --reps 6 --precision bf16 2>&1 | grep "k_big\|rc=" | cut -c1-330
bash $R/tools/prof_pmc_mfma.sh ${tag}_big16f $R/tools/big_probe.py --mode fwd --log2 17 --reps 6 --precision bf16 2>&1 | grep "k_big\|rc=" | cut -c1-330
bash $R/tools/prof_pmc_mfma.sh ${tag}_bench16 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --profile-steps 2 --precision bf16 2>&1 | grep "^k_\|rc=" | cut -c1-330
# r03: the sorted priority write-back by leaf segments: time per call over batch sizes, FETCH_SIZE / WRITE_SIZE per launch
bash $R/tools/prof_pmc_write.sh $tag 2>&1 | tail -8
bash $R/tools/prof_pmc_mfma.sh ${tag}_cnn $R/tools/cnn_probe.py --mode update --reps 6 2>&1 | grep "^k_cnn\|rc=" | cut -c1-330
bash $R/tools/prof_pmc_mfma.sh ${tag}_cnnf32 $R/tools/cnn_probe.py --mode update --reps 6 --precision f32 2>&1 | grep "^k_cnn\|rc=" | cut -c1-330
# r03: the trunk kernel (conv1 + conv2 + conv3 in one persistent kernel) at 4 pairs per workgroup: MFMA / wave-state and LDS counter passes
bash $R/tools/prof_pmc_mfma.sh ${tag}_trunk $R/tools/cnn_probe.py --mode forward --precision bf16 --batch 2048 2>&1 | grep "^k_cnn_trunk\|rc=" | cut -c1-400
bash $R/tools/prof_pmc_lds.sh ${tag}_trunk $R/tools/cnn_probe.py --mode forward --precision bf16 --batch 2048 2>&1 | grep "^k_cnn_trunk\|rc=" | cut -c1-400
(bash $R/tools/kt_cnn.sh ${tag}_ub --mode update; bash $R/tools/kt_cnn.sh ${tag}_uf --mode update --precision f32; bash $R/tools/kt_cnn.sh ${tag}_fb --mode forward; bash $R/tools/kt_cnn.sh ${tag}_ff --mode forward --precision f32) > $out/cnn_$tag.txt 2>&1
tail -3 $out/cnn_$tag.txt
cd $R
python tools/per_sample_probe.py --json $out/probe_$tag.json 2>&1 | tail -8
python tools/per_write_probe.py --json $out/wprobe_$tag.json 2>&1 | tail -22
python tools/sweep.py --max-log2 18 --json $out/sweep_${tag}_f32.json 2>&1 | tail -3 | cut -c1-300
python tools/sweep.py --max-log2 17 --no-actor --precision bf16 --json $out/sweep_${tag}_bf16.json 2>&1 | tail -2 | cut -c1-300
python tools/cnn_sweep.py --log2 9 11 13 --json $out/cnn_sweep_$tag.json 2>&1 | tail -8 | cut -c1-260
python tools/cnn_sweep.py --log2 9 11 13 --precision bf16 --flags 4 --json $out/cnn_sweep_${tag}_layerwise.json 2>&1 | tail -4 | cut -c1-260
# gpurun merges at most 64 MiB back: keep the summaries (json / txt / stats) and the one kernel trace collect_profiles.py reads, drop the raw counter dumps
find $out -name "*counter_collection.csv" -delete 2>/dev/null
find $out -name "*.db" -delete 2>/dev/null
for d in $out/pmc_* $out/pmcs_* $out/pmcw_* $out/pmcm_* $out/pmcl_* $out/kt_*; do [ -d "$d" ] && rm -rf "$d"; done
du -sh $out | tail -1
