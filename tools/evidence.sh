#!/bin/bash
# round evidence, one GPU box (usage: bash tools/evidence.sh r6): rocprofv3 kernel stats + timeline of `python bench.py` (default arguments,
# --config script, --config 5), PMC HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), the bench lines (default with parity + cpu_baseline,
# 100 steps, script, script at batch 40, config 5, conditional, ragged lengths, through the trainer), the non-profiled main-stream phase times,
# and the data-parallel rehearsals one GPU allows: a ONE-rank RCCL process group (every collective through RCCL) and two ranks sharing the
# GPU over gloo.  Everything lands under gpurun_out/; the summaries worth keeping are copied into profiles/ by hand.
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
NP="--no-cpu-baseline --no-parity"
timeout -k 10 300 bash tools/prof.sh ${TAG}final --no-parity > gpurun_out/${TAG}_prof.txt 2>&1 &&
python tools/timeline.py gpurun_out/prof_${TAG}final > gpurun_out/${TAG}_timeline.txt 2>&1 &&
cp $(find gpurun_out/prof_${TAG}final -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv &&
timeout -k 10 300 bash tools/prof.sh ${TAG}script --no-parity --config script > gpurun_out/${TAG}_prof_script.txt 2>&1 &&
python tools/timeline.py gpurun_out/prof_${TAG}script > gpurun_out/${TAG}_timeline_script.txt 2>&1 &&
timeout -k 10 400 bash tools/prof.sh ${TAG}c5 --no-parity --config 5 --steps 6 --warmup 2 > gpurun_out/${TAG}_prof_c5.txt 2>&1 &&
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err &&
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_driver_args.json 2>/dev/null &&
timeout -k 10 300 python bench.py --steps 100 --warmup 10 $NP > gpurun_out/${TAG}_bench_100.json 2>/dev/null &&
timeout -k 10 300 python bench.py --config script $NP > gpurun_out/${TAG}_bench_script.json 2>/dev/null &&
timeout -k 10 300 python bench.py --config script --batch 40 $NP > gpurun_out/${TAG}_bench_script_b40.json 2>/dev/null &&
timeout -k 10 300 python bench.py --config 5 --steps 10 --warmup 3 $NP > gpurun_out/${TAG}_bench_cfg5.json 2>/dev/null &&
timeout -k 10 300 python bench.py --n-img 290000 $NP > gpurun_out/${TAG}_bench_cfg4_table.json 2>/dev/null &&
timeout -k 10 300 python bench.py --conditional --steps 50 --warmup 10 $NP > gpurun_out/${TAG}_bench_cond.json 2>/dev/null &&
timeout -k 10 300 python bench.py --config script --conditional $NP > gpurun_out/${TAG}_bench_script_cond.json 2>/dev/null &&
timeout -k 10 300 python bench.py --lengths ragged $NP > gpurun_out/${TAG}_bench_ragged.json 2>/dev/null &&
timeout -k 10 300 python bench.py --through-trainer > gpurun_out/${TAG}_bench_trainer.json 2>/dev/null &&
timeout -k 10 300 python bench.py --through-trainer --config script > gpurun_out/${TAG}_bench_trainer_script.json 2>/dev/null &&
timeout -k 10 300 python tools/phase_times.py > gpurun_out/${TAG}_phase_times.txt 2>&1 &&
timeout -k 10 300 python tools/hbm_kernels.py > gpurun_out/${TAG}_hbm_kernels.txt 2>&1 &&
timeout -k 10 300 python tools/lazy_rows_bench.py > gpurun_out/${TAG}_lazy_rows_kernels.txt 2>&1 &&
timeout -k 10 300 python tools/critical.py > gpurun_out/${TAG}_critical.txt 2>&1 &&
(for t in dwg_layouts gemm_square wgrad_gemm gemm_vs_blas; do echo "== python tools/$t.py"; timeout -k 10 300 python tools/$t.py 2>&1 | grep -v amdgpu.ids; echo; done) > gpurun_out/${TAG}_gemm_products_alone.txt 2>&1 &&
VMMT_DP_FORCE=1 VMMT_DP_DIRECT=1 timeout -k 10 300 python bench.py $NP > gpurun_out/${TAG}_bench_rccl_world1.json 2> gpurun_out/${TAG}_bench_rccl_world1.err &&
VMMT_BENCH_ONE_GPU=1 VMMT_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 $NP > gpurun_out/${TAG}_bench_2ranks_one_gpu_gloo.json 2> gpurun_out/${TAG}_bench_2ranks.err &&
# the PMC passes LAST: profiles/traffic.json must be measured on the kernels the lines above ran (tools/traffic_key.py merge gpurun_out/traffic_${TAG}final_config2.json)
timeout -k 10 400 bash tools/pmc.sh ${TAG}final --no-parity > gpurun_out/${TAG}_pmc.txt 2>&1 &&
timeout -k 10 500 bash tools/pmc.sh ${TAG}c5 --no-parity --config 5 > gpurun_out/${TAG}_pmc_c5.txt 2>&1
echo "rc=$?"
for f in default driver_args 100 script script_b40 cfg5 cfg4_table cond script_cond ragged trainer trainer_script rccl_world1 2ranks_one_gpu_gloo; do echo "$f: $(tail -n 1 gpurun_out/${TAG}_bench_$f.json | cut -c1-150)"; done
