#!/bin/bash
# round evidence, one GPU box (usage: bash tools/evidence.sh r3): rocprofv3 kernel stats + timeline of `python bench.py` (default arguments
# and --config script), PMC HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), and the bench lines (default, script, config 5,
# conditional, ragged lengths, through the trainer, 2-rank rehearsal on one GPU).  Everything lands under gpurun_out/; the summaries worth
# keeping are copied into profiles/ by hand.
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 300 bash tools/prof.sh ${TAG}final > gpurun_out/${TAG}_prof.txt 2>&1 &&
python tools/timeline.py gpurun_out/prof_${TAG}final > gpurun_out/${TAG}_timeline.txt 2>&1 &&
cp $(find gpurun_out/prof_${TAG}final -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv &&
timeout -k 10 300 bash tools/prof.sh ${TAG}script --config script > gpurun_out/${TAG}_prof_script.txt 2>&1 &&
python tools/timeline.py gpurun_out/prof_${TAG}script > gpurun_out/${TAG}_timeline_script.txt 2>&1 &&
timeout -k 10 400 bash tools/pmc.sh ${TAG}final > gpurun_out/${TAG}_pmc.txt 2>&1 &&
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err &&
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench_100.json 2>/dev/null &&
timeout -k 10 300 python bench.py --config script --no-cpu-baseline > gpurun_out/${TAG}_bench_script.json 2>/dev/null &&
timeout -k 10 300 python bench.py --config script --batch 40 --no-cpu-baseline > gpurun_out/${TAG}_bench_script_b40.json 2>/dev/null &&
timeout -k 10 300 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg5.json 2>/dev/null &&
timeout -k 10 300 python bench.py --conditional --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench_cond.json 2>/dev/null &&
timeout -k 10 300 python bench.py --lengths ragged --no-cpu-baseline > gpurun_out/${TAG}_bench_ragged.json 2>/dev/null &&
timeout -k 10 300 python bench.py --through-trainer > gpurun_out/${TAG}_bench_trainer.json 2>/dev/null &&
timeout -k 10 300 python bench.py --through-trainer --config script > gpurun_out/${TAG}_bench_trainer_script.json 2>/dev/null &&
VMMT_BENCH_ONE_GPU=1 VMMT_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_2ranks_one_gpu_gloo.json 2> gpurun_out/${TAG}_bench_2ranks.err
echo "rc=$?"
for f in default 100 script script_b40 cfg5 cond ragged trainer trainer_script 2ranks_one_gpu_gloo; do tail -n 1 gpurun_out/${TAG}_bench_$f.json | cut -c1-160; done
