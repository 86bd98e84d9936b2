#!/bin/bash
# round-2 evidence, one GPU box: rocprofv3 kernel stats + timeline of `python bench.py` (default arguments), PMC HBM traffic
# (FETCH_SIZE / WRITE_SIZE, separate passes), and the bench lines (default, --config 5, --conditional).  Everything lands under
# gpurun_out/; the summaries worth keeping are copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 300 bash tools/prof.sh r2final > gpurun_out/r2_prof.txt 2>&1 &&
python tools/timeline.py gpurun_out/prof_r2final > gpurun_out/r2_timeline.txt 2>&1 &&
cp $(find gpurun_out/prof_r2final -name "*kernel_stats.csv" | head -1) gpurun_out/r2_kernel_stats.csv &&
timeout -k 10 400 bash tools/pmc.sh r2final > gpurun_out/r2_pmc.txt 2>&1 &&
timeout -k 10 400 python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err &&
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r2_bench_100.json 2>/dev/null &&
timeout -k 10 300 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_cfg5.json 2>/dev/null &&
timeout -k 10 300 python bench.py --conditional --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r2_bench_cond.json 2>/dev/null
echo "rc=$?"
tail -n 1 gpurun_out/r2_bench_default.json | cut -c1-400
tail -n 1 gpurun_out/r2_bench_100.json | cut -c1-200
tail -n 1 gpurun_out/r2_bench_cfg5.json | cut -c1-300
tail -n 1 gpurun_out/r2_bench_cond.json | cut -c1-200
