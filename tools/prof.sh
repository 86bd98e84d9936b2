#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> [bench args...]
# runs rocprofv3 --kernel-trace --stats on bench.py and prints a per-kernel summary (also saved under gpurun_out/)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --no-cpu-baseline --repeats 1 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
grep '"metric"' $R/gpurun_out/prof_$TAG.log
NST=$(python -c "import sys; a=sys.argv[1:]; g=lambda k,d: int(a[a.index(k)+1]) if k in a else d; print(g('--steps',50)+max(8,g('--warmup',10)))" "$@")
python $R/tools/prof_summary.py $(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) $NST | tee $R/gpurun_out/prof_${TAG}_summary.txt
