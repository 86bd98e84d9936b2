#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> [bench args...]
# runs rocprofv3 --kernel-trace --stats on bench.py and prints a per-kernel summary (also saved under gpurun_out/)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# (--no-overlap-probe: bench.py's stream-overlap self-check runs 13 extra steps, 6 of them on ONE stream -- they would be averaged into the
#  profile; --no-parity: the parity gate runs a cfg-1-shaped step of its own.  The divisor is what bench.py says it executed: "steps_executed")
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --no-cpu-baseline --no-parity --no-overlap-probe --repeats 1 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
grep '"metric"' $R/gpurun_out/prof_$TAG.log
NST=$(grep '"metric"' $R/gpurun_out/prof_$TAG.log | python -c "import sys, json; print(json.loads(sys.stdin.read())['steps_executed']['steps'])")
python $R/tools/prof_summary.py $(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) $NST | tee $R/gpurun_out/prof_${TAG}_summary.txt
