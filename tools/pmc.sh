#!/bin/bash
# HBM traffic of the kernels via PMC counters, one counter per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not
# fit one pass; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled in tools/pmc_summary.py)
TAG=$1; shift          # further arguments go to bench.py (e.g. --config 5)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python $R/bench.py --no-cpu-baseline --no-parity --no-overlap-probe --repeats 1 --steps 3 --warmup 2 "$@" > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1
done
CFG=2; for a in "$@"; do if [ "$prev" = "--config" ]; then CFG=$a; fi; prev=$a; done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG}_FETCH_SIZE $R/gpurun_out/pmc_${TAG}_WRITE_SIZE --fragment $R/gpurun_out/traffic_${TAG}_config$CFG.json --config $CFG | tee $R/gpurun_out/pmc_${TAG}_summary.txt
