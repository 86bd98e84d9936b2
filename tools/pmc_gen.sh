#!/bin/bash
# usage: bash tools/pmc_gen.sh [tag]   -- counters of the generator kernels of tools/gen_one.py (gen2 = fused sweep; fwd / bwd = gen_kernel_q<0/1>),
# one counter set per pass
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=${1:-r2}
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmcgen_${V}_$TAG -- python $R/tools/gen_one.py > $R/gpurun_out/pmcgen_${V}_$TAG.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob('$R/gpurun_out/pmcgen_${V}_$TAG/*/*counter_collection.csv')
agg = collections.defaultdict(list)
if not f: print("no counter file for $TAG")
else:
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'gen_kernel' in k or 'gen2_kernel' in k or 'gen2p_kernel' in k:
            mode = 'gen2' if ('gen2_kernel' in k or 'gen2p_kernel' in k) else ('bwd' if 'gen_kernel_q<1>' in k else 'fwd')
            agg[(mode, r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()): print("%-4s %-30s %16.0f" % (k[0], k[1], sum(v[2:]) / max(1, len(v[2:]))))
PY
done
