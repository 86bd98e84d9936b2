#!/bin/bash
# PMC counter passes over one large TN product (tools/gemm_one.py): bash tools/pmc_gemm.sh <tag> [M N K]
TAGX=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmcg_${TAGX}_$TAG -- python3 $R/tools/gemm_one.py "$@" > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob('$R/gpurun_out/pmcg_${TAGX}_$TAG/*/*counter_collection.csv')
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'gemm_kernel' in r['Kernel_Name'] or 'tn256' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items(): print("%-28s %14.0f" % (k, sum(v[2:]) / max(1, len(v[2:]))))
PY
done
