run() { env "$@" python bench.py --no-cpu-baseline --no-parity --steps 100 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do
echo "all-new $(run X=1) | no-group $(run VMMT_GROUP_WGRADS=0) | no-zxt $(run VMMT_Z_IN_XT=0) | dxt-aux $(run VMMT_DXT_ON_SIDE=0) | norm-late $(run VMMT_GEN_NORM_EARLY=0) | all-old $(run VMMT_GROUP_WGRADS=0 VMMT_Z_IN_XT=0 VMMT_DXT_ON_SIDE=0 VMMT_GEN_NORM_EARLY=0)"
done
