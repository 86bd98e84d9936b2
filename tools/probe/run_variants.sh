for v in V1 V2 V3 V4; do echo "== $v"; VMMT_LIB_PATH=variational_mmt_amd/libvmmt_exp_$v.so python tools/gen_one.py 2>&1 | grep fused; done
echo "== PROBE(V1)"; VMMT_LIB_PATH=variational_mmt_amd/libvmmt_exp_PROBE.so python tools/gen_one.py 2>&1 | tail -2
