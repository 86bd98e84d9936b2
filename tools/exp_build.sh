#!/bin/bash
# builds experiment variants of the library (one macro each) next to libvmmt.so: tools/exp_build.sh NOMFMA NOREAD NODMA
cd $(dirname $0)/..
for V in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ivariational_mmt_amd/csrc -Wno-unused-result -DVMMT_EXP_$V -c variational_mmt_amd/csrc/gemm.hip -o /tmp/gemm_$V.o || exit 1
  OBJS=$(ls variational_mmt_amd/csrc/build/*.o | grep -v "gemm")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variational_mmt_amd/libvmmt_exp_$V.so /tmp/gemm_$V.o $OBJS || exit 1
  echo built $V
done
