#!/bin/bash
# experiment variants of the GENERATOR kernels (one macro each): tools/exp_build_gen.sh NOMFMA NOREAD NODMA
cd $(dirname $0)/..
for V in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ivariational_mmt_amd/csrc -Wno-unused-result -DVMMT_EXP_$V -c variational_mmt_amd/csrc/generator.hip -o /tmp/generator_$V.o || exit 1
  OBJS=$(ls variational_mmt_amd/csrc/build/*.o | grep -v generator.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variational_mmt_amd/libvmmt_exp_gen$V.so /tmp/generator_$V.o $OBJS || exit 1
  echo built $V
done
