#!/bin/bash
# builds experiment variants of the library (one macro each) next to libvmmt.so: tools/exp_build.sh <source.hip> MACRO [MACRO ...]
#   EXTRA="-DG2_PD=8" adds compiler flags.
#   e.g. tools/exp_build.sh generator_fused.hip G2PROBE   ->  variational_mmt_amd/libvmmt_exp_G2PROBE.so  (load with VMMT_LIB_PATH=...)
cd $(dirname $0)/..
SRC=$1; shift
BASE=$(basename $SRC .hip)
for V in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ivariational_mmt_amd/csrc -Wno-unused-result -DVMMT_EXP_$V $EXTRA -c variational_mmt_amd/csrc/$SRC -o /tmp/${BASE}_$V.o || exit 1
  OBJS=$(ls variational_mmt_amd/csrc/build/*.o | grep -v "/$BASE.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variational_mmt_amd/libvmmt_exp_$V.so /tmp/${BASE}_$V.o $OBJS || exit 1
  echo built $V
done
