#!/bin/bash
# Build side-by-side variants of libvf_hip.so that differ in -D flags of ONE source file (kernel tuning).
# usage: bash tools/variants.sh <source stem, e.g. conv3_halo_x6> <name>:"<-D flags>" ...
# -> viewformer_amd/variants/libvf_<name>.so ; run with VF_HIP_LIB=<path> python tools/microbench.py ...
set -eu
cd "$(dirname "$0")/.."
python -m viewformer_amd.build > /dev/null
STEM=$1; shift
mkdir -p viewformer_amd/variants
OBJS=$(ls viewformer_amd/build/*.o | grep -v "/$STEM.o")
BASE=$(python -c "from viewformer_amd.build import EXTRA_FLAGS; print(' '.join(EXTRA_FLAGS.get('$STEM', [])))")
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $BASE $flags -c viewformer_amd/csrc/$STEM.hip -o viewformer_amd/variants/$STEM.$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o viewformer_amd/variants/libvf_$name.so $OBJS viewformer_amd/variants/$STEM.$name.o &&
    echo built $name ) &
done
wait
