#!/bin/bash
# Build ONE side-by-side library whose listed sources are compiled with extra -D flags (a switch that spans several files).
# usage: bash tools/variant_multi.sh <name> "<-D flags>" <stem> [<stem> ...]   -> viewformer_amd/variants/libvf_<name>.so
set -eu
cd "$(dirname "$0")/.."
python -m viewformer_amd.build > /dev/null
NAME=$1; FLAGS=$2; shift 2
mkdir -p viewformer_amd/variants
OBJS=$(ls viewformer_amd/build/*.o)
for STEM in "$@"; do
  OBJS=$(echo "$OBJS" | grep -v "/$STEM.o")
  BASE=$(python -c "from viewformer_amd.build import EXTRA_FLAGS; print(' '.join(EXTRA_FLAGS.get('$STEM', [])))")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $BASE $FLAGS -c viewformer_amd/csrc/$STEM.hip -o viewformer_amd/variants/$STEM.$NAME.o &
done
wait
VOBJS=""
for STEM in "$@"; do VOBJS="$VOBJS viewformer_amd/variants/$STEM.$NAME.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o viewformer_amd/variants/libvf_$NAME.so $OBJS $VOBJS
echo built viewformer_amd/variants/libvf_$NAME.so
