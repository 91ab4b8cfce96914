#!/bin/bash
# Build timing-experiment variants of ncw_pp.hip only (other objects reused) as libneuconw_hip_<tag>.so
# usage: scripts/pp_variants.sh tag1:"-DFLAG ..." tag2:"..."
set -e
cd "$(dirname "$0")/../neuralrecon-w_amd"
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value $flags -c csrc/ncw_pp.hip -o /tmp/ncw_pp_$tag.o
  objs=$(ls csrc/build/*.o | grep -v ncw_pp.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libneuconw_hip_$tag.so $objs /tmp/ncw_pp_$tag.o
  echo built libneuconw_hip_$tag.so
done
