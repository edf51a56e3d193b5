#!/bin/bash
# Diagnostic builds of the fused encoder backward (csrc/enc_bwd.hip) with compile-time ablations -> ab/libmmif_eb_<n>.so
# (timing only: results are WRONG; tools/bench_encbwd.py runs every ab/libmmif_eb_*.so in a child process).  Usage: tools/build_ab_encbwd.sh 1 2 4 ...
set -e
cd "$(dirname "$0")/../multi-modal-image-fusion_amd/csrc"
mkdir -p ../../ab
OBJS=$(ls *.o | grep -v '^enc_bwd.o$')
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFB_ABL=$n -c enc_bwd.hip -o /tmp/enc_bwd_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/enc_bwd_abl$n.o -o ../../ab/libmmif_eb_$n.so
  echo "built ab/libmmif_eb_$n.so"
done
