#!/bin/bash
# Diagnostic builds of the round-5 streaming encoder (csrc/enc_stream2.hip) with compile-time ablations -> ab/libmmif_e2_<n>.so
# (timing only: results are WRONG; tools/bench_enc.py runs every ab/lib*.so in a child process).  Usage: tools/build_ab_enc2.sh 1 2 4 8 ...
set -e
cd "$(dirname "$0")/../multi-modal-image-fusion_amd/csrc"
mkdir -p ../../ab
OBJS=$(ls *.o | grep -v '^enc_stream2.o$')
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DE2_ABL=$n -c enc_stream2.hip -o /tmp/enc_stream2_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/enc_stream2_abl$n.o -o ../../ab/libmmif_e2_$n.so
  echo "built ab/libmmif_e2_$n.so"
done
