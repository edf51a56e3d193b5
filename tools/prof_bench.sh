#!/bin/bash
# rocprofv3 kernel trace of the default bench (run on the GPU box from the repo root):  tools/prof_bench.sh <tag> [bench args]
# -> gpurun_out/prof_<tag>/ (rocpd db) + gpurun_out/kstats_<tag>.txt (per-kernel table, tools/rocpd_stats.py)
tag=$1; shift
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs "$@" > gpurun_out/prof_$tag.log 2>&1
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python3 tools/rocpd_stats.py $db 13 > gpurun_out/kstats_$tag.txt
head -40 gpurun_out/kstats_$tag.txt
