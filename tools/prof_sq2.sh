#!/bin/bash
# SQ counters for an arbitrary bench config:  tools/prof_sq2.sh <tag> <pmc list in quotes> [bench args]
# (under `timeout`: an unsupported counter combination makes rocprofv3 abort and then hang in its finaliser -- a TA_* / TCP_* set of 8 cost
# a whole 20-minute gpurun limit once; on this image the TA_* / TCP_* counters abort rocprofv3 (signal 6) even four at a time: SQ_* and
# the TCC_* sets of tools/prof_pmc.sh work)
tag=$1; pmc=$2; shift; shift
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
rm -rf gpurun_out/sq_$tag
timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d gpurun_out/sq_$tag -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-path --no-other-configs "$@" > gpurun_out/sq_$tag.log 2>&1
csv=$(find gpurun_out/sq_$tag -name "*counter_collection.csv" | head -1)
python3 tools/pmc_stats.py $csv > gpurun_out/sq_$tag.txt
grep -i "kernel\|pairconv" gpurun_out/sq_$tag.txt | head -12
find gpurun_out/sq_$tag -name "*.csv" -size +1M -delete
