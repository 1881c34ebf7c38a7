#!/bin/bash
# Run ON THE GPU BOX: kernel trace of a short bench run of a workload, per-stream timeline of its last full step
# usage: tools/c4_trace.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-boundary --no-ip-loop --no-ip-loop-dynamic "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
f=$(find $out/stats -name '*kernel_trace.csv' | head -1)
head -1 $f > $out/trace_header.txt
python3 tools/trace_streams.py $f 700 > $out/timeline_streams.txt
find $out -name '*kernel_trace.csv' -size +8M -delete
cp $(find $out/stats -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
head -40 $out/timeline_streams.txt
