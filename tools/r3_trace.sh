#!/bin/bash
# Run ON THE GPU BOX: kernel trace of a short bench run, per-launch timeline of one iteration (tools/trace_levels.py)
# usage: tools/r3_trace.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-boundary --no-ip-loop --profile-steps 1 "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
f=$(find $out/stats -name '*kernel_trace.csv' | head -1)
python3 tools/trace_levels.py $f 120 > $out/timeline.txt
find $out -name '*kernel_trace.csv' -size +8M -delete
cp $(find $out/stats -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
tail -3 $out/timeline.txt
