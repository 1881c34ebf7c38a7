#!/bin/bash
# Run ON THE GPU BOX: kernel trace of a short default bench run and the per-stream timeline of its last full step
# (tools/trace_streams.py): which launches sit on the critical stream, their durations and the gaps between them.
# usage: tools/step_timeline.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-boundary --no-ip-loop --no-shares "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
f=$(find $out/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_streams.py $f 200 > $out/timeline.txt
python3 tools/trace_streams.py $f 200 6 > $out/timeline_timed_step.txt
rm -rf $out/trace
tail -2 $out/timeline.txt
