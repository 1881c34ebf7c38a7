#!/bin/bash
# Run ON THE GPU BOX (gpurun): kernel-trace stats + separate PMC passes of the default bench workload.
# usage: tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/{stats,pmc_fetch,pmc_write}/...
set -o pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-boundary > $out/bench_stats.json 2> $out/bench_stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-boundary > $out/bench_fetch.json 2> $out/bench_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-boundary > $out/bench_write.json 2> $out/bench_write.err || exit 1
# keep only the small summaries (traces of every dispatch are large)
find $out -name '*kernel_trace.csv' -size +8M -delete
ls -la $out/*/*/ | head -40
