#!/bin/bash
# GPU box: C4 bench under PP_PLAN_TUNE variants
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  PP_PLAN_TUNE="$t" python3 bench.py --workload C4 --no-cpu-baseline --no-boundary --no-ip-loop --steps 30 > gpurun_out/c4_tune.json 2>/dev/null
  echo "== $t"; python3 tools/show_bench.py gpurun_out/c4_tune.json | cut -c1-120
done
