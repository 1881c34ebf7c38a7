#!/bin/bash
# Run ON THE GPU BOX: the secondary workloads of the bench on the current build -> gpurun_out/<tag>/*.json + one summary line each
tag=${1:-r06_workloads}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; python3 bench.py --no-cpu-baseline --no-boundary --no-ip-loop --no-shares "$@" > $out/$name.json 2> $out/$name.err || { tail -3 $out/$name.err; return 1; }
  python3 - $out/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
sc = d.get('solution_check') or {}
print(sys.argv[2], 'value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 4), 'no_prefetch_ms', round(d['ms_per_step_no_prefetch'], 4),
      'unchecked_ms', round(sc.get('ms_per_step_unchecked', 0), 4), 'correct', d['correct'], 'launches', d['kernel_launches_per_step'],
      {k: round(v['ms_per_step'], 3) for k, v in d['phases'].items()})
PY
}
run C3_128 --blocks 128
run C2 --workload C2
run C4 --workload C4 --steps 10 --warmup 2
run C5_512 --workload C5 --blocks 512 --steps 10 --warmup 2 --value-sets 2
