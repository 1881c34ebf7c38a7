#!/bin/bash
# Run ON THE GPU BOX: the default C3 bench line (no CPU baseline / boundary / IP loop) for several rows-per-wave of the
# residual kernel (csrc/refine.hip, switch PP_RES_ROWS) -> gpurun_out/<tag>/rows_<n>.json
tag=${1:-res_rows}; out=gpurun_out/$tag; mkdir -p $out
for r in 2 4 8 16 32; do
  PP_RES_ROWS=$r python3 bench.py --no-cpu-baseline --no-boundary --no-ip-loop > $out/rows_$r.json 2> $out/rows_$r.err || { tail -3 $out/rows_$r.err; exit 1; }
  python3 - $out/rows_$r.json $r <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('rows', sys.argv[2], 'value', round(d['value'], 1), 'no_prefetch', round(d.get('value_no_prefetch', 0), 1), 'ms', round(d['ms_per_step'], 4))
PY
done
