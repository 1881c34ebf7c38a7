#!/bin/bash
out=gpurun_out/r06_rows128; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 128 256 512; do
for r in 1 2 4 8 16; do
  export PP_RES_ROWS=$r
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/s_${b}_$r -- python3 bench.py --blocks $b --steps 20 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-boundary --no-ip-loop --no-shares > $out/b_${b}_$r.json 2>/dev/null
  python3 - $out/s_${b}_$r $out/b_${b}_$r.json $b $r <<'PY'
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
k = {r['Name']: float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(f))}
res = [v for n, v in k.items() if 'k_residual<' in n]
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print('blocks', sys.argv[3], 'rows', sys.argv[4], 'k_residual %.1f us' % (res[0] if res else -1), 'value', round(d['value'], 1), 'unchecked', round(d['value_unchecked'], 1))
PY
  rm -rf $out/s_${b}_$r
done
done
