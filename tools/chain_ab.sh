#!/bin/bash
# GPU box: per-grid-size durations of k_chain_front for library variants (PP_LIB_VARIANT), C4 workload
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $1; do
  if [ $v = base ]; then unset PP_LIB_VARIANT; else export PP_LIB_VARIANT=$v; fi
  rm -rf gpurun_out/cab_$v
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cab_$v -- python3 bench.py --workload C4 --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-boundary --no-ip-loop > gpurun_out/cab_$v.json 2>/dev/null
  echo "== $v"
  python3 - gpurun_out/cab_$v <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    for key in ("k_chain_front", "k_chain_xpose<false>", "k_chain_xpose<true>", "k_gather_tiles"):
        if key in n:
            d[(key, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(d): print('  ', k, len(d[k]), 'avg %.1f us' % (sum(d[k]) / len(d[k])))
PY
  rm -rf gpurun_out/cab_$v
done
