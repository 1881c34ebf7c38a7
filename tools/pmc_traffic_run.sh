#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE passes of bench.py with the given flags -> gpurun_out/<tag>/pmc_traffic.json
# usage: tools/pmc_traffic_run.sh <tag> [bench flags...]
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-boundary "$@" > $out/bench_fetch.json 2> $out/fetch.err || { tail -3 $out/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-boundary "$@" > $out/bench_write.json 2> $out/write.err || { tail -3 $out/write.err; exit 1; }
python3 - $out <<'PY'
import json, os, sys
out = sys.argv[1]
d = json.loads(open(os.path.join(out, 'bench_fetch.json')).read().strip().splitlines()[-1])
st = d['plan']
open(os.path.join(out, 'plan_args.txt'), 'w').write('%d %d %d\n' % (st['raw_entries'], st['n'], d['config']['blocks_per_gpu']))
PY
read raw n batch < $out/plan_args.txt
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $raw $n $batch $out/pmc_traffic.json "bench.py $* --steps 3 --profile-steps 1 under rocprofv3 --pmc (one pass per counter)" || exit 1
python3 - $out/pmc_traffic.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d['phases'].items()):
    print('  %-16s %8.3f GB/step  %6.0f launches' % (k, v['hbm_bytes_per_step'] / 1e9, v['launches_per_step']))
for k, v in sorted(d['kernels'].items(), key=lambda kv: -(kv[1]['hbm_read_bytes_per_launch'] + kv[1]['hbm_write_bytes_per_launch']) * kv[1]['launches_per_step'])[:12]:
    print('  %-28s %6.1f launches/step  read %8.2f MB  write %8.2f MB per launch' % (k, v['launches_per_step'], v['hbm_read_bytes_per_launch'] / 1e6, v['hbm_write_bytes_per_launch'] / 1e6))
PY
find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -delete; rm -rf $out/pmc_fetch $out/pmc_write
