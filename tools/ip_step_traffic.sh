#!/bin/bash
# Run ON THE GPU BOX: HBM traffic of the interior-point step kernels (k_ip_*) from separate --pmc FETCH_SIZE / WRITE_SIZE passes
# of tools/ip_c3.py (the loop at C3 dimensions), per launch, next to the bytes each kernel has to move once (bench.py: ip_loop.step_kernels).
#   tools/ip_step_traffic.sh [out dir under gpurun_out]      ->  <out>/ip_step_traffic.json
out=gpurun_out/${1:-ip_traffic}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 tools/ip_c3.py 1024 > $out/fetch.json 2> $out/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 tools/ip_c3.py 1024 > $out/write.json 2> $out/write.err || exit 1
python3 - $out <<'PY'
import json, os, sys
sys.path.insert(0, 'tools')
from pmc_summary import load, short
out = sys.argv[1]
fetch, write = load(os.path.join(out, 'fetch'), 'FETCH_SIZE'), load(os.path.join(out, 'write'), 'WRITE_SIZE')
cal = json.load(open('profiles/pmc_traffic.json')).get('fetch_calibration', 1.0)     # gfx950: FETCH_SIZE under-reports coalesced reads
# algorithmic MB per launch (every array a kernel has to read or write once; 1024 scenarios x (5000 + 0) variables, 4000 + 200 rows)
n, mi, me, nfs, B = 5000, 0, 4000, 200, 1024
nnzH, nnzAe = 4000, 4000 + 12 * 1000 - 8        # (values: diagonal Hessian on y, [I | -A] with 4 tridiagonal blocks)
alg = {'k_ip_rhs': 5.0 * (n + mi), 'k_ip_stats': 6.0 * (n + mi), 'k_ip_step': 10.0 * (n + mi) + 3.0 * mi + 3.0 * (me + nfs),
       'k_ip_rows': nnzH + 2.0 * nnzAe + (n + me + 2 * mi + nfs) + (n + me) + 3.0 * n + n + (me + mi + nfs)}
res = {}
for name in sorted(set(fetch) | set(write)):
    s = short(name)
    if not s.startswith('k_ip_'):
        continue
    f, nf = fetch.get(name, [0.0, 0])
    w, nw = write.get(name, [0.0, 0])
    rd = cal * f / nf * 1024 / 1e6 if nf else None
    wr = w / nw * 1024 / 1e6 if nw else None
    e = {'launches': max(nf, nw), 'read_MB_per_launch': rd, 'write_MB_per_launch': wr}
    key = next((k for k in alg if s.startswith(k) and not s.startswith('k_ip_stats_final')), None)
    if key is not None and rd is not None and wr is not None:
        e['algorithmic_MB_per_launch'] = 8.0 * alg[key] * B / 1e6
        e['traffic_over_algorithmic'] = (rd + wr) / e['algorithmic_MB_per_launch']
    res[s] = e
json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/ip_c3.py 1024; read side x %.2f (calibration of profiles/pmc_traffic.json)' % cal,
           'kernels': res}, open(os.path.join(out, 'ip_step_traffic.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -delete; rm -rf $out/fetch $out/write
