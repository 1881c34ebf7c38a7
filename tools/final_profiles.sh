#!/bin/bash
# Run ON THE GPU BOX: everything profiles/<round>_final/ holds, into gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   bench_default.json         python bench.py --steps 40            (with cpu_baseline, boundary_host, shares; written AFTER the PMC passes so that roofline.traffic carries this build's counters)
#   kernel_stats.csv, bench_under_rocprof.json    rocprofv3 --kernel-trace --stats of bench.py --steps 20
#   pmc/*_per_kernel.csv, pmc_traffic.json        separate --pmc FETCH_SIZE / WRITE_SIZE passes
#   bench_C2.json, bench_C4.json, kernel_stats_C4.csv, bench_128_blocks.json, bench_C5.json, bench_C5_512_blocks.json
#   ip_loop.json, kernel_stats_ip_loop.csv    the interior-point loop at C3 dimensions (tools/ip_c3.py), plain and under rocprofv3 --stats
#   dynamic_ip_loop.json, kernel_stats_dynamic_ip_loop.csv    the loop of a time-staged problem at the C4 dimensions (tools/dynamic_ip.py)
#   burgers_ip_configuration_4.json    BASELINE configs[3] to the letter through ip_solve with the host producer (tools/burgers_ip.py)
#   ip_step_traffic.json    counter traffic of the interior-point step kernels (tools/ip_step_traffic.sh)
#   mfma_util.json, mfma_util_C4.json, mfma_util_C5.json    own --pmc SQ_VALU_MFMA_BUSY_CYCLES passes (matrix-core kernels)
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-boundary --no-ip-loop > $out/bench_under_rocprof.json 2> $out/stats.err || exit 1
cp $(find $out/stats -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
echo "kernel stats done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_mfma -- python3 bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-boundary --no-ip-loop > $out/bench_mfma.json 2> $out/mfma.err \
  && python3 tools/mfma_util.py $out/pmc_mfma $(find $out/stats -name '*kernel_trace.csv' | head -1) $out/mfma_util.json "bench.py (C3)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-boundary --no-ip-loop > $out/bench_fetch.json 2> $out/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-boundary --no-ip-loop > $out/bench_write.json 2> $out/write.err || exit 1
python3 - $out <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, 'tools')
from pmc_summary import load
out = sys.argv[1]
for tag, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    per = load(os.path.join(out, 'pmc_' + tag), counter)
    with open(os.path.join(out, 'pmc', '%s_size_per_kernel.csv' % tag), 'w') as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(['Kernel_Name', 'Dispatches', counter + '_sum_KiB'])
        for name, (v, n) in per.items():
            w.writerow([name, n, round(v, 3)])
d = json.loads(open(os.path.join(out, 'bench_fetch.json')).read().strip().splitlines()[-1])
st = d['plan']
open(os.path.join(out, 'plan_args.txt'), 'w').write('%d %d %d\n' % (st['raw_entries'], st['n'], d['config']['blocks_per_gpu']))
PY
read raw n batch < $out/plan_args.txt
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $raw $n $batch $out/pmc_traffic.json "bench.py --steps 3 --profile-steps 1 under rocprofv3 --pmc (one pass per counter)" || exit 1
echo "pmc done"
# the default line AFTER the counter passes, with their summary in place: roofline.traffic is then stamped with this build's hash
cp $out/pmc_traffic.json profiles/pmc_traffic.json
python3 bench.py --steps 40 > $out/bench_default.json 2> $out/bench_default.err || { tail -3 $out/bench_default.err; exit 1; }
echo "default done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_ip -- python3 tools/ip_c3.py 1024 > $out/ip_loop_under_rocprof.json 2> $out/stats_ip.err && cp $(find $out/stats_ip -name '*kernel_stats.csv' | head -1) $out/kernel_stats_ip_loop.csv
python3 tools/ip_c3.py 1024 > $out/ip_loop.json 2> $out/ip_loop.err
bash tools/ip_step_traffic.sh $tag/ip_traffic > $out/ip_step_traffic.log 2>&1 && cp $out/ip_traffic/ip_step_traffic.json $out/ip_step_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_dyn -- python3 tools/dynamic_ip.py 512 49 2 40 > $out/dynamic_ip_loop_under_rocprof.json 2> $out/stats_dyn.err && cp $(find $out/stats_dyn -name '*kernel_stats.csv' | head -1) $out/kernel_stats_dynamic_ip_loop.csv
python3 tools/dynamic_ip.py 512 49 2 40 > $out/dynamic_ip_loop.json 2> $out/dynamic_ip_loop.err
python3 tools/burgers_ip.py 512 50 40 > $out/burgers_ip_configuration_4.json 2> $out/burgers_ip.err
python3 bench.py --workload C2 --no-cpu-baseline > $out/bench_C2.json 2> $out/c2.err
python3 bench.py --blocks 128 --no-cpu-baseline --no-boundary --no-ip-loop > $out/bench_128_blocks.json 2> $out/b128.err
python3 bench.py --workload C4 --no-cpu-baseline --steps 10 --warmup 2 > $out/bench_C4.json 2> $out/c4.err
bash tools/pmc_traffic_run.sh $tag/pmc_C4 --workload C4 --no-ip-loop > $out/pmc_traffic_C4.log 2>&1 && cp $out/pmc_C4/pmc_traffic.json $out/pmc_traffic_C4.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_C4 -- python3 bench.py --workload C4 --steps 5 --warmup 2 --no-cpu-baseline --no-boundary --no-ip-loop > $out/bench_C4_under_rocprof.json 2> $out/stats_C4.err && cp $(find $out/stats_C4 -name '*kernel_stats.csv' | head -1) $out/kernel_stats_C4.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_mfma_C4 -- python3 bench.py --workload C4 --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-boundary --no-ip-loop > $out/bench_mfma_C4.json 2> $out/mfma_C4.err \
  && python3 tools/mfma_util.py $out/pmc_mfma_C4 $(find $out/stats_C4 -name '*kernel_trace.csv' | head -1) $out/mfma_util_C4.json "bench.py --workload C4"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_C5 -- python3 bench.py --workload C5 --blocks 512 --steps 5 --warmup 2 --value-sets 2 --no-cpu-baseline --no-boundary --no-ip-loop > $out/bench_C5_under_rocprof.json 2> $out/stats_C5.err && cp $(find $out/stats_C5 -name '*kernel_stats.csv' | head -1) $out/kernel_stats_C5_512_blocks.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_mfma_C5 -- python3 bench.py --workload C5 --blocks 512 --steps 3 --warmup 1 --value-sets 2 --profile-steps 1 --no-cpu-baseline --no-boundary --no-ip-loop > $out/bench_mfma_C5.json 2> $out/mfma_C5.err \
  && python3 tools/mfma_util.py $out/pmc_mfma_C5 $(find $out/stats_C5 -name '*kernel_trace.csv' | head -1) $out/mfma_util_C5.json "bench.py --workload C5 --blocks 512"
python3 bench.py --workload C5 --no-cpu-baseline --no-boundary --no-ip-loop --steps 5 --warmup 2 --value-sets 2 --profile-steps 2 > $out/bench_C5.json 2> $out/c5.err
python3 bench.py --workload C5 --blocks 512 --no-cpu-baseline --no-boundary --no-ip-loop --steps 10 --warmup 2 --value-sets 2 --profile-steps 2 > $out/bench_C5_512_blocks.json 2> $out/c5b.err
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
rm -rf $out/stats $out/stats_ip $out/stats_dyn $out/stats_C4 $out/stats_C5 $out/pmc_fetch $out/pmc_write $out/pmc_mfma $out/pmc_mfma_C4 $out/pmc_mfma_C5
for f in bench_default bench_C2 bench_128_blocks bench_C4 bench_C5 bench_C5_512_blocks; do python3 - $out/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d['value'], d['unit'], d['ms_per_step'], 'correct', d.get('correct'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
