#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass per counter group over a short bench run, then per-(kernel, grid)
# averages.  usage: tools/pmc_metrics.sh <tag> [bench args]   -> gpurun_out/<tag>/metrics.txt
tag=${1:-metrics}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
# counter groups: one per line of $PMC_GROUPS_FILE, or the default set
if [ -n "$PMC_GROUPS_FILE" ]; then mapfile -t groups < "$PMC_GROUPS_FILE"; else
groups=("MeanOccupancyPerCU" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" \
        "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "VALUBusy SALUBusy" "MemUnitStalled" \
        "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES" "GRBM_GUI_ACTIVE")
fi
for grp in "${groups[@]}"; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-boundary "$@" > $out/bench_p$i.json 2> $out/bench_p$i.err || { echo "pass $i ($grp) failed"; tail -3 $out/bench_p$i.err; }
done
python3 tools/pmc_metrics.py $out > $out/metrics.txt
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -size +6M -delete
cat $out/metrics.txt
