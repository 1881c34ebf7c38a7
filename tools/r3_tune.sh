#!/bin/bash
# Run ON THE GPU BOX: bench.py under a list of "variant:PP_PLAN_TUNE" settings (arguments; variant "base" = product
# library, else PP_LIB_VARIANT), at C3 and for a 128-block share
mkdir -p gpurun_out/r3_tune
for a in "$@"; do
  var=${a%%:*}; v=${a#*:}
  if [ "$var" = base ]; then unset PP_LIB_VARIANT; else export PP_LIB_VARIANT=$var; fi
  tag=$(echo $a | tr ',=.:' '____')
  for blocks in 0 128; do
    PP_PLAN_TUNE=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-boundary --no-ip-loop --no-shares --blocks $blocks --steps 30 > gpurun_out/r3_tune/b${blocks}_$tag.json 2> gpurun_out/r3_tune/b${blocks}_$tag.err || { echo "$a FAILED"; tail -3 gpurun_out/r3_tune/b${blocks}_$tag.err; continue; }
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/r3_tune/b${blocks}_$tag.json").read().strip().splitlines()[-1])
ph=d['phases']
print("%-60s blocks %4s  %8.1f it/s %.4f ms  factor %.3f fwd %.3f bwd %.3f levels %d correct %s" % ("$a", "$blocks", d['value'], d['ms_per_step'], ph['factor_levels']['ms_per_step'], ph['fwd_levels']['ms_per_step'], ph['bwd_levels']['ms_per_step'], d['plan']['n_levels'], d.get('correct')))
PY
  done
done
