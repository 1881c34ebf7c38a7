set -e
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1 || { tail -30 gpurun_out/r3a/pytest.log; exit 1; }
tail -3 gpurun_out/r3a/pytest.log
for v in "order_mode=2" "order_mode=0" "order_mode=0,close_supernodes=1" "order_mode=1,round_relax_pop=0"; do
  tag=$(echo $v | tr ',=' '__')
  PP_PLAN_TUNE=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary > gpurun_out/r3a/bench_$tag.json 2> gpurun_out/r3a/bench_$tag.err
  PP_PLAN_TUNE=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary --blocks 128 > gpurun_out/r3a/bench128_$tag.json 2> gpurun_out/r3a/bench128_$tag.err
  python - <<PY
import json
for f in ("gpurun_out/r3a/bench_$tag.json","gpurun_out/r3a/bench128_$tag.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("$v", f.split('/')[-1][:9], d['value'], d['ms_per_step'], d.get('correct'), {k: round(v,4) for k,v in d.get('phases_ms',{}).items()} if isinstance(d.get('phases_ms'),dict) else '')
    except Exception as e:
        print("$v", f, 'ERR', e)
PY
done
