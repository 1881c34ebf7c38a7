"""One-line summary of a bench.py JSON line (value, step, gate, device-only step, phase times).
    python tools/show_bench.py FILE [FILE ...]"""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ph = {k: round(v['ms_per_step'], 3) for k, v in (d.get('phases') or {}).items()}
    print(f, round(d['value'], 2), 'it/s', round(d['ms_per_step'], 3), 'ms', 'correct', d.get('correct'), 'residual',
          d.get('residual'), 'inertia ok', d.get('inertia') == d.get('expected_inertia'), 'device_only_ms',
          (d.get('device_only') or {}).get('ms_per_step'), ph, d.get('bcr_block_paths'), 'launches', d.get('kernel_launches_per_step'))
