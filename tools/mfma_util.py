#!/usr/bin/env python
"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES pass -> MFMA busy cycles per launch of the dense kernels.

Usage: python tools/mfma_util.py <pass dir> <kernel trace csv of a --kernel-trace --stats run> [out.json [command]]
MFMA utilisation of a kernel = busy cycles / (duration x shader clock x SIMDs of the chip); the counter counts
cycles in which an MFMA is executing, summed over the SIMDs (MI355X_MICROARCH.md, cycle-constants table)."""
import csv
import json
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from pmc_summary import load, short   # noqa: E402

CLOCK_GHZ, SIMDS = 2.4, 256 * 4


def main():
    busy = load(sys.argv[1], 'SQ_VALU_MFMA_BUSY_CYCLES')
    dur = {}
    for r in csv.DictReader(open(sys.argv[2])):
        s = short(r['Kernel_Name'])
        d = dur.setdefault(s, [0.0, 0])
        d[0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
        d[1] += 1
    out = {}
    for name, (cycles, n) in busy.items():
        s = short(name)
        if cycles <= 0 or s not in dur:
            continue
        avg_s = dur[s][0] / dur[s][1]
        per_launch = cycles / n
        out[s] = {'mfma_busy_cycles_per_launch': per_launch, 'avg_duration_us': 1e6 * avg_s,
                  'mfma_util_of_chip': per_launch / (avg_s * CLOCK_GHZ * 1e9 * SIMDS),
                  'mfma_util_of_one_cu': per_launch / (avg_s * CLOCK_GHZ * 1e9 * 4)}
        print('%-28s %10.0f busy cycles/launch  %8.1f us  chip %.5f %%  one CU %.2f %%' %
              (s, per_launch, 1e6 * avg_s, 100 * out[s]['mfma_util_of_chip'], 100 * out[s]['mfma_util_of_one_cu']))
    if len(sys.argv) > 3:
        json.dump({'source': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace (own pass) of ' +
                             (sys.argv[4] if len(sys.argv) > 4 else 'bench.py at C3') +
                             '; durations from a separate --kernel-trace --stats run of the same command',
                   'clock_GHz': CLOCK_GHZ, 'kernels': out}, open(sys.argv[3], 'w'), indent=1)


if __name__ == '__main__':
    main()
