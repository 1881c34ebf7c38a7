"""Run ON THE GPU BOX: tools/fuzz_solver.py through the PRODUCT engine (the C ABI on the device) instead of the test
interpreter -- seeds FIRST .. FIRST + COUNT - 1, optionally --hard; one process, one line per failure, a summary with the
solver's counters (refined solves, repairs, refused solves).  usage: python tools/fuzz_gpu.py FIRST COUNT [--hard]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import fuzz_solver
from parapint_amd.linalg.hip_engine import HipEngine

hard = '--hard' in sys.argv
args = [a for a in sys.argv[1:] if a != '--hard']
first, count = int(args[0]), int(args[1])
stats, bad, t0 = {}, [], time.time()
for i, seed in enumerate(range(first, first + count)):
    r = fuzz_solver.one(seed, hard=hard, engine=HipEngine, stats=stats)
    if r is not None:
        bad.append(r)
        print(str(r)[:300], flush=True)
    if (i + 1) % 200 == 0:
        print('done', i + 1, 'bad', len(bad), 'elapsed %.0fs' % (time.time() - t0), stats, flush=True)
print('TOTAL', count, 'bad', len(bad), stats)
