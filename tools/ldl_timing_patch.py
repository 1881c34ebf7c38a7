"""Diagnostic: write a copy of dense.hip (round 3: the dense kernels' translation unit) whose k_ldl_regs prints the wall-clock ticks (100 MHz) it spends in its
four per-panel phases (a: tiles -> LDS, b: diagonal block, c: panel solve, d: trailing update).
usage: python tools/ldl_timing_patch.py <in dense.hip> <out dense.hip>   (build the copy in place of the original,
run a short bench, restore)"""
import sys
s = open(sys.argv[1]).read()
i = s.index("__global__ __launch_bounds__(LDL_THREADS) void k_ldl_regs(")
j = s.index("// x = S^-1 b with the blocked factor", i)
k = s[i:j]
k2 = k.replace("  for (int jt_loop = 0; jt_loop < nt; ++jt_loop) {",
               "  long long tA = 0, tB = 0, tC = 0, tD = 0, t0 = 0, t1 = 0;\n  for (int jt_loop = 0; jt_loop < nt; ++jt_loop) {\n    t0 = wall_clock64();", 1)
k2 = k2.replace("    lds_barrier();\n    // (b) diagonal block by wave 0",
                "    lds_barrier();\n    t1 = wall_clock64(); tA += t1 - t0; t0 = t1;\n    // (b) diagonal block by wave 0", 1)
k2 = k2.replace("    lds_barrier();\n    // finished diagonal block -> global",
                "    lds_barrier();\n    t1 = wall_clock64(); tB += t1 - t0; t0 = t1;\n    // finished diagonal block -> global", 1)
k2 = k2.replace("    lds_barrier();\n    // (d) trailing update",
                "    lds_barrier();\n    t1 = wall_clock64(); tC += t1 - t0; t0 = t1;\n    // (d) trailing update", 1)
k2 = k2.replace("    lds_barrier();\n  }\n  if (tid == 0) {\n    const bool ok = (sflags[0] == 0)",
                "    lds_barrier();\n    t1 = wall_clock64(); tD += t1 - t0; t0 = t1;\n  }\n  if (tid == 0) printf(\"ldl_regs ticks (100 MHz): a %lld b %lld c %lld d %lld\\n\", tA, tB, tC, tD);\n  if (tid == 0) {\n    const bool ok = (sflags[0] == 0)", 1)
assert k2.count("wall_clock64") == 5, k2.count("wall_clock64")
open(sys.argv[2], 'w').write(s.replace(k, k2))
