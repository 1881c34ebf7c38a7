// Every environment switch the library reads, in ONE place.  None of them is part of the product's interface: the product
// runs with none set.  They exist for (a) the A/B measurements DESIGN.md quotes -- each selects the predecessor of a kernel
// path, so that "before / after" is one build and one box -- and (b) the tests that keep those predecessors and the
// fallback paths correct (tests/test_hip_solver.py: test_measurement_switches_select_paths_that_agree, the Bunch-Kaufman
// paths of the cyclic reduction).  A name that is not in this table is never read (env_switch returns null for it);
// tests/test_cabi.py checks that the sources ask for no other name.  Python-side switches (read by the host classes, not by
// this library): PP_DIRECT_RCCL (0: keep the torch.distributed collectives; 1: RCCL calls of the library for a one-rank
// group as well), PP_BENCH_REHEARSAL (bench.py: several ranks share one GPU over gloo), PP_ROCTX (timer labels as roctx
// ranges), PP_LIB_VARIANT (load csrc/libparapint_hip_<variant>.so: kernel experiments built beside the product).
#pragma once
#include <cstdlib>
#include <cstring>

namespace pp {

struct EnvSwitch { const char* name; const char* what; };

static constexpr EnvSwitch kEnvSwitches[] = {
    {"PP_PLAN_TUNE", "developer knob: 'key=value,...' over PlanOptions (plan.hpp: apply_plan_tune), library and test interpreter alike"},
    {"PP_ORDER_CANDIDATES", "mapped groups: elimination-order candidates planned besides the configured one (0, 1, 2; plan.hpp: tune_for_mapped_group)"},
    {"PP_DEBUG_ROUNDS", "symbolic.cpp: print the rounds of independent clusters (diagnostic)"},
    {"PP_NO_WIDE_SCHUR_TILES", "n_c >= 512: 16 x 16 tiles instead of 32 x 32 super-tiles in the MFMA Schur update"},
    {"PP_NO_SCHUR_MFMA", "unmapped groups: the register-tile Schur update (k_schur_tiles) instead of the matrix-core form"},
    {"PP_SCHUR_SIDE", "Schur update on the dense stream as well (measured and not adopted, DESIGN.md)"},
    {"PP_NO_STAGE_COMPARE", "host boundary: send every staged row whole (no compare with what the device holds)"},
    {"PP_NO_FUSED_SOURCES", "f2: assemble the sources into the transposed input first (k_assemble_sources) instead of reading them through the entry records"},
    {"PP_NO_LANE_PAIRS", "one instance per lane in the gather / bottom solve kernels"},
    {"PP_NO_DENSE_DPP", "k_ldl_regs: v_readlane broadcasts instead of DP-ALU DPP"},
    {"PP_NO_DENSE_OVERLAP", "dense phase of S on the handle's stream (no side stream)"},
    {"PP_DENSE_PANEL32", "n_c > 512: the 32-column panel scheme of round 3 instead of the fat panels"},
    {"PP_NO_GROUP_STREAMS", "pattern groups one after the other on the handle's stream"},
    {"PP_NO_ENQUEUE_THREADS", "one host thread enqueues all group streams"},
    {"PP_NO_EARLY_FORWARD", "time-staged problems: the announced forward sweep behind the S phase instead of behind each group's factorisation"},
    {"PP_TRANSPOSE_TILES", "tiles per workgroup of the transposition kernels (integer)"},
    {"PP_BCR_NO_LDS", "cyclic reduction: Bunch-Kaufman blocks in global memory"},
    {"PP_NO_BCR_MFMA", "cyclic reduction: scalar block products and thread-per-column inverses"},
    {"PP_NO_BCR_LDL", "cyclic reduction: every diagonal block through Bunch-Kaufman (the path of a rejected block; used by the tests)"},
    {"PP_BCR_THREADS", "cyclic reduction: threads of the Bunch-Kaufman workgroup (integer)"},
    {"PP_BCR_FWD_PHASES", "cyclic reduction solve: the three-launch forward part of round 3"},
    {"PP_PINNED_LIMIT_MB", "pp_host_alloc: page-locked host memory one process may hold through the library (MiB, default 16384); beyond it NULL -> the caller's pageable fallback"},
    {"PP_RES_ROWS", "a-posteriori check (refine.hip): residual rows per wave (integer; default 4 up to 8 chunks of 64 instances, else 1; eight waves per workgroup)"},
    {"PP_BCR_LBOUND", "cyclic reduction: largest multiplier the unpivoted block factorisation accepts (default 100 = 1 / u; used by the tests to mix both paths in one level)"},
};

// getenv for the names of the table above; null for any other name
inline const char* env_switch(const char* name) {
  for (const EnvSwitch& s : kEnvSwitches)
    if (std::strcmp(s.name, name) == 0) return std::getenv(name);
  return nullptr;
}

}  // namespace pp
