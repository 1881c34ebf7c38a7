"""The C-ABI shared library loads on a GPU-less machine and exports every symbol that
include/parapint_hip.h declares (no compute is attempted)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'parapint_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pp_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound():
    import __graft_entry__ as entry
    entry.build()
    from parapint_amd import _native
    lib = _native.load_library()
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), name
        assert name in _native.SIGNATURES, name
    assert sorted(_native.SIGNATURES) == names


def test_a_library_older_than_its_sources_is_refused(monkeypatch):
    """The loader compares the library's build stamp with the sources beside it: an edited kernel without a rebuild
    (the library on the GPU box is the one built here) raises instead of running."""
    import pytest
    import __graft_entry__ as entry
    entry.build()
    from parapint_amd import _native
    lib = _native.load_library()
    assert lib.pp_source_sha1().decode() == _native.kernel_source_sha1()
    monkeypatch.setattr(_native, '_lib', None)
    monkeypatch.setattr(_native, 'kernel_source_sha1', lambda: 'edited')
    with pytest.raises(RuntimeError, match='rebuild'):
        _native.load_library()


def test_product_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from parapint_amd.linalg.hip_schur_complement import HipSchurComplementLinearSolver
    with pytest.raises(RuntimeError):
        HipSchurComplementLinearSolver({}, None)


def test_product_package_does_not_import_the_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import parapint_amd.linalg.hip_schur_complement, "
            "parapint_amd._native, parapint_amd.examples.performance.schur_complement.synthetic_kkt, parapint_amd.linalg.hip_engine, "
            "parapint_amd.linalg.device_ip_ops, parapint_amd.interfaces.schur_complement.device_sc_ip_interface, "
            "parapint_amd.algorithms.device_interior_point, parapint_amd.examples.stochastic_qp; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % ROOT)
    subprocess.check_call([sys.executable, '-c', code])


def test_host_staging_helper_matches_numpy():
    """pp_stage_values (host only: runs without a GPU): blocks in the reference entry order are copied to their
    staging rows, any other block is left to the caller."""
    import ctypes
    import numpy as np
    from parapint_amd import _native
    lib = _native.load_library()
    rng = np.random.default_rng(3)
    nk, nb, nblocks = 1000, 17, 9
    ref = [rng.integers(0, 50, size=nk).astype(np.int32), rng.integers(0, 50, size=nk).astype(np.int32),
           rng.integers(0, 5, size=nb).astype(np.int32), rng.integers(0, 50, size=nb).astype(np.int32)]
    blocks = []
    for i in range(nblocks):
        arrs = [ref[0].copy(), ref[1].copy(), rng.normal(size=nk), ref[2].copy(), ref[3].copy(), rng.normal(size=nb)]
        if i == 2:
            arrs[0][7] += 1                      # different entry order
        if i == 5:
            arrs[4] = arrs[4][::-1].copy()
        if i == 7:
            arrs[2] = rng.normal(size=nk - 1)    # different length
            arrs[0], arrs[1] = arrs[0][:-1].copy(), arrs[1][:-1].copy()
        blocks.append(arrs)
    ptr = np.array([[a.__array_interface__['data'][0] for a in blk] for blk in blocks], dtype=np.uint64).T.copy()
    knnz = np.array([blk[2].size for blk in blocks], dtype=np.int64)
    bnnz = np.array([blk[5].size for blk in blocks], dtype=np.int64)
    slots = np.arange(nblocks, dtype=np.int32)[::-1].copy()
    staging = np.full((nblocks, nk + nb + 3), -7.0)
    same = np.zeros(nblocks, dtype=np.uint8)
    for threads in (1, 4):
        staging[:] = -7.0
        rc = lib.pp_stage_values(nblocks, threads, ptr[0].ctypes.data, ptr[1].ctypes.data, ptr[2].ctypes.data,
                                 knnz.ctypes.data, ptr[3].ctypes.data, ptr[4].ctypes.data, ptr[5].ctypes.data,
                                 bnnz.ctypes.data, ref[0].ctypes.data, ref[1].ctypes.data, nk, ref[2].ctypes.data,
                                 ref[3].ctypes.data, nb, staging.ctypes.data, staging.shape[1], slots.ctypes.data,
                                 same.ctypes.data)
        assert rc == 0
        assert same.tolist() == [1, 1, 0, 1, 1, 0, 1, 0, 1]
        for i, blk in enumerate(blocks):
            row = staging[slots[i]]
            if same[i]:
                assert np.array_equal(row[:nk], blk[2]) and np.array_equal(row[nk:nk + nb], blk[5])
                assert np.all(row[nk + nb:] == -7.0)
            else:
                assert np.all(row == -7.0)


def test_environment_switches_are_listed_in_one_table():
    """Round 5 (code health): the library reads its environment switches through pp::env_switch only, whose table
    (csrc/switches.hpp) names and describes each one; no other getenv in the sources."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, 'parapint_amd', 'csrc')
    table = open(os.path.join(csrc, 'switches.hpp')).read()
    listed = set(re.findall(r'\{"(PP_[A-Z0-9_]+)",', table))
    assert len(listed) >= 20
    asked = set()
    for f in glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.hpp')) + glob.glob(os.path.join(csrc, '*.cpp')):
        text = open(f).read()
        if not f.endswith('switches.hpp'):
            assert 'getenv' not in text, f
        asked |= set(re.findall(r'env_switch\("(PP_[A-Z0-9_]+)"\)', text))
    assert asked <= listed, asked - listed
