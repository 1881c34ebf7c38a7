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
            "parapint_amd._native, parapint_amd.examples.performance.schur_complement.synthetic_kkt; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % ROOT)
    subprocess.check_call([sys.executable, '-c', code])
