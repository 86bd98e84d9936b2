"""CPU: the C-ABI shared library loads and exports every symbol include/vmmt.h declares (no compute calls)."""
import os
import re


def test_library_exports_every_declared_symbol():
    from variational_mmt_amd import build as B
    from variational_mmt_amd import _lib as L
    B.build(verbose=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "vmmt.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|int64_t)\s+(vmmt_\w+)\s*\(", hdr, flags=re.M)))
    assert declared, "no declarations parsed"
    assert declared == L.EXPORTS, (set(declared) ^ set(L.EXPORTS))
    h = L.lib()
    for name in declared:
        assert hasattr(h, name), name
    assert h.vmmt_version() >= 1
    assert h.vmmt_gen_npart(30000) == 470


def test_no_cpu_fallback():
    import pytest
    import torch
    from variational_mmt_amd.engine import Dims, Engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        Engine(Dims(10, 10, 8, 8, 4), device="cpu")
