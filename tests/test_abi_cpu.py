"""-m "not gpu": the C-ABI library builds, loads, and exports exactly the entry
points include/st_hip.h declares (no compute calls - there is no GPU here)."""
import os
import re

from st_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "st_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\bint\s+(st_[a-z0-9_]+)\s*\(", text))


def test_library_exports_header_symbols():
    lib = native.load()
    declared = _header_functions()
    assert declared == set(native.SIGNATURES), (declared ^ set(native.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.st_version() >= 1


def test_header_argument_counts_match_binding():
    text = open(os.path.join(ROOT, "include", "st_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, args in re.findall(r"\bint\s+(st_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        n = 0 if args.strip() == "void" else args.count(",") + 1
        assert n == len(native.SIGNATURES[name]), (name, n, len(native.SIGNATURES[name]))


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently emulating."""
    import pytest
    import torch
    import transformer.SubLayers as S
    ff = S.PositionwiseFeedForward(128, 256, dropout=0.0).eval()
    with pytest.raises(RuntimeError):
        ff(torch.randn(2, 3, 128))
    with pytest.raises(RuntimeError):
        native.gemm(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16),
                    torch.zeros(8, 8, dtype=torch.bfloat16))
