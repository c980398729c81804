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
    assert lib.st_version() == native.ABI_VERSION


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


def test_generated_instruction_streams_are_the_generators_output(tmp_path):
    """csrc/st_attn_bwd64_*.inc are GENERATED (tools/gen_attn_bwd64.py): the committed files must be what the committed
    generator writes, byte for byte (no hand edit of either side goes unnoticed)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if not k.startswith("BWD64_")}      # the development knobs change the output
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_bwd64.py"), str(tmp_path)], check=True, env=env,
                   stdout=subprocess.DEVNULL)
    csrc = os.path.join(ROOT, "speech-tranformer-pytorch_amd", "csrc")
    made = sorted(os.listdir(tmp_path))
    assert made == sorted(n for n in os.listdir(csrc) if n.startswith("st_attn_bwd64_") and n.endswith(".inc")), made
    for name in made:
        with open(os.path.join(tmp_path, name), "rb") as a, open(os.path.join(csrc, name), "rb") as b:
            assert a.read() == b.read(), name


def test_host_side_plans_of_the_round6_scratch_buffers():
    """The pure host queries of ABI 4 (no launch, no GPU): which backward-chain launches write their column sums to a workspace and
    how many rows it has; which attention backward launches split their dQ items over workgroups and how much scratch they want."""
    lib = native.load()._cdll
    # encoder-sized HEAD + FFN + TAIL launches: 96-row workgroups above 16,384 rows, 64-row above 8,192; everything else: atomics
    assert lib.st_row_chain_bwd_colsum_rows(24060, 1, 1024, 1) == 251
    assert lib.st_row_chain_bwd_colsum_rows(9000, 1, 1024, 1) == 141
    assert lib.st_row_chain_bwd_colsum_rows(1206, 1, 1024, 1) == 0
    assert lib.st_row_chain_bwd_colsum_rows(24060, 0, 1024, 1) == 0 and lib.st_row_chain_bwd_colsum_rows(24060, 1, 0, 1) == 0
    assert lib.st_row_chain_bwd_colsum_rows(24060, 1, 1024, 0) == 0 and lib.st_row_chain_bwd_colsum_rows(0, 1, 1024, 1) == 0
    # few queries against many keys, not causal: up to four parts of whole 128-key tiles; 16 KiB of tickets + parts x 64 x d_k fp32 per item
    assert lib.st_attn_bwd_split_kib(32, 4, 64, 50, 1000, 0) == 16 + 32 * 4 * 4 * 16
    assert lib.st_attn_bwd_split_kib(4, 4, 64, 50, 300, 0) == 16 + 4 * 4 * 3 * 16            # three key tiles: three parts
    assert lib.st_attn_bwd_split_kib(32, 4, 64, 50, 1000, 1) == 0                              # causal
    assert lib.st_attn_bwd_split_kib(32, 4, 64, 1000, 1000, 0) == 0                            # self-attention shapes
    assert lib.st_attn_bwd_split_kib(32, 4, 64, 50, 200, 0) == 0                               # too few keys for the key-split kernels
    assert lib.st_attn_bwd_split_kib(2000, 4, 64, 50, 1000, 0) == 0                            # more items than tickets
