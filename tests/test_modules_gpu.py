"""-m gpu: the drop-in module tree on a real MI355X, real kernels, against the
fp64 oracle / the import-generated goldens.  Same bodies as the CPU composition
test (tests/test_composition_cpu.py), without the emulation.

Tolerances (bf16 activations, fp32 accumulate, vs fp64 truth):
logits rel-L2 <= 2e-2, per-tensor gradient rel-L2 <= 8e-2, the analytically
zero linear_k.bias gradients by absolute bound only."""
import pytest

from tests import test_composition_cpu as comp

pytestmark = pytest.mark.gpu


def test_c1_full_step_gpu(golden_dir):
    """BASELINE config 1: 2+2 layers, d128, h4 - loss, logits and all 90 gradients."""
    comp.run_c1_step(golden_dir, "cuda")


def test_standalone_modules_gpu(golden_dir):
    comp.run_standalone_modules(golden_dir, "cuda")


def test_encoder_padded_api_gpu(golden_dir):
    comp.run_encoder_padded_api(golden_dir, "cuda")


def test_native_library_is_loaded():
    """The GPU tests must run the HIP library, not a fallback."""
    from st_amd import native
    lib = native.load(build_if_missing=False)
    assert lib.st_version() >= 1
    with open("/proc/self/maps") as f:
        assert "libst_hip.so" in f.read()
