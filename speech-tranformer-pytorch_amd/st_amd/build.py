"""Build libst_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build
container; the resulting .so travels to the GPU box with the source tree.
Staleness is decided by a content hash of csrc/ stored next to the library
(file mtimes do not survive the copy to the GPU box).
"""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIBDIR = os.path.join(os.path.dirname(HERE), "lib")
LIB = os.path.join(LIBDIR, "libst_hip.so")
STAMP = LIB + ".srchash"
SOURCES = ["st_gemm_sym.hip", "st_wgrad.hip", "st_gemm_ws.hip", "st_gemm_ln.hip", "st_gemm_lnbwd.hip", "st_rowchain.hip", "st_attn.hip",
           "st_misc.hip"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fPIC", "-shared",
         "-Wno-unused-result"]


def source_hash() -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for name in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    return h.hexdigest()


def _stale() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Compile the library if it is missing or was built from different sources."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libst_hip.so")
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        f.write(source_hash())
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
