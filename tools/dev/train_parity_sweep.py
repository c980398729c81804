#!/usr/bin/env python3
"""Dev: the full-size training-mode parity figures (tests/test_fullsize_gpu.py::run_train_mode_parity: config 2, B = 32,
model.train(), every gradient against the fp64 oracle with the kernels' own dropout masks, next to the reference arithmetic
under bf16 autocast with the same masks) over several MASK SEEDS and attention-backward variants - what is kernel, what is
the realisation noise of the figure.   usage: train_parity_sweep.py [seeds, comma separated] [ST_ATTN_BWD64 values, comma separated]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from st_amd import native as nv  # noqa: E402
from tests import test_fullsize_gpu as T  # noqa: E402

seeds = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "20260928,1,2").split(",")]
modes = (sys.argv[2] if len(sys.argv) > 2 else "1,e").split(",")
print("# mask seed | ST_ATTN_BWD64 | HIP global, median, worst | bf16 reference global, median, worst | ratios global, median")
for seed in seeds:
    for mode in modes:
        os.environ["ST_ATTN_BWD64"] = mode
        nv.env_refresh()
        r = T.run_train_mode_parity(seed, out_name="parity_c2_b32_train_%d_%s.txt" % (seed, mode), check=False)
        print("%10d  %s  %.3e %.3e %.3e   %.3e %.3e %.3e   %.3f %.3f   loss %.6f (%.6f)"
              % (seed, mode, r["glob"], r["med"], r["worst"], r["floor_glob"], r["floor_med"], r["floor_worst"],
                 r["glob"] / r["floor_glob"], r["med"] / r["floor_med"], r["loss"], r["loss_oracle"]), flush=True)
        torch.cuda.empty_cache()
