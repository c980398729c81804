"""Dev: the full-size parity table (tests/test_fullsize_gpu.py:run_step_parity) over attention-sharpening factors: where does the
configuration stop being a measurement (the reference arithmetic under bf16 autocast itself far from fp64)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
from tests import test_fullsize_gpu as t
for f in [float(v) for v in (sys.argv[1:] or ["1.5", "2.0", "2.5"])]:
    try:
        rep = t.run_step_parity(t.C2, 32, "c2_b32_sharp_scan", use_graph=True, sharp=f)
        print("== factor", f, "PASS"); print("\n".join(rep.split("\n")[1:4]))
    except AssertionError as e:
        print("== factor", f, "FAIL"); print("\n".join(str(e).split("\n")[1:4]))
