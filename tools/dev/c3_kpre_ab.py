#!/usr/bin/env python3
"""Dev: full-size parity figures (tests/test_fullsize_gpu.py::run_step_parity) with the encoder's keys pre-scaled in the projection's
epilogue (the default: st_row_chain's post_kscale on the chain path, st_gemm_kscale on the per-GEMM path) and with plain keys (the
round-4 form: PRESCALE_KEYS = False), over WEIGHT SEEDS - is a difference kernel or realisation?
usage: c3_kpre_ab.py <config 2|3> <utterances> <seeds, comma separated>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from st_amd import functional as F_  # noqa: E402
from st_amd.chains import EncoderChains  # noqa: E402
from tests import test_fullsize_gpu as T  # noqa: E402

cfgn, n = int(sys.argv[1]), int(sys.argv[2])
seeds = [int(v) for v in sys.argv[3].split(",")]
cfg = T.C3 if cfgn == 3 else T.C2
for seed in seeds:
    for pre in (False, True):
        F_.MhaFn.PRESCALE_KEYS = EncoderChains.PRESCALE_KEYS = pre
        tag = "c%d_b%d_s%d_%s" % (cfgn, n, seed, "kpre" if pre else "plain")
        try:
            T.run_step_parity(cfg, n, tag, seed=seed)
        except AssertionError as e:
            print("(assertion: %s)" % str(e).splitlines()[0][:100])
        lines = open(os.path.join(ROOT, "gpurun_out", "parity_%s.txt" % tag)).read().splitlines()
        g = lines[2].split()
        r = lines[3].split()
        print("config %d, %d utterances, weight seed %d, %s keys: global %s median %s max %s | bf16 reference %s %s %s"
              % (cfgn, n, seed, "pre-scaled" if pre else "plain     ", g[3].rstrip(","), g[6].rstrip(","), g[8].rstrip(";"), r[7].rstrip(","), r[9].rstrip(","), r[11]), flush=True)
        torch.cuda.empty_cache()
