#!/usr/bin/env python3
"""Dev: full-size parity figures (tests/test_fullsize_gpu.py::run_step_parity, config 2) over WEIGHT SEEDS - run once per setting of
ST_CHAIN_PIPE (the library reads it once per process): is the pipelined forward chain's 3.36e-2 -> 3.84e-2 kernel or realisation?
usage: [ST_CHAIN_PIPE=b] pipe_parity_seeds.py <utterances> <seeds, comma separated>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from tests import test_fullsize_gpu as T  # noqa: E402
n = int(sys.argv[1])
mode = os.environ.get("ST_CHAIN_PIPE", "on")
for seed in [int(v) for v in sys.argv[2].split(",")]:
    tag = "c2_b%d_s%d_pipe_%s" % (n, seed, mode)
    try:
        T.run_step_parity(T.C2, n, tag, seed=seed)
    except AssertionError as e:
        print("(assertion: %s)" % str(e).splitlines()[0][:100])
    lines = open(os.path.join(ROOT, "gpurun_out", "parity_%s.txt" % tag)).read().splitlines()
    g, r = lines[2].split(), lines[3].split()
    print("ST_CHAIN_PIPE=%-3s %d utterances, weight seed %d: global %s median %s max %s | bf16 reference %s %s %s"
          % (mode, n, seed, g[3].rstrip(","), g[6].rstrip(","), g[8].rstrip(";"), r[7].rstrip(","), r[9].rstrip(","), r[11]), flush=True)
    torch.cuda.empty_cache()
