#!/usr/bin/env python3
"""What does bf16 cost the REFERENCE itself?  Runs the reference modules (imported from /root/reference with the
documented repairs of tools/make_goldens.py - nothing is copied) on the C1 fixture batch twice: in float64 (truth) and
under ``torch.autocast("cpu", dtype=torch.bfloat16)`` (every Linear / matmul in bf16, softmax / LayerNorm in fp32 - the
same rounding points a bf16-activation implementation has), and prints the per-tensor gradient rel-L2 table.

This is the floor the HIP path's per-tensor tolerance is judged against (SURVEY.md section 8c quotes its median).
Build container only (needs /root/reference).  Usage:
    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/ref_bf16_floor.py > profiles/r02_reference_bf16_floor.txt
"""
import os
import runpy
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.argv = ["make_goldens.py"]
g = runpy.run_path(os.path.join(HERE, "make_goldens.py"), run_name="not_main")     # imports the reference + the repairs
M, U = g["M"], g["U"]
g["install_repairs"]()
torch.set_num_threads(4)

fx = dict(np.load(os.path.join(HERE, "..", "tests", "golden", "transformer_c1_step.npz")))
w = {k[2:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("w/")}
x, in_len, tokens = torch.from_numpy(fx["x"]), torch.from_numpy(fx["in_len"]), torch.from_numpy(fx["tokens"])
tgt_len, gt = torch.from_numpy(fx["tgt_len"]), torch.from_numpy(fx["gt"])
cfg = g["c1_config"]()


def grads(dtype, autocast):
    m = M.Transformer(cfg)
    m.load_state_dict(w)
    m = m.to(dtype).eval()
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        logits, _ = m(x.to(dtype), in_len, tokens, tgt_len)
        loss = torch.nn.CrossEntropyLoss(ignore_index=0)(logits.float().contiguous().view(-1, cfg.vocab_size) if autocast
                                                         else logits.contiguous().view(-1, cfg.vocab_size), gt.view(-1))
    loss.backward()
    return loss.item(), {n: p.grad.double() for n, p in m.named_parameters()}


l64, g64 = grads(torch.float64, False)
l16, g16 = grads(torch.float32, True)
rows = []
for n in g64:
    if "linear_k.bias" in n:
        continue
    rows.append((((g16[n] - g64[n]).norm() / g64[n].norm()).item(), n))
rows.sort(reverse=True)
allg = torch.cat([g16[n].reshape(-1) for _, n in rows]), torch.cat([g64[n].reshape(-1) for _, n in rows])
print("# reference (repaired R1-R4) under torch.autocast(cpu, bf16) vs float64, C1 fixture (2+2 layers, d128, h4)")
print("loss %.6f vs %.6f; gradient rel-L2: global %.3e, per-tensor median %.3e, max %.3e"
      % (l16, l64, ((allg[0] - allg[1]).norm() / allg[1].norm()).item(), rows[len(rows) // 2][0], rows[0][0]))
for r in rows:
    print("  %.3e  %s" % r)
