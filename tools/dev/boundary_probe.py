#!/usr/bin/env python3
"""Dev: what a dependent kernel boundary costs on this stack - N launches of the same ~8 us kernel (st_zero over 53 MB, write-through or
not is irrelevant: the same buffer every time) issued eagerly into one stream (the host runs ahead) and replayed from a HIP graph;
and the same with a ~1 us kernel in the graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch
from st_amd import native as nv
big = torch.empty(13_300_000, device="cuda")
small = torch.empty(4096, device="cuda")


def eager(t, n):
    for _ in range(20): nv.zero_(t)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): nv.zero_(t)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def graph(t, n):
    for _ in range(3): nv.zero_(t)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): nv.zero_(t)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * n) * 1e3


print("53 MB zero kernel: eager %.2f us per launch, graph %.2f us per launch" % (eager(big, 200), graph(big, 200)))
print("16 KB zero kernel: graph %.2f us per launch (eager is host-bound: %.2f)" % (graph(small, 400), eager(small, 200)))
