#!/usr/bin/env python3
"""Dev tool: do two independent kernel chains captured on two streams of ONE HIP graph run concurrently?
Chain = n small GEMMs (M=1206: ~20 workgroups, latency-bound).  Prints the replay time of one chain, two chains
on one stream, and two chains on two streams (fork at the start, join at the end)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from st_amd import native as nv  # noqa: E402

dev = "cuda"
N = int(os.environ.get("N", "200"))
M = int(os.environ.get("M", "1206"))


def chain(bufs):
    x, w, y = bufs
    for _ in range(N):
        nv.gemm(x, w, y)


def mk():
    return (torch.randn(M, 256, device=dev).bfloat16(), torch.randn(256, 256, device=dev).bfloat16(),
            torch.empty(M, 256, device=dev, dtype=torch.bfloat16))


a, b = mk(), mk()
side = torch.cuda.Stream()


def one():
    chain(a)


def two_serial():
    chain(a)
    chain(b)


def two_streams():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    chain(a)
    with torch.cuda.stream(side):
        chain(b)
    main.wait_stream(side)


def timed(fn, name):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    print("%-14s %8.1f us per replay  (%5.2f us per kernel)" % (name, s.elapsed_time(e) * 100, s.elapsed_time(e) * 100 / N))


timed(one, "one chain")
timed(two_serial, "two, 1 stream")
timed(two_streams, "two, 2 streams")
