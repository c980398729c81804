"""-m "not gpu": the data-parallel path (st_amd.dp - the Horovod role of train_multi.py) on
world_size = 2 with the gloo backend: bucketed, backward-overlapped gradient averaging over a
flat gradient buffer must equal the single-process average of the per-shard gradients
(train_multi.py:136-139,161-163 semantics: per-rank token-mean loss, rank-averaged gradients),
parameters are broadcast from rank 0 (train_multi.py:176) and metrics averaged (:31)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(16, 64)
        self.b = nn.Linear(64, 64)
        self.c = nn.Linear(64, 8)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, wire_bf16, out_q, explicit=False):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "speech-tranformer-pytorch_amd"))
    from st_amd import dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    r, _, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)              # different init per rank: broadcast must fix it
    model = Toy()
    flat = dp.FlatGrads(model)
    dp.broadcast_parameters(model, root=0)
    # tiny buckets so that several all-reduces fire during backward
    red = dp.GradReducer(flat, bucket_bytes=4096, wire_dtype=torch.bfloat16 if wire_bf16 else None)
    assert len(red.buckets) > 3
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 16, generator=g)
    y = torch.randn(8, 8, generator=g)
    sl = slice(rank * 4, rank * 4 + 4)
    if explicit:                               # HIP-graph mode: no hooks, the trainer fires ranges itself
        red.detach()
    for step in range(2):                      # second step: buckets re-arm, grads re-zeroed
        flat.zero_grad()
        loss = ((model(x[sl]) - y[sl]) ** 2).mean()
        loss.backward()
        if explicit:
            red.fire_from(flat.total // 2)     # the tail buckets first (TrainStep: after the decoder's backward graph)
            assert any(red._fired) and not all(red._fired)
        red.synchronize()
    mean_loss = dp.allreduce_mean(loss.detach())
    if rank == 0:
        # plain numpy through the queue (torch tensors travel by fd passing, which races with process exit)
        out_q.put(({n: p.detach().numpy().copy() for n, p in model.named_parameters()},
                   {n: p.grad.detach().numpy().copy() for n, p in model.named_parameters()}, float(mean_loss)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wire_bf16,explicit", [(False, False), (True, False), (False, True)])
def test_bucketed_allreduce_matches_shard_average(wire_bf16, explicit):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, wire_bf16, q, explicit)) for r in range(world)]
    for p in procs:
        p.start()
    params, grads, mean_loss = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process emulation with rank 0's (broadcast) parameters
    torch.manual_seed(100)
    ref = Toy()
    params = {n: torch.from_numpy(v) for n, v in params.items()}
    grads = {n: torch.from_numpy(v) for n, v in grads.items()}
    ref.load_state_dict(params)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 16, generator=g)
    y = torch.randn(8, 8, generator=g)
    acc = {n: torch.zeros_like(p) for n, p in ref.named_parameters()}
    losses = []
    for r in range(world):
        ref.zero_grad()
        sl = slice(r * 4, r * 4 + 4)
        loss = ((ref(x[sl]) - y[sl]) ** 2).mean()
        loss.backward()
        losses.append(loss.detach())
        for n, p in ref.named_parameters():
            acc[n] += p.grad / world
    tol = 2e-2 if wire_bf16 else 1e-6
    for n in acc:
        assert torch.allclose(grads[n], acc[n], rtol=tol, atol=tol * acc[n].abs().max().item()), n
    assert abs(mean_loss - torch.stack(losses).mean().item()) < 1e-6


def test_bucket_layout_covers_buffer_back_to_front():
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "speech-tranformer-pytorch_amd"))
    from st_amd import dp
    flat = dp.FlatGrads(Toy())
    red = dp.GradReducer(flat, bucket_bytes=1024)
    assert red.buckets[0][1] == flat.total and red.buckets[-1][0] == 0
    for (lo, hi), (lo2, hi2) in zip(red.buckets, red.buckets[1:]):
        assert hi2 == lo and lo2 < hi2
