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


# ---- the PRODUCT's data-parallel path: Transformer + ParamArena + GradReducer (+ TrainStep) on two ranks -------------
def _model_worker(rank, world, port, mode, out_q):
    """mode "hooks": TrainStep's eager step - gradient-ready callbacks fire bucket all-reduces from inside backward;
    mode "explicit": the sequence TrainStep replays in HIP-graph mode - decoder-side backward, ``fire_from`` the
    decoder part of the buffer, encoder backward, ``synchronize``.  Kernels are the test-only torch emulations."""
    import sys
    import numpy as np
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (here, os.path.join(here, "speech-tranformer-pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from st_amd import dp
    from st_amd.arena import arena_of
    from st_amd.trainer import TrainStep
    from tests._emul import emulated_kernels
    from tests import test_composition_cpu as comp
    import transformer.Utils as U
    from transformer.Optim import ScheduledOptim
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    dp.init_from_env(backend="gloo")
    golden = os.path.join(here, "tests", "golden")
    fx = dict(np.load(os.path.join(golden, "dp8_c1.npz")))
    _, w, _ = comp._load_c1(golden)
    if rank != 0:                                   # only rank 0 holds the real weights: broadcast must deliver them
        w = {k: (v if k.endswith(".pe") else torch.randn_like(v)) for k, v in w.items()}
    per = fx["x"].shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)        # contiguous-by-rank shards (train_multi.py:136-139)
    x, in_len, tokens = torch.from_numpy(fx["x"])[sl], torch.from_numpy(fx["in_len"])[sl], torch.from_numpy(fx["tokens"])[sl]
    tgt_len, gt = torch.from_numpy(fx["tgt_len"])[sl], torch.from_numpy(fx["gt"])[sl]
    with emulated_kernels():
        m = comp._build(w)
        arena = arena_of(m)
        dp.broadcast_parameters(arena)               # train_multi.py:176
        red = dp.GradReducer(arena, bucket_bytes=64 << 10)
        assert red.active and len(red.buckets) > 8
        opt = ScheduledOptim(m, 128, U.AttrDict(n_warmup_steps=100))
        step = TrainStep(m, opt, 30, max_grad_norm=1e9, reducer=red)        # no clipping: arena.grad stays the average
        if mode == "hooks":
            loss, gnorm = step(x, in_len, tokens, tgt_len, gt)
        else:
            red.detach()
            t_max, l_max = int(in_len.max()), int(tgt_len.max())
            loss = step._forward_decoder_backward(x[:, :t_max], in_len, tokens[:, :l_max], tgt_len, gt[:, :l_max])
            lo = step._decoder_grad_start()
            assert 0 < lo < arena.total
            red.fire_from(lo)                        # decoder-side buckets: in flight while the encoder's backward runs
            assert any(red._fired) and not all(red._fired)
            step._encoder_backward()
            red.synchronize()
            opt.update_learning_rate(1)
            gnorm = step._clip_and_update()
        mean_loss = dp.allreduce_mean(loss.detach().float())
        grads = {n: arena.grad_view(p).detach().numpy().copy() for n, p in m.named_parameters()}
        flat = arena.flat.detach().clone()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)      # every rank applied the identical update
    if rank == 0:
        out_q.put((grads, float(mean_loss), float(gnorm), same))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("hooks", 2), ("explicit", 2), ("hooks", 8)])
def test_product_dp_path_matches_oracle_shard_average(mode, world):
    """Transformer + ParamArena + GradReducer on world 2 (gloo; 4 + 4 utterances) and on world 8 (one utterance per rank -
    the rank count of BASELINE's DP = 8 configuration, contiguous shards as bench.shard_batch cuts them) == the oracle's
    ``dp_average_grads`` (pinned to fixture F8 by tests/test_oracle_golden.py::test_dp8_average) on the same 8-utterance batch."""
    import sys
    import numpy as np
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "speech-tranformer-pytorch_amd"))
    import oracle as orc
    from tests import test_composition_cpu as comp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_model_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    grads, mean_loss, gnorm, same = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same
    golden = os.path.join(here, "tests", "golden")
    fx = dict(np.load(os.path.join(golden, "dp8_c1.npz")))
    _, w, _ = comp._load_c1(golden)
    batch = {k: torch.from_numpy(fx[k]) for k in ("x", "in_len", "tokens", "tgt_len", "gt")}
    batch["x"] = batch["x"].double()
    loss64, gavg = orc.dp_average_grads({k: v.double() for k, v in w.items()}, batch, int(fx["n_head"]), world)
    assert abs(mean_loss - loss64.item()) <= 2e-2 * loss64.item()
    rels, fg, ft = [], [], []
    for n, t in gavg.items():
        if "linear_k.bias" in n:
            continue
        g = torch.from_numpy(grads[n]).double()
        rels.append(comp.rel(g, t))
        fg.append(g.reshape(-1))
        ft.append(t.reshape(-1))
    assert max(rels) < comp.GRAD_TOL_TENSOR and sorted(rels)[len(rels) // 2] < comp.GRAD_TOL_MEDIAN
    assert comp.rel(torch.cat(fg), torch.cat(ft)) < comp.GRAD_TOL_GLOBAL
    total = torch.cat(ft).norm().item()
    assert abs(gnorm - total) <= 2e-2 * total          # the norm TrainStep clips with is the norm of the AVERAGED gradient
