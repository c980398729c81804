#!/usr/bin/env python3
"""Dev: how often does ProcessGroupNCCL's watchdog thread kill the process around a capture of the DP step?  (Seen once in ~15
bench runs in round 5: 'operation not permitted on an event last recorded in a capturing stream' from Watchdog::runLoop, 4 s after
bench.py's dp_probe created its one-rank group.)  K fresh TrainStep captures (2 eager steps, capture, replays) on a one-rank RCCL
group in ONE process; run the script several times and count the processes that die.  argv: K [drain]  (drain: wait until the
watchdog has retired the eager steps' collectives before capturing - trainer.TrainStep's own mitigation switch ST_DP_DRAIN_MS)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from st_amd import dp, synthetic  # noqa: E402
from st_amd.trainer import TrainStep  # noqa: E402
from transformer import Models as M, Utils as U  # noqa: E402
from transformer.Optim import ScheduledOptim  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CFG = bench.CFG
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "512")      # (the flight recorder: trainer.drain_collective_watchdog)
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
torch.manual_seed(0)
model = M.Transformer(U.AttrDict(CFG))
U.init_parameters(model)
model = model.eval().cuda()
from st_amd.arena import arena_of  # noqa: E402
arena = arena_of(model)
optim = ScheduledOptim(model, CFG["d_model"], U.AttrDict(n_warmup_steps=12000))
x, tokens, in_len, tgt_len, gt = synthetic.make_batch(8, 600, 30, CFG["feature_dim"], CFG["vocab_size"], seed=0, t_min=300, l_min=15)
xg, tg, gg = x.cuda(), tokens.cuda(), gt.cuda()
t0 = time.time()
for k in range(K):
    red = dp.GradReducer(arena, bucket_bytes=8 << 20, force=True)
    step = TrainStep(model, optim, CFG["vocab_size"], max_grad_norm=5.0, reducer=red, use_graph=True)
    for _ in range(5):
        loss, gn = step(xg, in_len, tg, tgt_len, gg)
    torch.cuda.synchronize()
    red.detach()
    del step, red
print("ok: %d captures in %.1f s, last loss %.4f, mode in-graph" % (K, time.time() - t0, float(loss)), flush=True)
dist.destroy_process_group()
