#!/usr/bin/env python3
"""bench.py - frames/sec/step of the speech-transformer training step on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: 6 enc + 6 dec layers, d_model 256, 4 heads,
d_ff 1024, vocab 4337, 80-d fbank, B = 32 utterances PER GPU, T <= 1000 frames,
L <= 50 tokens (seeded synthetic batch of BASELINE.md section 3, resident in HBM).
One step = zero_grad + forward + CE(ignore_index=0) + backward + [RCCL gradient
average] + global-norm clip + Noam-Adam, i.e. train.py:37-46 / train_multi.py:58-68.
bf16 activations / fp32 accumulate, fp32 master weights.  The headline number is the
dropout-free step (eval-mode modules under autograd = the parity mode the oracle, the
goldens and the CPU baseline are quoted in); the same step in training mode
(model.train(): in-kernel dropout, p = 0.1 / front-end 0.5) is timed in a second pass
and reported as ``train_mode`` on the same line.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
``roofline`` (dominant kernel, live HIP-event timing) and ``cpu_baseline`` (the CPU
oracle restatement timed on this box's host cores; N = 1 only).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

C2 = dict(feature_dim=80, max_inputs_length=1000, max_target_length=50, num_enc_layer=6, num_dec_layer=6, n_heads=4,
          d_k=64, d_v=64, d_model=256, d_inner_hid=1024, dropout=0.1, vocab_size=4337)
# BASELINE config 3 (configs[2]; train_multi.py path): 12+6 layers, d_model 512, 8 heads of 64, d_ff 1024 (the reference's default;
# BASELINE.json does not fix it - SURVEY section 5), same synthetic batch.  `--config 3` times it.
C3 = dict(feature_dim=80, max_inputs_length=1000, max_target_length=50, num_enc_layer=12, num_dec_layer=6, n_heads=8,
          d_k=64, d_v=64, d_model=512, d_inner_hid=1024, dropout=0.1, vocab_size=4337)
CFG = C2          # the configuration being timed (main() rebinds it for --config 3)
BATCH, T_MAX, L_MAX, T_MIN, L_MIN = 32, 1000, 50, 500, 25
PEAK_BF16_TFLOPS = 2500.0   # dense MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
NOMINAL_CLOCK_MHZ = 2400.0   # the clock the 2.5 PFLOP/s figure is quoted at


def box_identity():
    """What lets a reader normalise one box against another: device name, compute units, the power cap (rocm-smi)."""
    info = {}
    try:
        p = torch.cuda.get_device_properties(0)
        info["device"], info["compute_units"] = p.name, int(p.multi_processor_count)
    except Exception:      # noqa: BLE001
        pass
    try:
        import subprocess
        r = subprocess.run(["rocm-smi", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=20)
        j = json.loads(r.stdout)
        for card, v in j.items():
            for k, val in v.items():
                if "power" in k.lower():
                    info["power_cap_w"] = float(val)
                    break
            break
    except Exception:      # noqa: BLE001 - informational only
        info.setdefault("power_cap_w", None)
    return info


def measure_pmc_traffic(args):
    """--pmc: the two counter passes of tools/pmc_traffic.sh as child processes of THIS run -> {class: {"bytes": ..}} or None."""
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    tmp = tempfile.mkdtemp(prefix="st_pmc_")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "2", "--no-graph", "--no-cpu-baseline",
           "--no-train-mode", "--no-decode", "--no-dp-probe", "--config", str(args.config)]
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    try:
        for counter, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", os.path.join(tmp, d), "-o", "p", "--"] + cmd,
                           check=True, capture_output=True, text=True, timeout=900, cwd=tmp, env=env)
        subprocess.run([sys.executable, os.path.join(here, "tools", "summarize_pmc.py"), os.path.join(tmp, "fetch", "p_results.db"),
                        os.path.join(tmp, "write", "p_results.db"), os.path.join(tmp, "traffic")], check=True, timeout=120, env=env)
        with open(os.path.join(tmp, "traffic.json")) as f:
            return json.load(f)
    except Exception as e:      # noqa: BLE001 - the benchmark line must still be produced
        print("bench.py --pmc: counter passes failed (%s: %s); falling back to the committed summary" % (type(e).__name__, e), file=sys.stderr)
        return None


def step_flops(in_len, tgt_len, c):
    """Algorithmic FLOPs of one step at the valid lengths."""
    d, dff, F, V = c["d_model"], c["d_inner_hid"], c["feature_dim"], c["vocab_size"]
    fwd = 0.0
    for t, l in zip(in_len.tolist(), tgt_len.tolist()):
        fwd += 2 * t * F * d + c["num_enc_layer"] * (8 * t * d * d + 4 * t * t * d + 4 * t * d * dff)
        fwd += c["num_dec_layer"] * (8 * l * d * d + 4 * l * l * d + 4 * l * d * d + 4 * t * d * d + 4 * l * t * d
                                     + 4 * l * d * dff) + 2 * l * d * V
    return 3.0 * fwd


def shard_batch(full, rank, world, global_batch):
    """The specified partition (SURVEY 8e / train_multi.py:136-139): the first `global_batch` utterances of `full` (a tuple
    of per-utterance tensors) split contiguously, rank r taking utterances [r * per, (r + 1) * per)."""
    per = global_batch // world
    return tuple(t[rank * per:(rank + 1) * per] for t in full)


def kernel_report(records):
    """Aggregate per-launch HIP-event timings into kernel classes with algorithmic work: flops from the launch's shape
    tag, HBM bytes from the launch's own operand list (native._tag(io=...): every operand once) - nothing here is derived
    from the model configuration, so a class cannot be credited with traffic its launches did not have."""
    agg = {}
    for name, tag, ms, nbytes in records:
        flops, kind = 0.0, name
        if tag is not None:
            if tag[0] == "gemm":
                _, xt, yt, M, N, K, epi = tag
                kind = {(0, 0): "gemm_fwd", (0, 1): "gemm_dgrad", (1, 1): "gemm_wgrad"}[(xt, yt)]
                flops = 2.0 * M * N * K
            elif tag[0] in ("wgrad_group", "wgrad_wide"):      # several weight gradients in one launch; tag = (name, n, flops)
                kind, flops = "gemm_wgrad", float(tag[2])
            elif tag[0] in ("gemm_ln", "gemm_lnbwd"):
                kind, flops = tag[0], 2.0 * tag[1] * tag[2] * tag[3]
            elif tag[0] in ("attn_fwd", "attn_bwd"):
                H, dk, causal, ql, kl = tag[1:6]
                pairs = float((ql.double() * kl.double()).sum().item())
                if causal:
                    pairs *= 0.5
                flops = 4.0 * pairs * dk * H          # two contractions of the non-recomputed work per kernel
                kind = "attn_fwd" if tag[0] == "attn_fwd" else {1: "attn_bwd_dq", 2: "attn_bwd_dkv", 3: "attn_bwd"}[tag[6]]
                if kind == "attn_bwd":
                    flops *= 2.0                      # dQ and dK/dV bodies in one launch
            elif tag[0] == "ln_bwd":
                kind = "ln_bwd"
            elif tag[0] in ("row_chain", "row_chain_bwd"):       # (name, rows, 256 x 256 weight blocks, d_ff)
                # encoder-sized launches (HBM-bound: 96-row workgroups) and decoder-sized ones (bound by one CU's weight
                # stream and by launch latency) are different regimes: two classes
                kind, flops = tag[0] + ("" if tag[1] > 8192 else "_dec"), 2.0 * tag[1] * tag[2] * 256 * 256
        a = agg.setdefault(kind, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "untagged": 0, "big": {}})
        if flops > 0:      # the launches of the class's largest problem (the encoder-sized ones of an attention class), kept apart
            b = a["big"].setdefault(round(flops), [0.0, 0])
            b[0] += ms
            b[1] += 1
        a["ms"] += ms
        a["launches"] += 1
        a["flops"] += flops
        if nbytes is None:
            a["untagged"] += 1
        else:
            a["bytes"] += nbytes
    return agg


def cpu_worker(args):
    """Child process: time the CPU oracle (fp32) on the first --cpu-utts utterances (default: all 32) of the benchmark
    batch, same seeded weights as the GPU model; dropout OFF (the parity mode) and dropout ON (how train.py:21 runs the
    reference: p = 0.1 in the layers, 0.5 in the front-end; masks drawn with torch.bernoulli as nn.Dropout does).
    Prints the cpu_baseline JSON object."""
    import oracle as orc
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd import synthetic
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 64))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = M.Transformer(U.AttrDict(CFG))
    U.init_parameters(model)
    p = {k: v.detach().float().clone() for k, v in model.state_dict().items()}
    x, tokens, in_len, tgt_len, gt = synthetic.make_batch(BATCH, T_MAX, L_MAX, CFG["feature_dim"], CFG["vocab_size"],
                                                          seed=0, t_min=T_MIN, l_min=L_MIN)
    n = args.cpu_utts
    b = {"x": x[:n], "in_len": in_len[:n], "tokens": tokens[:n], "tgt_len": tgt_len[:n], "gt": gt[:n]}
    # the oracle's forward / loss (autograd backward) + the stock CPU optimiser path of train.py:44-46
    leaves = {k: (v.requires_grad_(True) if not k.endswith(".pe") else v) for k, v in p.items()}
    params = [v for k, v in leaves.items() if not k.endswith(".pe")]
    opt = torch.optim.Adam(params, lr=orc.noam_lr(CFG["d_model"], 12000, 1), betas=(0.9, 0.98), eps=1e-9)
    frames = int(in_len[:n].sum())
    t_start = time.perf_counter()

    def one_step():
        t = time.perf_counter()
        opt.zero_grad()
        logits, _ = orc.transformer(leaves, b["x"], b["in_len"], b["tokens"], b["tgt_len"], CFG["n_heads"])
        loss = orc.cross_entropy(logits, b["gt"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
        return time.perf_counter() - t, loss.item()

    def bernoulli(site, shape):       # nn.Dropout: keep / (1 - p), p = 0.5 in the front-end (Models.py:31), 0.1 elsewhere
        pr = 0.5 if site == "front" else CFG["dropout"]
        return torch.empty(shape).bernoulli_(1.0 - pr).div_(1.0 - pr)

    times, loss = [], None
    for _ in range(args.cpu_steps + 1):       # first step is the warm-up
        dt, loss = one_step()
        times.append(dt)
    med = sorted(times[1:])[len(times[1:]) // 2]
    out = {"value": round(frames / med, 1), "unit": "frames/s", "cores": cores, "kind": "port",
           "sec_per_step": round(med, 3),
           "sample": "%d steps (after 1 warm-up) of the oracle restatement (fwd + CE + autograd bwd + clip + torch Adam) "
                     "on the first %d of the 32 utterances (%d valid frames), fp32, dropout off, torch %d threads; loss "
                     "%.4f" % (args.cpu_steps, n, frames, cores, loss)}
    # dropout ON, while the time budget lasts (the GPU number must not wait for it)
    if time.perf_counter() - t_start + 2.5 * med * 2 < args.cpu_timeout - 20:
        with orc.dropout_masks(bernoulli):
            td = [one_step()[0] for _ in range(2)]
        out["dropout_on"] = {"value": round(frames / min(td), 1), "unit": "frames/s", "sec_per_step": round(min(td), 3),
                             "sample": "best of 2 steps with Bernoulli masks at every nn.Dropout site of the reference "
                                       "(train.py:21 model.train())"}
    else:
        out["dropout_on"] = None
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3), help="BASELINE config: 2 = 6+6 d256 h4 (the metric's "
                    "configuration, default), 3 = 12+6 d512 h8 (the train_multi.py configuration)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-utts", type=int, default=32, help="utterances of the batch the CPU baseline is timed on")
    ap.add_argument("--cpu-timeout", type=int, default=200)
    ap.add_argument("--global-batch", type=int, default=None, help="strong scaling: this many utterances of the seed-0 batch "
                    "split contiguously over the ranks (SURVEY 8e: 32 -> 4 per GPU at N = 8).  Default: 32 - the specified "
                    "partition (BASELINE config 2 is ONE global B = 32 minibatch, train_multi.py:136-139,161-163)")
    ap.add_argument("--weak", action="store_true", help="weak scaling as the headline: 32 utterances per GPU (seed = rank); "
                    "at N > 1 the default run reports this partition as the extra `weak_scaling` block")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--wire-bf16", action="store_true", help="bf16 gradient all-reduce (train_multi.py -fp16_allreduce)")
    ap.add_argument("--dump-kernels", type=str, default=None, help="write per-shape launch timings (JSON) here")
    ap.add_argument("--no-train-mode", action="store_true", help="skip the extra (untimed-region) training-mode pass")
    ap.add_argument("--no-decode", action="store_true", help="skip the extra beam-search decode measurement")
    ap.add_argument("--force-dp", action="store_true", help="N = 1 only: run the data-parallel code path (split backward "
                    "graphs, bucketed RCCL all-reduce) on a one-rank group - its overhead without a second GPU")
    ap.add_argument("--bucket-mb", type=int, default=8, help="gradient all-reduce bucket size (MiB of fp32 gradients).  8 MiB = seven "
                    "buckets on config 2: with the collectives captured in the step graph a bucket costs no host time, and the "
                    "encoder's 19 MB must be more than one bucket for its upper layers to be exchanged under the backward of the "
                    "lower ones (TrainStep._encoder_backward(fire_layers=True)); a ring on point-to-point xGMI is bound by one "
                    "~153 GB/s link, so far smaller buckets would be latency-bound")
    ap.add_argument("--nccl-algo", type=str, default=None, help="sets NCCL_ALGO for RCCL (Ring / Tree / ...); recorded in config")
    ap.add_argument("--nccl-proto", type=str, default=None, help="sets NCCL_PROTO for RCCL (Simple / LL / LL128); recorded in config")
    ap.add_argument("--no-dp-probe", action="store_true", help="N = 1: skip the extra pass that runs the data-parallel code path "
                    "on a one-rank RCCL group to report what the exchange machinery itself costs (allreduce_exposed_ms)")
    ap.add_argument("--probe-only", action="store_true", help="internal (the dp_probe child): print the timed region's ms/step and exit")
    ap.add_argument("--pmc", action="store_true", help="N = 1: first run the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; "
                    "--kernel-trace only) of a short eager run of this workload in child processes and report the dominant "
                    "kernel's HBM traffic from THEM (roofline.traffic_source = measured) instead of the committed summary")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying "
                                                            "the captured HIP graph of the step")
    args = ap.parse_args()
    global CFG
    CFG = {2: C2, 3: C3}[args.config]
    if args.config != 2:          # the extras below are config 2's; config 3 reports its own 4-utterance shard instead
        args.no_decode = True
    if args.cpu_worker:
        return cpu_worker(args)
    # the headline partition: the global B = 32 batch split over the ranks (strong scaling; N = 1: the whole batch on one
    # GPU, the same workload either way) unless --weak
    if args.global_batch is None:
        args.global_batch = 0 if args.weak else BATCH

    import transformer.Models as M
    import transformer.Utils as U
    from transformer.Optim import ScheduledOptim
    from st_amd import dp, native, synthetic
    from st_amd.arena import arena_of
    from st_amd.trainer import TrainStep

    if args.nccl_algo:
        os.environ["NCCL_ALGO"] = args.nccl_algo
    if args.nccl_proto:
        os.environ["NCCL_PROTO"] = args.nccl_proto
    rank, local, world = dp.init_from_env()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local)
    native.load(build_if_missing=False)

    torch.manual_seed(0)
    model = M.Transformer(U.AttrDict(CFG))
    U.init_parameters(model)                      # train.py:116
    model = model.eval().cuda()                   # eval(): every Dropout is identity; autograd still runs
    arena = arena_of(model)
    dp.broadcast_parameters(arena)                # train_multi.py:176
    if args.force_dp and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "512")      # (the flight recorder: trainer.drain_collective_watchdog)
        dist.init_process_group("nccl", rank=0, world_size=1)
    reducer = dp.GradReducer(arena, bucket_bytes=args.bucket_mb << 20, wire_dtype=torch.bfloat16 if args.wire_bf16 else None,
                             force=args.force_dp) if (world > 1 or args.force_dp) else None
    optim = ScheduledOptim(model, CFG["d_model"], U.AttrDict(n_warmup_steps=12000))
    step = TrainStep(model, optim, CFG["vocab_size"], max_grad_norm=5.0, reducer=reducer, use_graph=not args.no_graph)

    def shard(global_batch):
        """global_batch = 0: weak scaling, 32 utterances per GPU (seed = rank, train_multi.py:136-139 keeps the per-rank
        batch fixed); else the seed-0 batch's first `global_batch` utterances split contiguously by rank."""
        if not global_batch:
            return synthetic.make_batch(BATCH, T_MAX, L_MAX, CFG["feature_dim"], CFG["vocab_size"], seed=rank, t_min=T_MIN,
                                        l_min=L_MIN)
        full = synthetic.make_batch(BATCH, T_MAX, L_MAX, CFG["feature_dim"], CFG["vocab_size"], seed=0, t_min=T_MIN, l_min=L_MIN)
        return shard_batch(full, rank, world, global_batch)

    if args.global_batch and (args.global_batch % world or args.global_batch > BATCH):
        raise SystemExit("--global-batch must be a multiple of the rank count and <= %d" % BATCH)
    if args.global_batch and world == 1 and args.global_batch == BATCH:
        pass                                       # N = 1: the seed-0 batch, whole (identical to the weak-scaling shard of rank 0)
    x, tokens, in_len, tgt_len, gt = shard(args.global_batch)
    xg, tg, gg = x.cuda(), tokens.cuda(), gt.cuda()          # inputs resident in HBM before timing

    def run(n):
        last = None
        for _ in range(n):
            last = step(xg, in_len, tg, tgt_len, gg)
        return last

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(max(args.warmup, 3 if not args.no_graph else 0))     # graph mode: 2 eager steps, then capture + first replay
    barrier()
    t0 = time.perf_counter()
    loss, gnorm = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    frames = torch.tensor([float(in_len.sum())], device="cuda")
    el = torch.tensor([elapsed], device="cuda")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(frames, op=dist.ReduceOp.SUM)
    elapsed, frames = el.item(), frames.item()
    ms_step = elapsed / args.steps * 1e3
    if args.probe_only:
        if rank == 0:
            print(json.dumps({"ms_per_step": round(ms_step, 4), "dp_mode": getattr(step, "dp_mode", None),
                              "dp_buckets": len(reducer.buckets) if reducer is not None else 0, "loss": round(float(loss), 4)}), flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    # ---- SURVEY 8d's protocol next to the contract's back-to-back mean: a device sync at both edges of EVERY step,
    # median of the per-step wall times; then the same with use_graph off (every kernel launched from Python: what a
    # loader whose batches never repeat a length signature gets today)
    def synced_median(n):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t = time.perf_counter()
            step(xg, in_len, tg, tgt_len, gg)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        return sorted(ts)[len(ts) // 2] * 1e3
    ms_median = synced_median(max(args.steps, 5))
    eager_ms = eager_ms_synced = None
    if not args.no_graph:
        step.use_graph = False
        run(2)
        eager_ms_synced = synced_median(max(args.steps // 2, 5))
        torch.cuda.synchronize()                      # free-running (how a training loop calls it): the host launches ahead
        t_e = time.perf_counter()
        run(max(args.steps // 2, 5))
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t_e) / max(args.steps // 2, 5) * 1e3
        step.use_graph = True
    exposed_ms = None
    if reducer is not None and reducer.active:
        # exposed gradient exchange = step time with the reducer minus the same step without any exchange (timing only)
        plain = TrainStep(model, optim, CFG["vocab_size"], max_grad_norm=5.0, reducer=None, use_graph=not args.no_graph)
        for _ in range(4):
            plain(xg, in_len, tg, tgt_len, gg)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            plain(xg, in_len, tg, tgt_len, gg)
        barrier()
        t_plain = torch.tensor([(time.perf_counter() - t1) / args.steps * 1e3], device="cuda")
        if world > 1:
            dist.all_reduce(t_plain, op=dist.ReduceOp.MAX)
        exposed_ms = round(ms_step - t_plain.item(), 3)
        dp.broadcast_parameters(arena)            # the un-exchanged steps let the replicas drift: re-align them
    dp_probe = None
    if reducer is None and world == 1 and not args.no_dp_probe and not args.no_graph:
        # N = 1: what the data-parallel machinery itself costs - the same step with a GradReducer on a ONE-rank RCCL group
        # (bucketed all-reduces captured inside the step graph); with no second GPU nothing is exchanged, so this is the
        # floor of `allreduce_exposed_ms` that the driver's N > 1 runs add their wire time to
        # ... in a CHILD process (`--force-dp`: the main path of this file with the reducer on): ProcessGroupNCCL's watchdog
        # thread can terminate a process around a capture (trainer.TrainStep._capture waits it out; round 5 saw it once in ~15
        # runs before that) and no try / except catches std::terminate - the headline must not depend on it
        import subprocess
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--force-dp", "--config", str(args.config), "--steps", str(args.steps),
                   "--warmup", str(args.warmup), "--bucket-mb", str(args.bucket_mb), "--no-cpu-baseline", "--no-train-mode", "--no-decode",
                   "--no-dp-probe", "--probe-only"] + (["--wire-bf16"] if args.wire_bf16 else [])
            # a clean stand-alone child: nothing of a torchrun agent's rendezvous (its store would be looked for at our port)
            env = {k: v for k, v in os.environ.items()
                   if not (k.startswith("TORCHELASTIC_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                   "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME"))}
            env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not line:
                raise RuntimeError("child exited %d: %s" % (r.returncode, (r.stderr or "")[-300:].replace("\n", " | ")))
            c = json.loads(line[-1])
            ms1 = float(c["ms_per_step"])
            exposed_ms = round(ms1 - ms_step, 3)
            dp_probe = {"ms_per_step": round(ms1, 3), "dp_mode": c.get("dp_mode"), "buckets": c.get("dp_buckets"),
                        "note": "one-rank RCCL group in a child process (bench.py --force-dp): the bucketed all-reduces run (captured in "
                                "the step graph) but exchange nothing"}
        except Exception as e:  # noqa: BLE001 - the headline must still be printed
            dp_probe = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- roofline pass: per-launch HIP events on the launch stream (outside the timed region) ----
    step.use_graph = False                      # per-launch events need real launches, not a graph replay
    native.timing_start()
    run(2)
    records = native.timing_stop()
    agg = kernel_report(records)
    if args.dump_kernels and rank == 0:
        shapes = {}
        for name, tag, ms, _nb in records:
            key = name + ":" + ",".join(str(t) for t in (tag or ()) if not torch.is_tensor(t))
            e = shapes.setdefault(key, [0, 0.0])
            e[0] += 1
            e[1] += ms
        with open(args.dump_kernels, "w") as f:
            json.dump({k: {"n": v[0] // 2, "avg_us": round(v[1] / v[0] * 1e3, 2), "ms_per_step": round(v[1] / 2, 4)}
                       for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])}, f, indent=1)
    total_ms = sum(a["ms"] for a in agg.values())
    dom = max(agg, key=lambda k: agg[k]["ms"])
    d = agg[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    # HBM bytes per launch of that kernel class from the PMC passes (tools/pmc_traffic.sh: separate rocprofv3
    # --pmc FETCH_SIZE / WRITE_SIZE runs of this same workload, gfx950 read correction applied); the summary
    # travels with the repo under profiles/ because counters cannot be collected inside the timed run.
    traffic, traffic_src = None, None
    measured = measure_pmc_traffic(args) if (args.pmc and world == 1) else None
    pmc = [] if measured else sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                        "*pmc_traffic_c3.json" if args.config == 3 else "*pmc_traffic.json")))
    if pmc:
        with open(pmc[-1]) as f:
            pj = json.load(f)
        traffic = pj.get(dom, {}).get("bytes")
        traffic_src = "offline: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload at git %s; not measured in this run)" % (
            os.path.basename(pmc[-1]), pj.get("git_sha", "unrecorded"))
    if measured:
        traffic = measured.get(dom, {}).get("bytes")
        traffic_src = ("measured: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of a 3-step eager run of "
                       "this workload on this box, in front of the timed region; FETCH_SIZE doubled per the gfx950 correction")
    kernels = {k: {"ms_per_step": round(v["ms"] / 2, 3), "launches": v["launches"] // 2,
                   "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 and v["flops"] else None}
               for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    # HBM side (SURVEY 8d "report both"): algorithmic bytes of every launch of a class (its own operands, each once:
    # native._tag(io=...)) / the class's summed event time -> achieved GB/s and its fraction of the 8 TB/s peak.  Classes
    # with an untagged launch report no figure rather than a partial one.
    hbm_bytes = {}
    for k, v in agg.items():
        if v["bytes"] > 0 and v["untagged"] == 0 and v["ms"] > 0:
            hbm_bytes[k] = v["bytes"] / 2          # per step (the pass ran two steps)
            gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            kernels[k]["hbm_gbs"] = round(gbs, 1)
            kernels[k]["hbm_frac"] = round(gbs / PEAK_HBM_GBS, 3)
            kernels[k]["hbm_mb_per_launch"] = round(v["bytes"] / v["launches"] / 1e6, 2)
    # the roofline that bounds the dominant class: HBM when its algorithmic intensity (flops / bytes, both per step) lies below
    # the machine balance 2500 TFLOP/s : 8 TB/s = 312 flop/byte (the row chains: ~160), else the MFMA peak (attention)
    dom_bytes = hbm_bytes.get(dom)
    mfma_frac = round(achieved / PEAK_BF16_TFLOPS, 4)
    # (attention recomputes: the backward EXECUTES 7 contractions for the 4 counted here, the forward's scores never leave
    # the chip - its operands are a few MB per launch - so these classes are matrix-pipe work whatever the quotient says)
    if dom_bytes and not dom.startswith("attn") and d["flops"] / 2 / dom_bytes < PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
        gbs = dom_bytes / (d["ms"] / 2 * 1e-3) / 1e9
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4), "mfma_tflops": round(achieved, 2), "mfma_frac": mfma_frac}
    else:
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": mfma_frac}
    # what the part sustains under chip-wide MFMA load (st_clock_probe): the same commit measures 4-8 % apart on different boxes,
    # and the nominal 2.4 GHz behind the 2.5 PFLOP/s peak is not held under matrix work - both fractions are reported
    clock_mhz = native.sustained_clock_mhz()
    peak_sustained = PEAK_BF16_TFLOPS * clock_mhz / NOMINAL_CLOCK_MHZ
    if roofline["bound"] == "mfma":
        roofline["frac_at_sustained_clock"] = round(achieved / peak_sustained, 4)
    roofline["sustained_mfma_clock_mhz"] = round(clock_mhz, 1)
    roofline["peak_at_sustained_clock"] = round(peak_sustained, 1)
    roofline.update({"traffic": None if traffic is None else round(traffic), "traffic_source": traffic_src,
                     "avg_launch_ms": round(d["ms"] / d["launches"], 4), "launches_per_step": d["launches"] // 2,
                     "share_of_kernel_time": round(d["ms"] / total_ms, 3)})
    if d.get("big"):
        # the class mixes launches of very different size (attention: six encoder-sized launches carry ~95 % of its work, twelve
        # decoder-sized ones are launch latency); `frac` above is over ALL of them - the heaviest launch on its own, for the record:
        fl = max(d["big"])
        bms, bn = d["big"][fl]
        roofline["largest_launch"] = {"launches_per_step": bn // 2, "avg_launch_ms": round(bms / bn, 4),
                                      "achieved": round(fl / (bms / bn * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                      "frac": round(fl / (bms / bn * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                      "frac_at_sustained_clock": round(fl / (bms / bn * 1e-3) / 1e12 / peak_sustained, 4)}

    # ---- training mode (model.train(): dropout 0.1 in every layer, 0.5 in the front-end, as train.py:21 runs
    # the reference) - a second, separately timed pass; the headline above stays the dropout-free parity step
    # that the CPU baseline and the oracle comparison use.  N = 1 only.
    train_mode = None
    if world == 1 and not args.no_train_mode:
        model.train()
        step_t = TrainStep(model, optim, CFG["vocab_size"], max_grad_norm=5.0, use_graph=not args.no_graph)
        for _ in range(4):
            step_t(xg, in_len, tg, tgt_len, gg)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            loss_t, _ = step_t(xg, in_len, tg, tgt_len, gg)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        step_t.use_graph = False
        native.timing_start()
        step_t(xg, in_len, tg, tgt_len, gg)
        agg_t = kernel_report(native.timing_stop())
        train_mode = {"ms_per_step": round(dt / args.steps * 1e3, 3), "value": round(frames * args.steps / dt, 1),
                      "unit": "frames/s", "loss": round(loss_t.item(), 4),
                      "kernel_ms": {k: round(v["ms"], 3) for k, v in sorted(agg_t.items(), key=lambda kv: -kv[1]["ms"])[:8]},
                      "note": "model.train(): in-kernel counter-based dropout, p = 0.1 (attention probabilities, FFN "
                              "x2 per layer) and 0.5 (front-end); same step otherwise"}
        model.eval()

    # ---- beam-search decode (BASELINE config 5: beam 10, same model and batch; N = 1 only, outside the timed
    # region): utterances/s and ms per decoder step with the KV cache.  Random weights never emit EOS, so every
    # utterance runs the full step budget - stated in the note.
    decode = None
    if world == 1 and not args.no_decode:
        from transformer.Decode import Decode
        dsteps = 50
        dec = Decode(U.AttrDict(beam_size=10, n_best=1, max_steps=dsteps), "cuda", model=model)
        dec.decode_batch((xg, in_len))                                 # warm-up at the measured shape (layouts, allocator,
        times = []                                                     # the eager first step): a server's steady state
        for _ in range(3):
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            hyps, _ = dec.decode_batch((xg, in_len))
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t2)
        dt = sorted(times)[1]                                          # median of three calls
        decode = {"utterances_per_s": round(BATCH / dt, 1), "ms_per_step": round(dt / dsteps * 1e3, 3),
                  "frames_per_s": round(float(in_len.sum()) / dt, 1),
                  "note": "transformer/Decode.py, beam 10, B = %d, %d decoder steps (random weights never emit "
                          "EOS), KV cache, device-side beams, one HIP-graph replay per step; a whole decode_batch call (encoder pass, "
                          "graph capture, read-out) after one warm-up call at this shape, median of 3"
                          % (BATCH, len(hyps[0][0]))}

    # ---- a loader whose batches never repeat a length signature (N = 1 only, outside the timed region): six different seeded
    # batches cycled through (a) TrainStep(bucket=(T_max, L_max)) - ONE captured step on padded layouts with device-resident
    # lengths - and (b) the eager packed step (every kernel launched from Python, new ragged layouts per batch)
    loader_proof = None
    if world == 1 and not args.no_graph and not args.no_decode:
        try:
            bs = []
            for sd in range(6):
                bx, bt, bil, btl, bgt = synthetic.make_batch(BATCH, T_MAX, L_MAX, CFG["feature_dim"], CFG["vocab_size"], seed=100 + sd,
                                                             t_min=T_MIN, l_min=L_MIN)
                bs.append((bx.cuda(), bil, bt.cuda(), btl, bgt.cuda()))
            res = {}
            # packed capacity: one round of the encoder-sized row chains (256 CUs x 96-row workgroups: one row more is a second
            # round); this loader's batch totals are 24,000 +- 3 %, the ones above the capacity take the padded bucket
            # (a second capacity - two rounds of 64-row workgroups, 26,880 rows - for the batches above the first)
            tgt_cap = int(BATCH * (L_MIN + L_MAX) / 2 * 1.2) // 32 * 32
            caps = [(256 * 96, tgt_cap), (420 * 64, tgt_cap)]
            for name, kw in (("bucket_graph_ms_per_step", dict(use_graph=True, graph_warmup=1, bucket=(T_MAX, L_MAX))),
                             ("packed_bucket_graph_ms_per_step", dict(use_graph=True, graph_warmup=1, bucket=(T_MAX, L_MAX),
                                                                      bucket_rows=caps)),
                             ("eager_ms_per_step", dict(use_graph=False))):
                st_l = TrainStep(model, optim, CFG["vocab_size"], max_grad_norm=5.0, **kw)
                for k in range(12):                       # two passes: every bucket has run eagerly once and is captured
                    st_l(*bs[k % 6])
                torch.cuda.synchronize()
                t_l = time.perf_counter()
                for k in range(18):
                    st_l(*bs[k % 6])
                torch.cuda.synchronize()
                res[name] = round((time.perf_counter() - t_l) / 18 * 1e3, 3)
            res["note"] = ("six batches with different lengths, cycled: one graph over B x T_max = %d padded rows, one graph over a "
                           "packed capacity of %d (batches above it: %d) frame rows / %d token rows (TrainStep(bucket_rows=...): offsets and "
                           "lengths on the device), eager launches over the ~%d packed rows"
                           % (BATCH * T_MAX, caps[0][0], caps[1][0], caps[0][1], int(in_len.sum())))
            loader_proof = res
        except Exception as e:  # noqa: BLE001 - the headline line must still be printed
            loader_proof = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- the per-GPU shard of the DP = 8 run (the first 4 utterances of the global batch) on one GPU - what every rank of the
    # specified partition executes between two all-reduces: the floor the strong-scaling curve runs into (both configs)
    shard4 = None
    if world == 1 and not args.no_graph and not args.no_decode or world == 1 and args.config == 3:
        try:
            full = synthetic.make_batch(BATCH, T_MAX, L_MAX, CFG["feature_dim"], CFG["vocab_size"], seed=0, t_min=T_MIN, l_min=L_MIN)
            xs, ts, ils, tls, gs = (t[:4] for t in full)
            xsg, tsg, gsg = xs.cuda(), ts.cuda(), gs.cuda()
            step_s = TrainStep(model, optim, CFG["vocab_size"], max_grad_norm=5.0, use_graph=not args.no_graph)
            for _ in range(4):
                step_s(xsg, ils, tsg, tls, gsg)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step_s(xsg, ils, tsg, tls, gsg)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            shard4 = {"ms_per_step": round(dt / args.steps * 1e3, 3), "value": round(float(ils.sum()) * args.steps / dt, 1), "unit": "frames/s",
                      "frames": int(ils.sum()), "note": "4 utterances = one rank's shard of the global B = 32 batch at DP = 8 (train_multi.py:136-139)"}
        except Exception as e:  # noqa: BLE001
            shard4 = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- BASELINE config 4 (N = 1, config 2's model): the joint 0.3 CTC + 0.7 attention objective of train_attn_and_ctc.py -
    # st_amd.trainer.JointTrainStep: two captured graphs around an eager torch ctc_loss (PyTorch-ROCm CTC, as the north star
    # prescribes) on the few log-probabilities it reads
    ctc_joint = None
    if world == 1 and args.config == 2 and not args.no_decode:
        try:
            from transformer.Loss import CTCAttentionLoss
            from st_amd.trainer import JointTrainStep
            torch.manual_seed(0)
            head = CTCAttentionLoss(CFG["d_model"], CFG["vocab_size"], ctc_weight=0.3).cuda()
            head._st_prepare("cuda")
            head_opt = torch.optim.Adam(head.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9, capturable=True, fused=True)
            jstep = JointTrainStep(model, optim, head, max_grad_norm=5.0, head_optimizer=head_opt, use_graph=not args.no_graph)
            for _ in range(4):
                lj = jstep(xg, in_len, tg, tgt_len, gg)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nj = max(args.steps, 5)
            for _ in range(nj):
                lj = jstep(xg, in_len, tg, tgt_len, gg)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            ctc_joint = {"ms_per_step": round(dt / nj * 1e3, 3), "value": round(float(in_len.sum()) * nj / dt, 1), "unit": "frames/s",
                         "loss": round(float(lj[0]), 4), "ctc": round(float(lj[2]), 4), "att": round(float(lj[1]), 4),
                         "note": "BASELINE config 4 (st_amd.trainer.JointTrainStep): graph A = forward + CE + the CTC head's projection over the "
                                 "ragged encoder rows + st_ctc_gather (no [T, B, V] log-softmax), eager torch ctc_loss on the [B, T, L+1] "
                                 "log-probabilities it reads, graph B = backward from both roots (st_ctc_dlogits, the head's backward "
                                 "GEMMs) + clip + Adam (model arena; torch Adam for the head's 1.1 M parameters); parity: tests/"
                                 "test_fullsize_gpu.py::test_config4_joint_trainstep_b32_graph_vs_fp64_oracle"}
        except Exception as e:  # noqa: BLE001
            ctc_joint = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- N > 1: also time the OTHER partition in the same run - headline strong (global B = 32 split contiguously, 4
    # utterances per GPU at N = 8: SURVEY 8e / north star) -> extra block weak (32 utterances per GPU, seed = rank), and
    # the other way round under --weak
    strong = weak = None
    other_gb = 0 if args.global_batch else BATCH
    if world > 1 and (other_gb == 0 or BATCH % world == 0):
        try:
            xs, ts, ils, tls, gs = shard(other_gb)
            xsg, tsg, gsg = xs.cuda(), ts.cuda(), gs.cuda()
            step_s = TrainStep(model, optim, CFG["vocab_size"], max_grad_norm=5.0, reducer=reducer, use_graph=not args.no_graph)
            for _ in range(max(args.warmup, 4)):
                step_s(xsg, ils, tsg, tls, gsg)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step_s(xsg, ils, tsg, tls, gsg)
            barrier()
            ts_ = torch.tensor([time.perf_counter() - t1], device="cuda")
            fr = torch.tensor([float(ils.sum())], device="cuda")
            dist.all_reduce(ts_, op=dist.ReduceOp.MAX)
            dist.all_reduce(fr, op=dist.ReduceOp.SUM)
            blk = {"value": round(fr.item() * args.steps / ts_.item(), 1), "unit": "frames/s",
                   "scaling": "strong" if other_gb else "weak", "ms_per_step": round(ts_.item() / args.steps * 1e3, 3),
                   "global_batch": other_gb or BATCH * world, "per_gpu_batch": (other_gb or BATCH * world) // world,
                   "note": ("the seed-0 batch of BASELINE config 2 split contiguously by rank (SURVEY 8e)" if other_gb else
                            "32 utterances per GPU, seed = rank (DistributedSampler semantics with the per-rank batch fixed)")}
            if other_gb:
                strong = blk
            else:
                weak = blk
        except Exception as e:  # noqa: BLE001 - the headline line must still be printed
            blk = {"error": "%s: %s" % (type(e).__name__, e)}
            if other_gb:
                strong = blk
            else:
                weak = blk

    out = None
    if rank == 0:
        flops = step_flops(in_len, tgt_len, CFG)
        out = {
            "metric": "frames/sec/step (80-d fbank, 6+6L d256) at 1/2/4/8 MI355X vs CPU ref" if args.config == 2 else
                      "frames/sec/step (80-d fbank, 12+6L d512 h8: BASELINE config 3) at 1/2/4/8 MI355X vs CPU ref",
            "value": round(frames * args.steps / elapsed, 1), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "ms_per_step_median_synced": round(ms_median, 3),
            "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 3),
            "eager_ms_per_step_synced": None if eager_ms_synced is None else round(eager_ms_synced, 3),
            "allreduce_exposed_ms": exposed_ms, "dp_mode": getattr(step, "dp_mode", None),
            "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE config %s, 80-d fbank, %s, "
                                   "T<=1000 (%d valid frames on rank 0), L<=50; fwd+CE+bwd+clip+Adam, dropout off"
                                   % ("2: 6+6L d256 h4 dff1024 V4337" if args.config == 2 else "3: 12+6L d512 h8 dff1024 V4337", ("global B=%d split %d per GPU (strong scaling, SURVEY 8e)" % (args.global_batch, args.global_batch // world))
                                      if args.global_batch else "B=32 per GPU (weak scaling, train_multi.py:136-139)", int(in_len.sum())),
                       "global_batch": args.global_batch or BATCH * world, "parallelism": "dp%d" % world,
                       "launch": "eager" if args.no_graph else "hipGraph replay of the whole step",
                       "wire": "bf16" if args.wire_bf16 else "fp32", "dp_bucket_mb": args.bucket_mb,
                       "nccl_algo": os.environ.get("NCCL_ALGO", "default"), "nccl_proto": os.environ.get("NCCL_PROTO", "default"),
                       "sustained_mfma_clock_mhz": roofline.get("sustained_mfma_clock_mhz"), "nominal_clock_mhz": NOMINAL_CLOCK_MHZ,
                       **box_identity()},
            "loss": round(loss.item(), 4), "grad_norm": round(gnorm.item(), 4),
            "step_tflops_valid": round(flops / (ms_step * 1e-3) / 1e12, 2),
            "step_frac_of_bf16_peak": round(flops / (ms_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "step_frac_of_bf16_peak_at_sustained_clock": round(flops / (ms_step * 1e-3) / 1e12 / roofline["peak_at_sustained_clock"], 4),
            "kernel_ms_per_step": round(total_ms / 2, 3),
            "roofline": roofline, "kernels": kernels,
            "hbm_bytes_source": "per launch: the launch's own operands, each counted once (st_amd.native._tag io lists)",
        }
        if dp_probe is not None:
            out["dp_probe"] = dp_probe
            if out["dp_mode"] is None:
                out["dp_mode"] = dp_probe.get("dp_mode")
        if strong is not None:
            out["strong_scaling"] = strong
        if weak is not None:
            out["weak_scaling"] = weak
        if train_mode is not None:
            out["train_mode"] = train_mode
        if decode is not None:
            out["decode"] = decode
        if loader_proof is not None:
            out["loader_proof"] = loader_proof
        if shard4 is not None:
            out["shard4"] = shard4
        if ctc_joint is not None:
            out["ctc_joint"] = ctc_joint

    # ---- CPU baseline: the oracle restatement on this box's host cores (rank 0, N = 1 only) -------
    # Runs in a child process with a hard timeout so that a slow / oversubscribed host can never
    # stall the benchmark: a bounded sample (the first CPU_UTTS utterances of the same batch).
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", "--config", str(args.config), "--cpu-steps",
                                str(args.cpu_steps), "--cpu-utts", str(args.cpu_utts if args.config == 2 else min(args.cpu_utts, 8)),
                                "--cpu-timeout", str(args.cpu_timeout)], capture_output=True, text=True, timeout=args.cpu_timeout)
            out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001 - the GPU number must still be reported
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": None, "kind": "port",
                                   "sample": "CPU baseline did not finish within %ds (%s)" % (args.cpu_timeout, type(e).__name__)}
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio, which a pipe would otherwise deliver at exit - AFTER this line
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)        # the last line of stdout


if __name__ == "__main__":
    main()
