"""The bench.py output contract, checked on the committed round artefact (profiles/*_bench.json is the JSON line a
real `python bench.py` run printed on the MI355X): every field the driver reads is present and well-formed."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_follows_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench.json")))
    assert files, "no committed bench line under profiles/"
    with open(files[-1]) as f:
        d = json.loads(f.read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong")
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = frames of the timed steps / wall time, consistent with ms_per_step
    frames = int(re.search(r"\((\d+) valid frames", d["config"]["workload"]).group(1))
    assert abs(d["value"] - frames * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    if "hbm_bytes_source" in d:      # round 3 on: per-class HBM figures come from the launches' own operand lists
        for k, v in d["kernels"].items():
            assert v.get("hbm_frac", 0.0) <= 1.0, (k, v)      # > 1 means the accounting, not the kernel, did the work
            assert v.get("tflops") is None or v["tflops"] <= 2500.0, (k, v)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_kernel_report_takes_bytes_from_the_launches():
    """bench.kernel_report: a class's HBM bytes are the sum of its launches' own operand lists (native._tag io=...); a class
    with an untagged launch reports none; st_amd.native._io_bytes counts every operand once, rows clipped to the launch."""
    import importlib.util
    import sys

    import torch
    sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
    from st_amd import native
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    x = torch.zeros(10, 8, dtype=torch.bfloat16)
    assert native._io_bytes((x, (x, 4), None, 100.0, (torch.zeros(7, dtype=torch.float32), 3))) == 10 * 8 * 2 + 4 * 8 * 2 + 100 + 3 * 4
    recs = [("st_gemm", ("gemm", 0, 0, 100, 64, 32, 0), 0.5, 1000.0), ("st_gemm", ("gemm", 0, 0, 100, 64, 32, 0), 0.5, 3000.0),
            ("st_gemm_ln", ("gemm_ln", 10, 20, 30), 1.0, None), ("st_row_chain", ("row_chain", 24060, 12, 1024), 2.0, 5e6),
            ("st_row_chain", ("row_chain", 1206, 12, 1024), 1.0, 1e5), ("st_adam_clip", ("adam_clip", 5), 0.25, 160.0)]
    agg = bench.kernel_report(recs)
    assert agg["gemm_fwd"]["bytes"] == 4000.0 and agg["gemm_fwd"]["launches"] == 2 and agg["gemm_fwd"]["untagged"] == 0
    assert agg["gemm_fwd"]["flops"] == 2 * (2.0 * 100 * 64 * 32)
    assert agg["gemm_ln"]["untagged"] == 1 and agg["gemm_ln"]["bytes"] == 0.0
    assert agg["row_chain"]["bytes"] == 5e6 and agg["row_chain_dec"]["bytes"] == 1e5      # encoder- / decoder-sized classes
    assert agg["st_adam_clip"]["bytes"] == 160.0


def test_strong_scaling_shards_partition_the_global_batch():
    """bench.shard_batch: the ranks' shards are contiguous, disjoint and cover the global B = 32 batch."""
    import importlib.util

    import torch
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = (torch.arange(32 * 3).view(32, 3), torch.arange(32))
    for world in (1, 2, 4, 8):
        parts = [bench.shard_batch(full, r, world, 32) for r in range(world)]
        assert all(p[0].shape[0] == 32 // world for p in parts)
        assert torch.equal(torch.cat([p[1] for p in parts]), full[1]) and torch.equal(torch.cat([p[0] for p in parts]), full[0])
