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
