#!/bin/bash
# Instruction mix / wave-cycle breakdown of the GEMM and attention kernels (SQ counters, two PMC passes of the
# per-kernel micro-benchmark; no other trace domains).  Run through gpurun; summary -> gpurun_out/pmc_issue_mix.txt
export TMPDIR=/tmp
cd /root/repo
# ST_PMC_CMD: the profiled command (default: the per-kernel micro-benchmark at config 2's shapes); ST_PMC_OUT: summary name
CMD=${ST_PMC_CMD:-python tools/bench_kernels.py gemm attn chain}
OUT=${ST_PMC_OUT:-pmc_issue_mix}
rm -rf /tmp/pmc2 /tmp/pmc3
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc2 -o p -- $CMD > gpurun_out/pmc2.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA --kernel-trace -d /tmp/pmc3 -o p -- $CMD > gpurun_out/pmc3.log 2>&1
python - <<'PY' > gpurun_out/$OUT.txt
import sqlite3
rows = {}
for db in ("/tmp/pmc2/p_results.db", "/tmp/pmc3/p_results.db"):
    c = sqlite3.connect(db)
    q = ("select kernel_name, grid_size, counter_name, avg(value), avg(duration) from counters_collection "
         "where kernel_name like '%gemm_sym_kernel%' or kernel_name like '%attn_%kernel%' or kernel_name like '%gemm_ln%' or kernel_name like '%wgrad_group%' or kernel_name like '%wgrad_wide%' or kernel_name like '%row_chain%' "
         "group by kernel_name, grid_size, counter_name")
    for name, grid, cn, val, dur in c.execute(q):
        rows.setdefault((name, grid), {})[cn] = val
        rows[(name, grid)]["duration_ns"] = dur
print("# SQ counters per launch (largest grid of each kernel = the encoder-sized problem of config 2).  Units as in MI355X_MICROARCH.md:")
print("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles (32 per 32x32x16 MFMA).")
best = {}
for (name, grid), v in rows.items():
    if name not in best or grid > best[name][0]:
        best[name] = (grid, v)
for name, (grid, v) in sorted(best.items()):
    if "SQ_WAVE_CYCLES" not in v or "SQ_ACTIVE_INST_ANY" not in v:
        continue
    wc = v["SQ_WAVE_CYCLES"]
    nm = name.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print("%s  grid=%d  %.1f us" % (nm, grid, v["duration_ns"] / 1e3))
    print("   instr per MFMA: valu %.1f salu %.1f lds %.1f vmem %.2f   (mfma %d)" % (
        (v["SQ_INSTS_VALU"] - v["SQ_INSTS_MFMA"]) / v["SQ_INSTS_MFMA"], v["SQ_INSTS_SALU"] / v["SQ_INSTS_MFMA"],
        v["SQ_INSTS_LDS"] / v["SQ_INSTS_MFMA"], v["SQ_INSTS_VMEM"] / v["SQ_INSTS_MFMA"], v["SQ_INSTS_MFMA"]))
    print("   wave cycles: issuing %.0f%%  waiting(s_waitcnt/barrier) %.0f%%  issue-stalled %.0f%%;  MFMA pipe busy %.0f%% of SIMD time" % (
        100 * v["SQ_ACTIVE_INST_ANY"] / wc, 100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc,
        100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * v["duration_ns"] * 2.4)))
PY
cat gpurun_out/$OUT.txt | head -60
