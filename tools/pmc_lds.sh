#!/bin/bash
# LDS activity / bank-conflict counters of the GEMM and attention kernels (one PMC pass of the per-kernel micro-benchmark).
# Run through gpurun; prints a per-kernel summary.
export TMPDIR=/tmp
cd /root/repo
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d gpurun_out/pmc_lds -o p -- python tools/bench_kernels.py attn gemm > gpurun_out/pmc_lds.log 2>&1
python - <<'PY'
import sqlite3
c = sqlite3.connect("gpurun_out/pmc_lds/p_results.db")
rows = {}
for name, grid, cn, val, dur in c.execute("select kernel_name, grid_size, counter_name, avg(value), avg(duration) from counters_collection where kernel_name like '%attn_%' or kernel_name like '%gemm_sym%' or kernel_name like '%gemm_ln%' group by kernel_name, grid_size, counter_name"):
    rows.setdefault((name, grid), {})[cn] = val
    rows[(name, grid)]["dur"] = dur
best = {}
for (n, g), v in rows.items():
    if n not in best or g > best[n][0]: best[n] = (g, v)
for n, (g, v) in sorted(best.items()):
    nm = n.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print("%-60s %6.1f us  idx_active %.2fM  bank_conflict %.2fM (%.0f%% of active)  addr_conflict %.2fM  data_fifo_full %.2fM cmd_fifo_full %.2fM" % (
        nm, v["dur"]/1e3, v.get("SQ_LDS_IDX_ACTIVE",0)/1e6, v.get("SQ_LDS_BANK_CONFLICT",0)/1e6,
        100*v.get("SQ_LDS_BANK_CONFLICT",0)/max(1,v.get("SQ_LDS_IDX_ACTIVE",1)), v.get("SQ_LDS_ADDR_CONFLICT",0)/1e6,
        v.get("SQ_LDS_DATA_FIFO_FULL",0)/1e6, v.get("SQ_LDS_CMD_FIFO_FULL",0)/1e6))
PY
