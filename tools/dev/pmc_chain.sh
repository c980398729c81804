#!/bin/bash
# Dev: SQ / TCC / LDS counters of the row-chain launches of tools/bench_kernels.py chain (four PMC passes, no other trace domain).
# usage (through gpurun): bash tools/dev/pmc_chain.sh <out name>   [env ST_HIP_LIB / ST_CHAIN_PIPE as wanted]
export TMPDIR=/tmp
cd /root/repo
OUT=${1:-pmc_chain}
CMD="python tools/bench_kernels.py chain"
rm -rf /tmp/pc1 /tmp/pc2 /tmp/pc3 /tmp/pc4
N_IT=10 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pc1 -o p -- $CMD > /tmp/pc1.log 2>&1
N_IT=10 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d /tmp/pc2 -o p -- $CMD > /tmp/pc2.log 2>&1
N_IT=10 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d /tmp/pc3 -o p -- $CMD > /tmp/pc3.log 2>&1
N_IT=10 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pc4 -o p -- $CMD > /tmp/pc4.log 2>&1
tail -3 /tmp/pc3.log
python - <<'PY' > gpurun_out/$OUT.txt
import sqlite3, os
rows = {}
for db in ("/tmp/pc1/p_results.db", "/tmp/pc2/p_results.db", "/tmp/pc3/p_results.db", "/tmp/pc4/p_results.db"):
    if not os.path.exists(db):
        print("# missing", db); continue
    c = sqlite3.connect(db)
    q = ("select kernel_name, grid_size, counter_name, avg(value), avg(duration) from counters_collection "
         "where kernel_name like '%row_chain%' group by kernel_name, grid_size, counter_name")
    for name, grid, cn, val, dur in c.execute(q):
        rows.setdefault((name, grid), {})[cn] = val
        rows[(name, grid)].setdefault("dur", dur)
for (name, grid), v in sorted(rows.items(), key=lambda kv: -kv[0][1]):
    nm = name.replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    g = lambda k: v.get(k, float("nan"))
    wc = g("SQ_WAVE_CYCLES")
    print("%s grid=%d %.1f us" % (nm, grid, g("dur") / 1e3))
    print("   per MFMA: valu %.2f salu %.2f lds %.2f vmem %.2f (mfma %d)" % ((g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA"), g("SQ_INSTS_SALU") / g("SQ_INSTS_MFMA"), g("SQ_INSTS_LDS") / g("SQ_INSTS_MFMA"), g("SQ_INSTS_VMEM") / g("SQ_INSTS_MFMA"), g("SQ_INSTS_MFMA")))
    print("   wave cycles: issuing %.0f%% waiting %.0f%% issue-stalled %.0f%% (lds-issue %.0f%%); MFMA busy %.0f%% of SIMD time at 2.4 GHz" % (
        100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_WAIT_INST_LDS") / wc,
        100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g("dur") * 2.4)))
    print("   L2: hit %.2fM miss %.2fM (hit rate %.1f%%) req %.2fM ea_rd %.2fM;  LDS: active %.2fM conflict %.2fM (%.0f%%) data_fifo_full %.2fM cmd_fifo_full %.2fM" % (
        g("TCC_HIT_sum") / 1e6, g("TCC_MISS_sum") / 1e6, 100 * g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum")), g("TCC_REQ_sum") / 1e6, g("TCC_EA0_RDREQ_sum") / 1e6,
        g("SQ_LDS_IDX_ACTIVE") / 1e6, g("SQ_LDS_BANK_CONFLICT") / 1e6, 100 * g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE")), g("SQ_LDS_DATA_FIFO_FULL") / 1e6, g("SQ_LDS_CMD_FIFO_FULL") / 1e6))
PY
cat gpurun_out/$OUT.txt
