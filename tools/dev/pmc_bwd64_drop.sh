#!/bin/bash
# Dev: the same SQ counters for the dropout variant of the streams (the launches of tools/dev/attn_bwd64_drop_check.py: general kernels and streams, same Drop).
export TMPDIR=/tmp
cd /root/repo
rm -rf /tmp/pa /tmp/pb /tmp/pc
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pa -o p -- python tools/dev/attn_bwd64_drop_check.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA --kernel-trace -d /tmp/pb -o p -- python tools/dev/attn_bwd64_drop_check.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --kernel-trace -d /tmp/pc -o p -- python tools/dev/attn_bwd64_drop_check.py > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
rows = {}
for d in ("pa", "pb", "pc"):
    for db in glob.glob("/tmp/%s/**/*.db" % d, recursive=True):
        c = sqlite3.connect(db)
        try:
            q = ("select kernel_name, grid_size, counter_name, avg(value), avg(duration) from counters_collection "
                 "where kernel_name like '%attn_bwd%' group by kernel_name, grid_size, counter_name")
            for name, grid, cn, val, dur in c.execute(q):
                rows.setdefault((name, grid), {})[cn] = val
                rows[(name, grid)]["duration_ns"] = dur
        except Exception as e:
            print("db error", db, e)
for (name, grid), v in sorted(rows.items()):
    if "SQ_WAVE_CYCLES" not in v:
        continue
    wc = v["SQ_WAVE_CYCLES"]; mf = max(v.get("SQ_INSTS_MFMA", 1), 1)
    print("%s grid=%d %.1f us" % (name.replace("(anonymous namespace)::", "")[:60], grid, v["duration_ns"] / 1e3))
    print("   per MFMA: valu %.2f salu %.2f lds %.2f vmem %.2f (mfma %d, waves %d)" % ((v["SQ_INSTS_VALU"] - mf) / mf, v["SQ_INSTS_SALU"] / mf, v["SQ_INSTS_LDS"] / mf, v["SQ_INSTS_VMEM"] / mf, mf, v.get("SQ_WAVES", 0)))
    g = lambda k: 100 * v.get(k, 0) / wc
    print("   wave cycles: issuing %.0f%% waiting %.0f%% issue-stalled %.0f%% (of which LDS-issue %.0f%%); valu-active %.0f%% lds-active %.0f%% sca %.0f%%; MFMA busy %.0f%% of SIMD time (at 2.4 GHz)" % (
        g("SQ_ACTIVE_INST_ANY"), g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY"), g("SQ_WAIT_INST_LDS"), g("SQ_ACTIVE_INST_VALU"), g("SQ_ACTIVE_INST_LDS"), g("SQ_ACTIVE_INST_SCA"),
        100 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * v["duration_ns"] * 2.4)))
    print("   wave quad-cycles per MFMA %.1f ; LDS bank conflict cycles %.0f of %.0f active (%.0f%%) ; busy cycles %.0f ; LDS level %.0f VMEM level %.0f" % (wc / mf, v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0), 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1), v.get("SQ_BUSY_CYCLES", 0), v.get("SQ_INST_LEVEL_LDS", 0), v.get("SQ_INST_LEVEL_VMEM", 0)))
PY
