"""Dev tool: kernel census of ONE graph-replayed step from a rocprofv3 --kernel-trace database (segment between two
consecutive fused-Adam launches in the middle of the timed loop).  usage: step_segment.py <results.db> [adam_index]"""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
if "kernels" in tabs:
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
else:       # rocprofv3 >= 7: dispatch + symbol tables
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    rows = c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
adam = [i for i, r in enumerate(rows) if "FusedOptimizerTensorListMeta" in r[0] or "adam_clip_kernel" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) // 2
seg = rows[adam[k] + 1:adam[k + 1] + 1]
span = (seg[-1][2] - seg[0][1]) / 1e3
busy = sum(e - s for _, s, e in seg) / 1e3
print("step %d: %d kernels, wall %.1f us, sum of kernel durations %.1f us, gaps %.1f us" % (k, len(seg), span, busy, span - busy))
cnt = collections.Counter(); tm = collections.Counter()
for n, s, e in seg:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    cnt[n] += 1; tm[n] += (e - s) / 1e3
for n, t in tm.most_common(60):
    print("%4d %9.1f us  %s" % (cnt[n], t, n))
gaps = sorted(((seg[i + 1][1] - seg[i][2]) / 1e3, seg[i][0][:40], seg[i + 1][0][:40]) for i in range(len(seg) - 1))
print("largest gaps:", [(round(g, 1), a, b) for g, a, b in gaps[-6:]])
print("median gap %.2f us" % gaps[len(gaps) // 2][0])
if len(sys.argv) > 3:      # full sequence: start offset, duration, gap to the next kernel
    with open(sys.argv[3], "w") as f:
        for i, (n, s, e) in enumerate(seg):
            gap = (seg[i + 1][1] - e) / 1e3 if i + 1 < len(seg) else 0.0
            f.write("%8.1f %7.2f %6.2f  %s\n" % ((s - seg[0][1]) / 1e3, (e - s) / 1e3, gap, n.replace("(anonymous namespace)::", "").replace("void ", "")[:90]))
