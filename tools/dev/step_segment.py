"""Dev tool: kernel census of ONE graph-replayed step from a rocprofv3 --kernel-trace database (segment between two
consecutive fused-Adam launches in the middle of the timed loop).  usage: step_segment.py <results.db> [adam_index]"""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if "FusedOptimizerTensorListMeta" in r[0]]
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
