import sys, heapq
sys.path.insert(0, "/root/repo/speech-tranformer-pytorch_amd")
from st_amd import synthetic
_, _, in_len, _, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
L = in_len.tolist()
print(sorted(L))
H = 4
items = []   # (tiles, b, t)
for b, l in enumerate(L):
    for t in range((l + 127) // 128):
        items.append(((l + 63) // 64, b, t))
print(len(items), "items; total tiles", sum(i[0] for i in items))
TILE = {1: 1.0, 2: 1.3, 3: 1.66}   # us per tile per WG at k resident
FIX = 2.35

def sim_cu(cols, dynamic_extra=None):
    """cols: list of lists of item tile counts (each column sequential); returns finish time of the CU"""
    rem = [ [c for c in col] for col in cols if col]
    cur = [col.pop(0) + FIX / 1.66 for col in rem]    # remaining work units of current item in 'tiles at k=3' equivalents; approx
    t = 0.0
    while cur:
        k = len(cur)
        rate = 1.0 / TILE[min(k, 3)]
        m = min(cur)
        dt = m / rate
        t += dt
        nxt, nrem = [], []
        for w, col in zip(cur, rem):
            w -= m
            if w > 1e-9:
                nxt.append(w); nrem.append(col)
            elif col:
                nxt.append(col.pop(0) + FIX / 1.66); nrem.append(col)
        cur, rem = nxt, nrem
    return t

# (a) static columns, LPT per column (what I built): 192 columns; CU bin = col % 64
def lpt_columns(C):
    cols = [[] for _ in range(C)]
    heap = [(0.0, c) for c in range(C)]
    for it in sorted(items, key=lambda x: -x[0]):
        l, c = heapq.heappop(heap)
        cols[c].append(it[0])
        heapq.heappush(heap, (l + it[0] + 1.4, c))
    return cols
cols = lpt_columns(192)
fin = [sim_cu([cols[c], cols[c + 64], cols[c + 128]]) for c in range(64)]
print("static LPT columns: span %.1f (min CU %.1f)" % (max(fin), min(fin)))

# (b) CU-aware: 64 bins x 3 slots; assign items LPT to the bin with least total; within bin to least-loaded slot
def cu_aware(nb, slots):
    bins = [[[] for _ in range(slots)] for _ in range(nb)]
    heap = [(0.0, b) for b in range(nb)]
    for it in sorted(items, key=lambda x: -x[0]):
        l, b = heapq.heappop(heap)
        s = min(range(slots), key=lambda s: sum(bins[b][s]) + 1.4 * len(bins[b][s]))
        bins[b][s].append(it[0])
        heapq.heappush(heap, (l + it[0] + 1.4, b))
    return bins
for slots in (3, 4, 2):
    bins = cu_aware(64, slots)
    fin = [sim_cu(b) for b in bins]
    print("CU-aware %d slots: span %.1f (min CU %.1f)" % (slots, max(fin), min(fin)))

# (c) hardware dynamic: sorted list, first 768 WGs breadth first (col c -> bin c % 64), rest to first-finishing
# approximate: simulate globally per bin with event-driven backfill is complex; approximate by giving extras to bins in order of earliest first completion
srt = sorted(items, key=lambda x: -x[0])
bins = [[[srt[c][0]], [srt[c + 64][0]], [srt[c + 128][0]]] for c in range(64)]
extra = srt[192:]
# the first slots to free are the shortest columns: col 191, 190, ... -> bins 63, 62, ...
for i, it in enumerate(extra):
    bins[63 - i][2].append(it[0])
fin = [sim_cu(b) for b in bins]
print("dynamic sorted dispatch (approx): span %.1f (min CU %.1f)" % (max(fin), min(fin)))
