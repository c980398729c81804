# Dev: what the data-parallel machinery (bucketed all-reduces captured in the step graph, per-layer weight-gradient flushes) costs on a
# one-rank RCCL group at the shard sizes of N = 8, 4, 2, 1 (4, 8, 16, 32 utterances): bench.py plain against --force-dp --probe-only.
cd /root/repo; export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
P='import sys,json; ls=[l for l in sys.stdin.read().splitlines() if l.startswith("{")]; d=json.loads(ls[-1]); print(sys.argv[1], d["ms_per_step"], d.get("dp_mode"), d.get("dp_buckets"))'
for gb in 4 8 16 32; do
  python bench.py --global-batch $gb --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-dp-probe --no-train-mode 2>/dev/null | python -c "$P" "plain gb=$gb"
  python bench.py --global-batch $gb --force-dp --probe-only --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-train-mode --no-dp-probe 2>/dev/null | python -c "$P" "dp    gb=$gb"
done
