# Dev: same-box A/B of an environment switch on the WHOLE step: `bench.py` ms/step (graph replay) for each value, alternating,
# several rounds.   usage: ab_bench_env.sh <VAR> <rounds> <value>...      (extra bench flags: env BENCH_FLAGS)
export TMPDIR=/tmp; cd /root/repo
VAR=$1; ROUNDS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for V in "$@"; do
    env $VAR=$V python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-mode --no-decode --no-dp-probe $BENCH_FLAGS > /tmp/ab_bench.json 2> /tmp/ab_bench.err
    python - "$VAR" "$V" <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/ab_bench.json').read().strip().splitlines()[-1])
    k = d.get('kernels', {})
    print("%s=%s  ms/step %.4f  median-synced %.4f  shard4 %s  loss %.4f  | %s" % (sys.argv[1], sys.argv[2], d['ms_per_step'], d.get('ms_per_step_median_synced', 0),
          (d.get('shard4') or {}).get('ms_per_step'), d.get('loss', 0), " ".join("%s %.3f" % (n, v['ms_per_step']) for n, v in list(k.items())[:6])))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e, open('/tmp/ab_bench.err').read()[-800:])
PY
  done
done
