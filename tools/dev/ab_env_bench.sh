# Dev: same-box A/B of the whole step under an environment switch.  usage: ab_env_bench.sh <rounds> "<VAR=value>" [bench args]
export TMPDIR=/tmp; cd /root/repo
R=$1; KV=$2; shift 2
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-decode --no-dp-probe --no-train-mode $@"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["ms_per_step_median_synced"], "shard4", d.get("shard4",{}).get("ms_per_step"))'
for r in $(seq 1 $R); do
  python bench.py $ARGS 2>/dev/null | python -c "$P" A
  env $KV python bench.py $ARGS 2>/dev/null | python -c "$P" "B($KV)"
done
