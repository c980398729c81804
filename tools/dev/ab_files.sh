# Dev: same-box A/B of a SET of source-file variants on the whole step.  usage: ab_files.sh <rounds> <csrc name>=<variant path> ...
# A = the tree as it is, B = every named file replaced by its variant; alternating, bench.py ms/step each time.
export TMPDIR=/tmp; cd /root/repo
ROUNDS=$1; shift
C=speech-tranformer-pytorch_amd/csrc
for kv in "$@"; do f=${kv%%=*}; cp $C/$f /tmp/_orig_$f; done
run() { python -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; return; }
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-mode --no-decode --no-dp-probe $BENCH_FLAGS > /tmp/ab_bench.json 2> /tmp/ab_bench.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/ab_bench.json').read().strip().splitlines()[-1]); k = d.get('kernels', {})
    print("%s  ms/step %.4f  median-synced %.4f loss %.4f | %s" % (sys.argv[1], d['ms_per_step'], d.get('ms_per_step_median_synced', 0), d.get('loss', 0),
          " ".join("%s %.3f" % (n, v['ms_per_step']) for n, v in list(k.items())[:8])))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open('/tmp/ab_bench.err').read()[-600:])
PY
}
for r in $(seq 1 $ROUNDS); do
  for kv in "$@"; do f=${kv%%=*}; cp /tmp/_orig_$f $C/$f; done; run A
  for kv in "$@"; do f=${kv%%=*}; cp ${kv#*=} $C/$f; done; run B
done
for kv in "$@"; do f=${kv%%=*}; cp /tmp/_orig_$f $C/$f; done
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
