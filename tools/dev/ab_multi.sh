# Dev: same-box A/B/C.. of several versions of one source file.
# usage: ab_multi.sh <csrc file> <name pattern> <bench section> <variant path | ->...   ('-' = as committed)
# env PYTEST_K: after each variant run `pytest tests/test_kernels_gpu.py -k $PYTEST_K`
export TMPDIR=/tmp; cd /root/repo
FILE=$1; PAT=$2; SECT=$3; shift 3
cp speech-tranformer-pytorch_amd/csrc/$FILE /tmp/_orig_$FILE
dur() { python - "$1" <<PY
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type=\"table\"")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "info_kernel_symbol" in t][0]
for r in c.execute("select s.kernel_name, d.grid_size_x/d.workgroup_size_x, count(*), avg(d.end-d.start), min(d.end-d.start) from %s d join %s s on d.kernel_id=s.id where s.kernel_name like '%%' || '$PAT' || '%%' group by s.kernel_name, d.grid_size_x order by s.kernel_name, d.grid_size_x" % (kd, ks)):
    print("%-60s WGs %5d n %3d avg %7.1f us min %7.1f us" % (r[0][17:77], r[1], r[2], r[3] / 1e3, r[4] / 1e3))
PY
}
i=0
for V in "$@"; do
  if [ "$V" = "-" ]; then cp /tmp/_orig_$FILE speech-tranformer-pytorch_amd/csrc/$FILE; else cp $V speech-tranformer-pytorch_amd/csrc/$FILE; fi
  python -c "import __graft_entry__ as g; g.build()" > /tmp/build_$i.log 2>&1 || { echo "BUILD FAILED $V"; tail -5 /tmp/build_$i.log; continue; }
  rm -rf /tmp/ab$i; rocprofv3 --kernel-trace -d /tmp/ab$i -o t -- python tools/bench_kernels.py $SECT > /tmp/bk_$i.log 2>&1
  echo "== $V"; dur /tmp/ab$i/t_results.db
  if [ -n "$PYTEST_K" ]; then python -m pytest tests/test_kernels_gpu.py -x -q -k "$PYTEST_K" 2>&1 | tail -3; fi
  i=$((i+1))
done
cp /tmp/_orig_$FILE speech-tranformer-pytorch_amd/csrc/$FILE
