# Dev: A/B of a source variant on ONE box (box-to-box variance is +-5 %): kernel-trace durations of the launches of
# `bench_kernels.py $3..` whose name matches $2, with the library as shipped, then with `sed` expression $4 applied to csrc/$1
# and the library rebuilt.   usage: ab_src.sh st_attn.hip attn 'attn' 's/a/b/'
export TMPDIR=/tmp; cd /root/repo
FILE=$1; PAT=$2; SECT=$3; EXPR=$4
dur() { python - "$1" <<PY
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type=\"table\"")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "info_kernel_symbol" in t][0]
for r in c.execute("select s.kernel_name, d.grid_size_x/d.workgroup_size_x, count(*), avg(d.end-d.start), min(d.end-d.start) from %s d join %s s on d.kernel_id=s.id where s.kernel_name like '%%' || '$PAT' || '%%' group by s.kernel_name, d.grid_size_x order by s.kernel_name, d.grid_size_x" % (kd, ks)):
    print("%-60s WGs %5d n %3d avg %7.1f us min %7.1f us" % (r[0][17:77], r[1], r[2], r[3] / 1e3, r[4] / 1e3))
PY
}
rocprofv3 --kernel-trace -d /tmp/abA -o t -- python tools/bench_kernels.py $SECT > /dev/null 2>&1; echo "== A (as shipped)"; dur /tmp/abA/t_results.db
sed -i "$EXPR" speech-tranformer-pytorch_amd/csrc/$FILE
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
rocprofv3 --kernel-trace -d /tmp/abB -o t -- python tools/bench_kernels.py $SECT > /dev/null 2>&1; echo "== B ($EXPR)"; dur /tmp/abB/t_results.db
