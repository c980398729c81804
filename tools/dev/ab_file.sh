# Dev: same-box A/B of two versions of one source file: usage ab_file.sh <csrc file> <variant B path (inside the repo)> <name pattern> <bench sections..>
export TMPDIR=/tmp; cd /root/repo
FILE=$1; B=$2; PAT=$3; shift 3
dur() { python - "$1" <<PY
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type=\"table\"")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "info_kernel_symbol" in t][0]
for r in c.execute("select s.kernel_name, d.grid_size_x/d.workgroup_size_x, count(*), avg(d.end-d.start), min(d.end-d.start) from %s d join %s s on d.kernel_id=s.id where s.kernel_name like '%%' || '$PAT' || '%%' group by s.kernel_name, d.grid_size_x order by s.kernel_name, d.grid_size_x" % (kd, ks)):
    print("%-60s WGs %5d n %3d avg %7.1f us min %7.1f us" % (r[0][17:77], r[1], r[2], r[3] / 1e3, r[4] / 1e3))
PY
}
rocprofv3 --kernel-trace -d /tmp/abA -o t -- python tools/bench_kernels.py "$@" > /dev/null 2>&1; echo "== A (as shipped)"; dur /tmp/abA/t_results.db
cp $B speech-tranformer-pytorch_amd/csrc/$FILE
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
rocprofv3 --kernel-trace -d /tmp/abB -o t -- python tools/bench_kernels.py "$@" > /dev/null 2>&1; echo "== B ($B)"; dur /tmp/abB/t_results.db
