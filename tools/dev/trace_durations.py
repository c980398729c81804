import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "info_kernel_symbol" in t][0]
for r in c.execute("select s.kernel_name, d.grid_size_x/d.workgroup_size_x, count(*), avg(d.end-d.start), min(d.end-d.start) from %s d join %s s on d.kernel_id=s.id where s.kernel_name like '%%%s%%' group by s.kernel_name, d.grid_size_x order by s.kernel_name" % (kd, ks, pat)):
    print("%-60s WGs %5d n %3d avg %7.1f us min %7.1f us" % (r[0][17:77], r[1], r[2], r[3] / 1e3, r[4] / 1e3))
