cd /root/repo
one() { (cd $1 && python bench.py --no-cpu-baseline --no-train-mode --no-decode --steps 40 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2 ms_per_step %.3f median_synced %.3f' % (d['ms_per_step'], d['ms_per_step_median_synced']))"); }
for r in 1 2 3; do one _old OLD; one . NEW; done
