import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/speech-tranformer-pytorch_amd")
import st_amd.native as nv
orig = nv.wgrad_group
def spy(problems, wide=False):
    problems = list(problems)
    print("FLUSH wide=%s n=%d" % (wide, len(problems)))
    from collections import Counter
    c = Counter((p[0].shape[0], p[0].shape[1], p[5], p[4], p[3] is not None) for p in problems)
    for k, v in sorted(c.items()): print("   tokens %d K_in %d N_out %d splits %d bias %s  x%d" % (*k, v))
    return orig(problems, wide=wide)
nv.wgrad_group = spy
import st_amd.functional as F_
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-train-mode", "--no-decode", "--no-graph"]
import runpy
runpy.run_path("/root/repo/bench.py", run_name="__main__")
