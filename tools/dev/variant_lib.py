"""Dev: a library variant for same-box A/Bs - one or more csrc files (a.hip,b.hip) recompiled with extra -D switches, linked with the objects of the
current build: python tools/dev/variant_lib.py <tag> <csrc file> [-DX=1 ...]  ->  tools/dev/_ab/libst_<tag>.so (run with
ST_HIP_LIB=tools/dev/_ab/libst_<tag>.so; native.load() takes that path as it is)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
from st_amd import build
tag, srcs, defs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]      # (several files: a.hip,b.hip)
build.build_lib()
stems = [os.path.splitext(s)[0] for s in srcs]
objs = [os.path.join(build.OBJDIR, o) for o in sorted(os.listdir(build.OBJDIR)) if o.endswith(".o") and o.split(".")[0] not in stems]
assert len(objs) == len(build.SOURCES) - len(srcs), objs
out_dir = os.path.join(ROOT, "tools", "dev", "_ab")
os.makedirs(out_dir, exist_ok=True)
from concurrent.futures import ThreadPoolExecutor
def one(src):
    obj = "/tmp/%s.%s.o" % (os.path.splitext(src)[0], tag)
    subprocess.run(["hipcc"] + build.FLAGS + defs + ["-c", os.path.join(build.CSRC, src), "-o", obj], check=True, cwd=build.CSRC)
    return obj
with ThreadPoolExecutor(8) as ex:
    new = list(ex.map(one, srcs))
lib = os.path.join(out_dir, "libst_%s.so" % tag)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + new + ["-o", lib], check=True)
print(lib)
