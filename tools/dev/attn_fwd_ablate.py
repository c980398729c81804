"""Dev: generate the timing-ablation variants of csrc/st_attn64.hip used for profiles/r03_attn_fwd_ablation.txt into
tools/dev/_ab/ (git-ignored).  Each variant removes or replaces ONE component of the forward's loop; their results are
numerically wrong on purpose - only their duration is read.  Run them on one box with
  bash tools/dev/ab_multi.sh st_attn64.hip attn_fwd64 attn - tools/dev/_ab/<variant>.hip ..."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "speech-tranformer-pytorch_amd", "csrc", "st_attn64.hip")).read()
out = os.path.join(ROOT, "tools", "dev", "_ab")
os.makedirs(out, exist_ok=True)
# the range fall-back would re-run items whose (deliberately wrong) row sums leave the plain-exponential range
base = src.replace("if (__syncthreads_or(q < lq && !(ltot > F64_SMALL && ltot < F64_BIG)))", "if (__syncthreads_or(ltot == 12345.f))")
assert base != src
i = base.index("__device__ __forceinline__ void lean_tile_pipe("); j = base.index("template <bool DROP>\n__global__", i)
body = base[i:j]
LOOP = "      store(ks);\n      load(it + 1);\n      __syncthreads();"
assert LOOP in base


def put(name, text):
    open(os.path.join(out, name + ".hip"), "w").write(text)


def tile(nok=False, nov=False, noexp=False, fma=False, nomsum=False):
    b = body
    if nok:
        b = b.replace("rd_nat<DK>(ks, (l & 31), t)", "qf[t]").replace("rd_nat<DK>(ks, 32 + (l & 31), t)", "qf[(t + 1) & 3]")
    if nov:
        b = re.sub(r"rd_tr<DK>\(vs, (\d+), ([^)]*)\)", r"qf[(\1 / 32 + 1) & 3]", b)
    if fma:
        b = b.replace("__builtin_amdgcn_exp2f(s0[r])", "fmaf(s0[r], 1e-3f, 0.5f)").replace("__builtin_amdgcn_exp2f(s1[r])", "fmaf(s1[r], 1e-3f, 0.5f)")
    if noexp:
        b = b.replace("__builtin_amdgcn_exp2f(s0[r])", "s0[r]").replace("__builtin_amdgcn_exp2f(s1[r])", "s1[r]")
        b = b.replace("pack_acc8(s0, 0)", "qf[0]").replace("pack_acc8(s0, 8)", "qf[1]")
        b = b.replace("  pa = pack_acc8(s1, 0);\n  pb = pack_acc8(s1, 8);", '  asm volatile("" :: "v"(s0), "v"(s1));\n  pa = qf[2];\n  pb = qf[3];')
    if nomsum:
        b = re.sub(r"\n\s*lacc = mfma32\(ones, p[ab], lacc\);", "", b)
    return base[:i] + b + base[j:]


put("x0_base", base)
put("h1_noload", base.replace("      load(it + 1);\n      __syncthreads();", "      if (it == 0) load(it + 1);\n      __syncthreads();"))
put("h2_nobar", base.replace("      load(it + 1);\n      __syncthreads();", "      load(it + 1);"))
put("h4_nostore", base.replace("      store(ks);\n      load(it + 1);", "      if (it == 0) store(ks);\n      load(it + 1);"))
put("h5_occ2", base.replace("__launch_bounds__(256, 3) void attn_fwd64_kernel", "__launch_bounds__(256, 2) void attn_fwd64_kernel"))
put("h6_bare", base.replace(LOOP, "      if (it == 0) { store(ks); load(it + 1); __syncthreads(); }"))
put("h7_fmaexp", tile(fma=True))
put("h10_nomsum", tile(nomsum=True))
put("x1_notr", tile(nov=True))
put("x2_nok", tile(nok=True))
put("x3_nolds", tile(nok=True, nov=True))
put("x4_mfmaonly", tile(nok=True, nov=True, noexp=True))
put("y1_mfma_nomem", tile(nok=True, nov=True, noexp=True).replace(LOOP, "      if (it == 0) { store(ks); load(it + 1); __syncthreads(); }"))
one = base.replace("const int ntiles = (lk + TILE - 1) / TILE;", "const int ntiles = 1;")
put("y2_onetile", one)
put("y3_noloop", one.replace("      lean_tile<DROP, true, EXACT>(ks, ks + G::E, qf, o, m, lsum, lacc, ones, it * TILE, lk, q, dr, bh);",
                            "      if (lk < 0) lean_tile<DROP, true, EXACT>(ks, ks + G::E, qf, o, m, lsum, lacc, ones, it * TILE, lk, q, dr, bh);"))
put("y4_empty", base.replace("  if (q0 >= lq) return;\n  const int wave", "  if (q0 >= lq || lq > 0) return;\n  const int wave"))
print("variants in", out)
