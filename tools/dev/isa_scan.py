"""Dev: compile every csrc/*.hip to gfx950 assembly and list, per kernel, scratch (spill) instructions, v_accvgpr moves
(AGPRs used as spill space) and v_mov_b64 - the three signatures of register trouble that cost the attention forward a
quarter of its VALU instructions in round 2 (a run-time branch whose two sides kept the accumulators in different registers).
usage: python tools/dev/isa_scan.py [file.hip ...]"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "speech-tranformer-pytorch_amd", "csrc")
files = [os.path.abspath(f) for f in sys.argv[1:]] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
for f in files:
    asm = "/tmp/isa_scan_%s.s" % os.path.basename(f)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-S", "--cuda-device-only", f, "-o", asm],
                   check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cur, cnt = None, {}
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); cnt[cur] = [0, 0, 0, 0]
            continue
        t = line.split()
        if cur is None or not t:
            continue
        if t[0].startswith("scratch_"): cnt[cur][0] += 1
        elif t[0].startswith("v_accvgpr"): cnt[cur][1] += 1
        elif t[0] == "v_mov_b64_e32": cnt[cur][2] += 1
        cnt[cur][3] += 1
    for k, v in cnt.items():
        if v[0] or v[1] > 20 or v[2] > 40:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
            print("%-18s %-100s scratch %4d  accvgpr %4d  v_mov_b64 %4d  of %5d lines" % (os.path.basename(f), name[:100], *v))
