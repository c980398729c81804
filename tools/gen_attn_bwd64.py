#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 instruction streams of the long-sequence attention backward
(csrc/st_attn_bwd64_dkv.inc, csrc/st_attn_bwd64_dq.inc - inline-asm bodies of csrc/st_attn_bwd64.hip).

Why a generator: the two bodies of the attention backward (dK/dV: lane = key, dQ: lane = query; reference
transformer/Attention.py:82-90 under autograd) are instruction-issue bound, and what decides their speed is the ORDER of
the stream - which the compiler does not keep.  Here every instruction of the tile loop is placed:

  * one 32 x 32 (query, key) block per STEP, software-pipelined one block deep:
        step n = [ M2(n-1): the block's output contractions ] [ M1(n+1): scores + dP of the block after next ]
                 with V(n) - exp2, P * dP', two bf16 packs per score pair - spread between those matrix instructions
    (dK/dV: 16 MFMAs + 48 VALU + 32 LDS reads per step; dQ: 12 + 40 + 16);
  * the score accumulators start from -lse / -delta THROUGH THE MFMA C OPERAND (dK/dV body: the vectors are read from the
    tile's statistics in LDS straight into the accumulator registers; dQ body: a lane-constant register vector), K (dK/dV
    body) or Q (dQ body) pre-multiplied by scale * log2(e): s - lse and dP - delta cost no vector instruction;
  * two 32-register sets X / Y alias everything a block needs: Y holds block n-1's packed P / dS (read by M2 as B
    operands) in the first half of step n and becomes block n+1's accumulators in the second; V(n) packs X in place;
  * an 8-slot ring of 4-register A-operand fragments: the LDS read for the MFMA 8 ahead is issued behind each MFMA, so
    every wait is a counted lgkmcnt that does not stall;
  * 64-row tiles through a 3-buffer LDS ring (padded rows: 144 B, conflict-free for ds_read_b128 rows and
    ds_read_b64_tr_b16), register-staged with buffer loads whose range ends at the utterance's last row (rows past it
    read as zeros: no clamps, no masks in the dK/dV body; the dQ body masks the last key tile's exponentials), ONE
    barrier per tile.

The emitter tracks outstanding LDS / VMEM operations and inserts the counted s_waitcnt itself, and checks the gfx950
software hazards a hand-written stream must respect (hipcc's recogniser does not see inline asm; rules read off hipcc's
own output: MFMA result -> VALU / memory read 12 wait states, VALU write -> MFMA source 2, transcendental -> VALU 1).

usage: python tools/gen_attn_bwd64.py [out_dir]     (rewrites the .inc files, by default in csrc/; the output is deterministic)
"""
import os
import sys



def knob(name, default):
    """development switches (tools/dev/attn_bwd64_sweep.sh): BWD64_<NAME>=value; the committed .inc files are the defaults"""
    return int(os.environ.get("BWD64_" + name, default))


STR = 144            # bytes per LDS tile row (64 bf16 + 8 padding)
MAT = 64 * STR       # one 64-row tile
STAT = 2 * MAT       # dK/dV body: -lse[64], -delta[64] behind the two tiles
BUF = STAT + 512     # one ring buffer
NBUF = 3

# ---- physical scratch registers (v128..v255 are declared clobbered by the asm statement) ----------------------------
SET_A, SET_B = 128, 160          # each: s[16] then dp[16]
FRAG = 192                       # 8 slots x 4 registers
# (v224..v240 free: the staging registers are asm operands - the C++ prologue fetches tile 0 into them while the
# register-resident fragments are on their way)
A_G0, A_G1, A_H0, A_H1, A_ST = 241, 242, 243, 244, 245    # global byte offsets of this lane's chunks (matrix 0 / 1, stats)
W_T, W_ST = 246, 247             # LDS write addresses (tile chunk 0, statistic)
R_NAT, R_TR, R_STAT = 248, 249, 250
V_M = 251                        # dQ body: lk - first key of the last tile - 4 * hi
T0, T1, T2, T3 = 252, 253, 254, 255
# scalar scratch (declared clobbered)
S_SRD0, S_SRD1, S_SRD2 = 36, 40, 44
S_CNT, S_STEP0, S_STEP1, S_T0, S_T1 = 48, 49, 50, 51, 52
# dropout variants only: hash key, keep threshold, the two multipliers of lowbias32, the odd-lane mask (pair), statistic transform
S_KEY, S_THR, S_C1, S_C2, S_NT, S_PAR = 53, 54, 55, 56, 57, 58
DR_CTR, DR_W, DR_T = 224, 225, 233      # lane counter; 8 hash words; two temporaries (v224..v234 of the free v224..v240)
DR_ND = 236                             # dK/dV body: four -delta/k values of the register group in work (v236..v239)


def v(base, n=1):
    if isinstance(base, str):
        return base
    return "v%d" % base if n == 1 else "v[%d:%d]" % (base, base + n - 1)


def regs_of(base, n=1):
    if isinstance(base, str):
        return [base]
    return list(range(base, base + n))


class Emit:
    """Linear instruction emitter with LDS / VMEM wait tracking and hazard checks."""

    MFMA_TO_VALU = 12
    VALU_TO_MFMA = 2
    TRANS_TO_VALU = 1

    def __init__(self, uid):
        self.lines = []
        self.uid = uid
        self.t = 1000
        self.lds_seq = 0          # number of LDS operations issued
        self.lds_done = 0         # operations with seq < lds_done are complete
        self.lds_pend = {}        # register -> seq of the read that fills it
        self.vm_seq = 0
        self.vm_done = 0
        self.vm_pend = {}
        self.mfma_w = {}          # register -> time of the MFMA that last wrote it
        self.valu_w = {}          # register -> (time, is_trans)
        self.stats = dict(mfma=0, valu=0, lds=0, vmem=0, salu=0, wait=0, nop_states=0)

    # -- state at block boundaries ----------------------------------------------------------------------------------
    def snapshot(self):
        """State relative to `now`: outstanding reads in issue order, ages of register writes."""
        lds = {r: q - self.lds_done for r, q in self.lds_pend.items() if q >= self.lds_done}
        vm = {r: q - self.vm_done for r, q in self.vm_pend.items() if q >= self.vm_done}
        st = None if self.last_store_seq is None or self.last_store_seq < self.lds_done else self.last_store_seq - self.lds_done
        return dict(lds_n=self.lds_seq - self.lds_done, lds=lds, vm_n=self.vm_seq - self.vm_done, vm=vm, store=st,
                    mfma={r: self.t - w for r, w in self.mfma_w.items()},
                    valu={r: (self.t - w, tr) for r, (w, tr) in self.valu_w.items()})

    def restore(self, s):
        self.t = 100000
        self.lds_seq, self.lds_done, self.lds_pend = s["lds_n"], 0, dict(s["lds"])
        self.vm_seq, self.vm_done, self.vm_pend = s["vm_n"], 0, dict(s["vm"])
        self.last_store_seq = s["store"]
        self.mfma_w = {r: self.t - a for r, a in s["mfma"].items()}
        self.valu_w = {r: (self.t - a, tr) for r, (a, tr) in s["valu"].items()}

    @staticmethod
    def merge(states):
        """The state a block may assume when it is entered from several predecessors: identical outstanding operations
        (checked), the YOUNGEST age of every register write."""
        # the predecessor with the most operations in flight is the assumption (a counted wait for an operation that has
        # already completed is merely stricter than necessary); the others must be suffixes of it: the same operations,
        # the same number of operations behind each
        a = max(states, key=lambda s: (s["lds_n"], s["vm_n"]))
        for b in states:
            for key, cnt in (("lds", "lds_n"), ("vm", "vm_n")):
                assert a[cnt] >= b[cnt]
                sh = a[cnt] - b[cnt]
                for r, q in b[key].items():
                    assert a[key].get(r) == q + sh, "predecessors disagree on outstanding operations (%s, register %r)" % (key, r)
            if b["store"] is not None:
                assert a["store"] == b["store"] + (a["lds_n"] - b["lds_n"])
            elif a["store"] is not None:
                assert a["store"] < a["lds_n"] - b["lds_n"], "a store in flight on one path only"
        out = dict(a)
        out["mfma"], out["valu"] = {}, {}
        for s in states:
            for r, age in s["mfma"].items():
                out["mfma"][r] = min(age, out["mfma"].get(r, 1 << 30))
            for r, (age, tr) in s["valu"].items():
                o = out["valu"].get(r, (1 << 30, False))
                out["valu"][r] = (min(age, o[0]), tr or o[1])
        return out

    def raw(self, text):
        self.lines.append(text)

    def label(self, name):
        self.lines.append("%s_%s:" % (name, self.uid))

    def lab(self, name):
        return "%s_%s" % (name, self.uid)

    def comment(self, text):
        self.lines.append("; " + text)

    # -- waits and hazards -------------------------------------------------------------------------------------------
    def _wait_regs(self, regs):
        need_l, need_v = None, None
        for r in regs:
            s = self.lds_pend.get(r)
            if s is not None and s >= self.lds_done:
                n = self.lds_seq - s - 1
                need_l = n if need_l is None else min(need_l, n)
            s = self.vm_pend.get(r)
            if s is not None and s >= self.vm_done:
                n = self.vm_seq - s - 1
                need_v = n if need_v is None else min(need_v, n)
        if need_l is not None or need_v is not None:
            parts = []
            if need_v is not None:
                need_v = min(need_v, 63)
                parts.append("vmcnt(%d)" % need_v)
                self.vm_done = max(self.vm_done, self.vm_seq - need_v)
            if need_l is not None:
                need_l = min(need_l, 15)
                parts.append("lgkmcnt(%d)" % need_l)
                self.lds_done = max(self.lds_done, self.lds_seq - need_l)
            self.lines.append("s_waitcnt " + " ".join(parts))
            self.t += 1
            self.stats["wait"] += 1

    def wait_lds_writes(self):
        """All LDS operations issued so far that have no destination register (stores) must be complete: wait for the
        youngest of them (the operations behind it - fragment prefetches - may stay in flight)."""
        if self.last_store_seq is None or self.last_store_seq < self.lds_done:
            return
        n = min(self.lds_seq - self.last_store_seq - 1, 15)
        self.lines.append("s_waitcnt lgkmcnt(%d)" % n)
        self.lds_done = max(self.lds_done, self.lds_seq - n)
        self.t += 1
        self.stats["wait"] += 1

    last_store_seq = None

    def _nop(self, states):
        while states > 0:
            k = min(states, 16)
            self.lines.append("s_nop %d" % (k - 1))
            self.t += k
            self.stats["nop_states"] += k
            states -= k

    def _hazard_read(self, regs, by_mfma):
        need = 0
        for r in regs:
            if not by_mfma and r in self.mfma_w:
                need = max(need, self.MFMA_TO_VALU - (self.t - self.mfma_w[r] - 1))
            if r in self.valu_w:
                tw, trans = self.valu_w[r]
                lim = self.VALU_TO_MFMA if by_mfma else (self.TRANS_TO_VALU if trans else 0)
                need = max(need, lim - (self.t - tw - 1))
        if need > 0:
            self._nop(need)

    # -- instructions ------------------------------------------------------------------------------------------------
    def mfma(self, d, a, b, c, dn=16):
        """v_mfma_f32_32x32x16_bf16 d, a, b, c   (d, c: 16 registers or an operand string; a, b: 4)"""
        ra, rb, rc, rd = regs_of(a, 4), regs_of(b, 4), regs_of(c, 16), regs_of(d, 16)
        self._wait_regs(ra + rb + rc + rd)
        self._hazard_read(ra + rb, by_mfma=True)
        # C operand: a chain (c == d) needs nothing; anything else written by a VALU needs the VALU -> MFMA distance
        if rc != rd:
            self._hazard_read(rc, by_mfma=True)
        for r in ra + rb:
            assert r not in self.mfma_w or self.t - self.mfma_w[r] > 20, "MFMA result used as MFMA A/B operand too early: %r" % r
        self.lines.append("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (v(d, 16), v(a, 4), v(b, 4), v(c, 16)))
        for r in rd:
            self.mfma_w[r] = self.t
            self.valu_w.pop(r, None)
        self.t += 1
        self.stats["mfma"] += 1

    def valu(self, op, d, srcs, trans=False, dn=1, extra_reads=(), sdwa=False, dpp=False):
        """generic VALU: `op d, srcs...`; srcs are registers (int), operand strings or literal text (prefixed '='; a literal
        that starts with a blank is a modifier suffix, appended without a comma)"""
        rr = []
        txt = []
        suffix = ""
        for s in srcs:
            if isinstance(s, str) and s.startswith("= "):
                suffix += s[1:]
            elif isinstance(s, str) and s.startswith("="):
                txt.append(s[1:])
            else:
                rr += regs_of(s)
                txt.append(v(s))
        rr += list(extra_reads)
        rd = regs_of(d, dn) if d is not None else []
        self._wait_regs(rr + rd)
        self._hazard_read(rr, by_mfma=False)
        if sdwa or dpp:      # VALU write -> DPP / SDWA read of the same register: 2 wait states
            need = 0
            for r in rr:
                if r in self.valu_w:
                    need = max(need, 2 - (self.t - self.valu_w[r][0] - 1))
            if need > 0:
                self._nop(need)
        for r in rd:      # WAW behind an MFMA
            if r in self.mfma_w and self.t - self.mfma_w[r] - 1 < self.MFMA_TO_VALU:
                self._nop(self.MFMA_TO_VALU - (self.t - self.mfma_w[r] - 1))
        self.lines.append("%s %s%s" % (op, ", ".join(([v(d, dn)] if d is not None else []) + txt), suffix))
        for r in rd:
            self.valu_w[r] = (self.t, trans)
            self.mfma_w.pop(r, None)
        self.t += 1
        self.stats["valu"] += 1

    def salu(self, text):
        self.lines.append(text)
        self.t += 1
        self.stats["salu"] += 1

    def ds_read(self, op, d, dn, addr, off):
        assert 0 <= off < 65536, off
        rd = regs_of(d, dn)
        self._wait_regs(rd)          # a second read into a register whose first read is still in flight: keep them ordered (in-order return: harmless)
        self.lines.append("%s %s, %s offset:%d" % (op, v(d, dn), v(addr), off))
        for r in rd:
            self.lds_pend[r] = self.lds_seq
            self.mfma_w.pop(r, None)
            self.valu_w.pop(r, None)
        self.lds_seq += 1
        self.t += 1
        self.stats["lds"] += 1

    def ds_write(self, op, addr, data, dn, off):
        assert 0 <= off < 65536, off
        rr = regs_of(data, dn)
        self._wait_regs(rr)
        self._hazard_read(rr, by_mfma=False)
        self.lines.append("%s %s, %s offset:%d" % (op, v(addr), v(data, dn), off))
        self.last_store_seq = self.lds_seq
        self.lds_seq += 1
        self.t += 1
        self.stats["lds"] += 1

    def buffer_load(self, d, dn, voff, srd):
        rd = regs_of(d, dn)
        self._wait_regs(rd)
        op = {4: "buffer_load_dwordx4", 1: "buffer_load_dword"}[dn]
        self.lines.append("%s %s, %s, s[%d:%d], 0 offen" % (op, v(d, dn), v(voff), srd, srd + 3))
        for r in rd:
            self.vm_pend[r] = self.vm_seq
            self.mfma_w.pop(r, None)
            self.valu_w.pop(r, None)
        self.vm_seq += 1
        self.t += 1
        self.stats["vmem"] += 1

    def barrier(self):
        self.lines.append("s_barrier")
        self.t += 1

    def text(self):
        out = []
        for ln in self.lines:
            out.append('    "%s\\n\\t"' % ln)
        return "\n".join(out) + "\n"


# ------------------------------------------------------------------------------------------------------------------
# what the two bodies share: fragments, steps, joins
# ------------------------------------------------------------------------------------------------------------------
class Mf:
    """One MFMA of a step with the LDS reads that produce its A operand (and, for the first MFMA of a score chain of the
    dK/dV body, the reads that fill the accumulator with -lse / -delta: `init`, not before MFMA `init_after` of the step)."""

    def __init__(self, d, b, c, reads, init=None, init_after=0, tag="", a=None):
        self.d, self.b, self.c, self.reads, self.init, self.init_after, self.tag = d, b, c, reads, init, init_after, tag
        self.a = a        # a register-resident A operand (no LDS read, no fragment slot)


def tr_reads(buf, mat, blk, hf, dcol):
    """the two ds_read_b64_tr_b16 of rd_tr(tile, dcol * 32, blk * 32 + 16 * hf + 4 * hi) (st_attn_common.cuh)"""
    base = buf * BUF + mat * MAT + (32 * blk + 16 * hf) * STR + dcol * 64
    return [("ds_read_b64_tr_b16", 0, 2, R_TR, base), ("ds_read_b64_tr_b16", 2, 2, R_TR, base + 8 * STR)]


def nat_read(buf, mat, blk, t):
    """rd_nat(tile, blk * 32 + (lane & 31), t): one ds_read_b128"""
    return [("ds_read_b128", 0, 4, R_NAT, buf * BUF + mat * MAT + blk * 32 * STR + t * 32)]


def stat_reads(buf, blk, which, dst):
    """dK/dV body: the 16 accumulator start values of lane half hi: -lse (which = 0) or -delta (1) of queries blk * 32 + 8 g + 4 hi + e"""
    return [("ds_read_b128", dst + 4 * g, 4, R_STAT, buf * BUF + STAT + which * 256 + (32 * blk + 8 * g) * 4) for g in range(4)]


class Body:
    """Common step machinery; subclasses describe the two bodies."""
    WAIT_GROUP = knob("WAIT_GROUP", 4)        # one counted wait covers the A operands of this many MFMAs

    def __init__(self, uid, joins=None):
        self.e = Emit(uid)
        self.gidx = 0                   # fragment ring position of the next MFMA
        self.joins_in = joins or {}     # label -> (merged state, ring position) from the previous pass
        self.joins_out = {}

    def leave(self, name):
        """control may reach label `name` from here with the current state"""
        self.joins_out.setdefault(name, []).append((self.e.snapshot(), self.gidx % 8))

    def enter(self, name):
        """emit label `name`; continue with what every predecessor guarantees"""
        self.e.label(name)
        if name in self.joins_in:
            st, g = self.joins_in[name]
        else:                           # first pass: the first predecessor emitted so far
            st, g = self.joins_out[name][0]
        self.e.restore(st)
        self.gidx = g

    def merged(self):
        out = {}
        for name, lst in self.joins_out.items():
            gs = set(g for _, g in lst)
            assert len(gs) == 1, (name, gs)
            out[name] = (Emit.merge([st for st, _ in lst]), lst[0][1])
        return out

    def run(self, mfs, nxt, valu, extras=None, valu_from=1, valu_to=None, defer_from=None, barrier_after=None, pre_deferred=None):
        """mfs: this block's MFMAs; nxt: the MFMAs that follow (the A reads of every MFMA that has one are issued 8 such MFMAs
        ahead, i.e. behind this block's last ones; reads of nxt entries from index defer_from on are held back - data that
        becomes visible only after the block's barrier - and returned as closures, or, with barrier_after = k, emitted right
        behind the barrier this function then places behind MFMA k); valu: closures emitting one VALU each, spread evenly
        behind MFMAs valu_from..valu_to; extras: {k: [closures]} emitted behind MFMA k (1-based).
        self.gidx counts the MFMAs that consume a fragment slot."""
        e = self.e
        extras = extras or {}
        K = len(mfs)
        allm = mfs + nxt
        ring = [i for i, m in enumerate(allm) if m.a is None]          # indices (in allm) of the slot consumers
        rpos = {i: j for j, i in enumerate(ring)}
        g0 = self.gidx
        done_v = 0
        inits = {}
        for m in mfs:
            if m.init:
                inits.setdefault(m.init_after, []).extend(m.init)
        pending_inits = []
        deferred = list(pre_deferred or [])      # reads a predecessor block could not issue yet (same rule)
        holding = defer_from is not None
        held = defer_from if callable(defer_from) else (lambda i: i >= defer_from)      # by index in nxt
        vf = max(1, valu_from)
        vt = K if valu_to is None else valu_to

        def slot_of(i):
            return FRAG + 4 * ((g0 + rpos[i]) % 8)

        def do_valu(k):
            nonlocal done_v
            if k >= vf and valu:
                target = len(valu) if k >= vt else -(-len(valu) * (k - vf + 1) // (vt - vf + 1))
                while done_v < min(target, len(valu)):
                    valu[done_v]()
                    done_v += 1

        for k in range(0, K + 1):
            if k >= 1:
                m = mfs[k - 1]
                if (k - 1) % self.WAIT_GROUP == 0:      # one wait for the A operands of the next few MFMAs
                    cover = []
                    for j in range(k - 1, min(K, k - 1 + self.WAIT_GROUP)):
                        if mfs[j].a is None:
                            cover += regs_of(slot_of(j), 4)
                    e._wait_regs(cover)
                if m.a is None:
                    slot = slot_of(k - 1)
                    e.mfma(m.d, slot, m.b, m.c)
                else:
                    e.mfma(m.d, m.a, m.b, m.c)
                if knob("VALU_FIRST", 0):                # (switch) vector instructions in front of the LDS reads
                    do_valu(k)
                if m.a is None and rpos[k - 1] + 8 < len(ring):      # the slot is free: read the A operand of the consumer 8 ahead
                    tgt = ring[rpos[k - 1] + 8]
                    for (op, sub, n, addr, off) in allm[tgt].reads:
                        if holding and tgt >= K and held(tgt - K):
                            deferred.append(lambda op=op, d=slot + sub, n=n, addr=addr, off=off: e.ds_read(op, d, n, addr, off))
                        else:
                            e.ds_read(op, slot + sub, n, addr, off)
            if k in inits:
                pending_inits += inits[k]
            for (op, dst, n, addr, off) in pending_inits[:2]:      # at most two accumulator-start reads behind one MFMA
                e.ds_read(op, dst, n, addr, off)
            pending_inits = pending_inits[2:]
            for f in extras.get(k, []):
                f()
            if barrier_after is not None and k == barrier_after:
                e.wait_lds_writes()
                e.barrier()
                for f in deferred:
                    f()
                deferred, holding = [], False
            do_valu(k)
        assert not pending_inits
        assert done_v == len(valu), (done_v, len(valu))
        self.gidx = g0 + sum(1 for m in mfs if m.a is None)
        return deferred

    # ---- dropout (training mode): the keep decisions of one 32 x 32 block, as csrc/st_attn_common.cuh keep16 makes them -------
    # One lowbias32 hash of the 2 x 2 block counter yields four keep bytes.  A lane's 16 elements lie in 8 such blocks, each shared
    # with the neighbouring lane (its fixed index differs in bit 0): the lane hashes four counters (register groups g = 2 par,
    # 2 par + 1) and takes the other four from lane ^ 1 with a DPP move.  DR_W[0..3] end up as the words of groups 0 / 1
    # (index (g & 1) * 2 + hb), DR_W[4..7] as those of groups 2 / 3, pre-shifted so that the element's byte is BYTE_0 (e = 0)
    # or BYTE_<DR_E1> (e = 1) whatever the lane's parity.  DR_CTR = the lane's counter of (register group 2 par, hb 0) of the
    # current block; it advances by DR_STEP per block.
    def drop_words(self, L):
        e = self.e
        W, T = DR_W, DR_T
        for j, cj in enumerate(self.DR_CJ):                                  # counters -> hashes (own[j])
            L.append(lambda j=j, cj=cj: e.valu("v_add_u32", W + j, ["=0x%x" % cj, DR_CTR]))
            L.append(lambda j=j: e.valu("v_xor_b32", W + j, ["=s%d" % S_KEY, W + j]))
        for j in range(4):
            L.append(lambda j=j: e.valu("v_xor_b32_sdwa", W + j, [W + j, W + j, "= dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"], sdwa=True))
        for j in range(4):
            L.append(lambda j=j: e.valu("v_mul_lo_u32", W + j, [W + j, "=s%d" % S_C1]))
        for j in range(4):
            L.append(lambda j=j: e.valu("v_lshrrev_b32", T, ["=15", W + j]))
            L.append(lambda j=j: e.valu("v_xor_b32", W + j, [T, W + j]))
        for j in range(4):
            L.append(lambda j=j: e.valu("v_mul_lo_u32", W + j, [W + j, "=s%d" % S_C2]))
        for j in range(4):
            L.append(lambda j=j: e.valu("v_xor_b32_sdwa", W + j, [W + j, W + j, "= dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"], sdwa=True))
        for j in range(4):                                                    # nb[j] = own[j] of lane ^ 1
            L.append(lambda j=j: e.valu("v_mov_b32_dpp", W + 4 + j, [W + j, "= quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"], dpp=True))
        sh = self.DR_PRESHIFT
        for j in range(4):
            # groups 0 / 1: even lanes own them (own[j]), odd lanes got them from the neighbour and read them shifted;
            # groups 2 / 3: odd lanes own them and read them shifted, even lanes got them from the neighbour
            L.append(lambda j=j: e.valu("v_lshrrev_b32", T, ["=%d" % sh, W + 4 + j]))
            L.append(lambda j=j: e.valu("v_lshrrev_b32", T + 1, ["=%d" % sh, W + j]))
            L.append(lambda j=j: e.valu("v_cndmask_b32_e64", W + j, [W + j, T, "=s[%d:%d]" % (S_PAR, S_PAR + 1)]))
            L.append(lambda j=j: e.valu("v_cndmask_b32_e64", W + 4 + j, [W + 4 + j, T + 1, "=s[%d:%d]" % (S_PAR, S_PAR + 1)]))
        L.append(lambda: e.valu("v_add_u32", DR_CTR, ["=0x%x" % self.DR_STEP, DR_CTR]))

    def drop_keep(self, L, r):
        """vcc = keep decision of accumulator register r (r = 4 g + 2 hb + e)"""
        e = self.e
        g, hb, el = r >> 2, (r >> 1) & 1, r & 1
        w = DR_W + (4 if g >= 2 else 0) + (g & 1) * 2 + hb
        sel = "BYTE_0" if el == 0 else "BYTE_%d" % self.DR_E1
        L.append(lambda: e.valu("v_cmp_ge_u32_sdwa", None, ["=vcc", w, "=s%d src0_sel:%s src1_sel:DWORD" % (S_THR, sel)], sdwa=True))

    def drop_setup(self, key_op, lane_base_op, packed_nt_op):
        """scalar constants of the hash; the lane's counter base; thresh rides in the tile-count operand's upper half"""
        e = self.e
        e.salu("s_mov_b32 s%d, %s" % (S_KEY, key_op))
        e.salu("s_lshr_b32 s%d, %s, 16" % (S_THR, packed_nt_op))
        e.salu("s_mov_b32 s%d, 0x7feb352d" % S_C1)
        e.salu("s_mov_b32 s%d, 0x846ca68b" % S_C2)
        e.salu("s_mov_b32 s%d, 0xaaaaaaaa" % S_PAR)
        e.salu("s_mov_b32 s%d, 0xaaaaaaaa" % (S_PAR + 1))
        e.valu("v_mov_b32", DR_CTR, [lane_base_op])

    def muls(self, L, S, D, i):
        """dS = P * dP' for the register pair (2i, 2i + 1)"""
        e = self.e
        if knob("PKMUL", 0):
            L.append(lambda: e.valu("v_pk_mul_f32", D + 2 * i, ["=" + v(S + 2 * i, 2), "=" + v(D + 2 * i, 2)], dn=2,
                                    extra_reads=[S + 2 * i, S + 2 * i + 1, D + 2 * i, D + 2 * i + 1]))
        else:
            for r in (2 * i, 2 * i + 1):
                L.append(lambda r=r: e.valu("v_mul_f32", D + r, [S + r, D + r]))

    def prime(self, mfs):
        """pipeline fill: the A reads of the first 8 MFMAs and their accumulator-start reads"""
        e = self.e
        for k, m in enumerate([m for m in mfs if m.a is None][:8]):
            slot = FRAG + 4 * ((self.gidx + k) % 8)
            for (op, sub, n, addr, off) in m.reads:
                e.ds_read(op, slot + sub, n, addr, off)
        self.inits_now(mfs)

    def inits_now(self, mfs):
        for m in mfs:
            if m.init:
                for (op, dst, n, addr, off) in m.init:
                    self.e.ds_read(op, dst, n, addr, off)
                m.init = None

    @staticmethod
    def spread(fs, first=1):
        return {first + i: [f] for i, f in enumerate(fs)}

    def lane_addresses(self, ld0, ld1, lds):
        """per-lane byte offsets: staging chunk (row = tid >> 3 and row + 32, 16-byte chunk tid & 7) of the two streamed
        matrices, LDS write address of that chunk, LDS read bases of the row fragments / transposing fragments"""
        e = self.e
        e.valu("v_lshrrev_b32", T0, ["=3", self.TID])
        e.valu("v_and_b32", T1, ["=7", self.TID])
        e.valu("v_lshlrev_b32", T1, ["=4", T1])
        e.valu("v_mad_u32_u24", A_G0, [T0, "=" + ld0, T1])
        e.valu("v_mad_u32_u24", A_H0, [T0, "=" + ld1, T1])
        e.valu("v_add_u32", T2, ["=32", T0])
        e.valu("v_mad_u32_u24", A_G1, [T2, "=" + ld0, T1])
        e.valu("v_mad_u32_u24", A_H1, [T2, "=" + ld1, T1])
        e.valu("v_mov_b32", T3, ["=%d" % STR])
        e.valu("v_mad_u32_u24", W_T, [T0, T3, T1])
        e.valu("v_add_u32", W_T, ["=" + lds, W_T])
        e.valu("v_and_b32", T2, ["=63", self.TID])                            # lane
        e.valu("v_lshrrev_b32", T0, ["=5", T2])                               # hi
        e.valu("v_and_b32", T1, ["=31", T2])
        e.valu("v_mul_u32_u24", R_NAT, [T1, T3])
        e.valu("v_lshl_add_u32", R_NAT, [T0, "=4", R_NAT])
        e.valu("v_add_u32", R_NAT, ["=" + lds, R_NAT])
        # transposing reads: (4 hi + (t >> 2)) * 144 + ((lane >> 4) & 1) * 32 + 8 * (t & 3),  t = lane & 15
        e.valu("v_and_b32", T1, ["=15", T2])
        e.valu("v_lshrrev_b32", T1, ["=2", T1])
        e.valu("v_lshl_add_u32", T1, [T0, "=2", T1])
        e.valu("v_mul_u32_u24", R_TR, [T1, T3])
        e.valu("v_bfe_u32", T1, [T2, "=4", "=1"])
        e.valu("v_lshl_add_u32", R_TR, [T1, "=5", R_TR])
        e.valu("v_and_b32", T1, ["=3", T2])
        e.valu("v_lshl_add_u32", R_TR, [T1, "=3", R_TR])
        e.valu("v_add_u32", R_TR, ["=" + lds, R_TR])

    def descriptor(self, srd, base, ld, rows):
        """{base, bytes up to the end of the last row's 64-column slice, raw-buffer flags}"""
        e = self.e
        e.salu("s_mov_b64 s[%d:%d], %s" % (srd, srd + 1, base))
        e.salu("s_sub_u32 s%d, %s, 1" % (S_T0, rows))
        e.salu("s_mul_i32 s%d, s%d, %s" % (srd + 2, S_T0, ld))
        e.salu("s_add_u32 s%d, s%d, 128" % (srd + 2, srd + 2))
        e.salu("s_mov_b32 s%d, 0x00020000" % (srd + 3))

    def advance_srd(self, srd, step):
        e = self.e
        e.salu("s_add_u32 s%d, s%d, %s" % (srd, srd, step))
        e.salu("s_addc_u32 s%d, s%d, 0" % (srd + 1, srd + 1))
        e.salu("s_sub_u32 s%d, s%d, %s" % (srd + 2, srd + 2, step))
        e.salu("s_max_i32 s%d, s%d, 0" % (srd + 2, srd + 2))

    def finish(self):
        e = self.e
        e.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e._nop(12)        # the compiler's code reads the accumulators next


# ------------------------------------------------------------------------------------------------------------------
# dK / dV body.  operands (in / out): %0 %1 = dk[0..1], %2 %3 = dv[0..1] (f32x16); %4..%7 = the staging registers of the
# streamed tiles (Q rows 0-31 / 32-63, dO likewise: this thread's 16-byte chunks; the C++ prologue has loaded tile 0 into them),
# %8 = the staging register of the statistics; (in): %9..%12 = K fragments * scale * log2 e, %13..%16 = V fragments;
# %17 = threadIdx.x; %18 = Q base (utterance row 0, head column 0), %19 = dO base, %20 = the wave's statistic base (lse for
# even waves, delta for odd ones); %21 = ldq * 2, %22 = lddo * 2 (bytes); %23 = lq; %24 = tiles (64 queries each); %25 = LDS
# byte address of the ring
# ------------------------------------------------------------------------------------------------------------------
class DKV(Body):
    G = ["%4", "%5", "%6", "%7"]
    GST = "%8"
    TID = "%17"
    DROP = False
    # dropout variant (class DKVDrop): %24 = tiles | thresh << 16, %26 = hash key (s), %27 = the lane's counter base
    # ((8 (lane & 1) + 2 hi) << 15) + (key >> 1) + (b H + h) * 0x85ebca6b (v), %28 / %29 = multiplier / addend that turn this wave's
    # raw statistic into the accumulator start value (s): (-1, log2 keep-scale) for the lse waves, (-1 / keep-scale, 0) for the
    # delta waves
    DR_CJ, DR_STEP, DR_PRESHIFT, DR_E1 = [0, 1 << 15, 4 << 15, 5 << 15], 1 << 19, 8, 2

    def m2(self, buf, blk, Y):
        out = []
        for mat, acc0, yoff, tag in ((1, 2, 0, "dV"), (0, 0, 16, "dK")):
            for hf in range(2):
                for d in range(2):
                    out.append(Mf("%%%d" % (acc0 + d), Y + yoff + 4 * hf, "%%%d" % (acc0 + d), tr_reads(buf, mat, blk, hf, d), tag=tag))
        return out

    def m1(self, buf, blk, Y):
        out = []
        for mat, yoff, b0, which, after in ((0, 0, 9, 0, 4), (1, 16, 13, 1, 8)):
            for t in range(4):
                out.append(Mf(Y + yoff, "%%%d" % (b0 + t), Y + yoff, nat_read(buf, mat, blk, t),
                              init=stat_reads(buf, blk, which, Y + yoff) if t == 0 else None, init_after=after,
                              tag="S" if mat == 0 else "dP"))
        return out

    def vlist(self, X, stat=None):
        """stat = (buffer, block) of the statistics the step's scores used (dropout variant: the -delta / k values are read again
        for the dropped pairs, four at a time)"""
        e = self.e
        S, D = X, X + 16
        L = [(lambda r=r: e.valu("v_exp_f32", S + r, [S + r], trans=True)) for r in range(16)]
        if self.DROP:
            # P k = exp2(.) (the keep-scale k rides in the accumulator start value), dS = P k (M ? dP - delta / k : -delta / k),
            # and dV takes M ? P k : 0
            H = []
            self.drop_words(H)
            L = H + L
            nd = stat_reads(stat[0], stat[1], 1, DR_ND)
            for g in range(4):
                op, dst, n, addr, off = nd[g]
                L.append(lambda op=op, n=n, addr=addr, off=off: e.ds_read(op, DR_ND, n, addr, off))
                for r in range(4 * g, 4 * g + 4):
                    self.drop_keep(L, r)
                    L.append(lambda r=r: e.valu("v_cndmask_b32", D + r, [DR_ND + (r & 3), D + r, "=vcc"]))
                    L.append(lambda r=r: e.valu("v_mul_f32", D + r, [S + r, D + r]))
                    L.append(lambda r=r: e.valu("v_cndmask_b32", S + r, ["=0", S + r, "=vcc"]))
                for i in (2 * g, 2 * g + 1):
                    L.append(lambda i=i: e.valu("v_cvt_pk_bf16_f32", S + i, [S + 2 * i, S + 2 * i + 1]))
                    L.append(lambda i=i: e.valu("v_cvt_pk_bf16_f32", D + i, [D + 2 * i, D + 2 * i + 1]))
            return L
        for i in range(8):
            self.muls(L, S, D, i)
            L.append(lambda i=i: e.valu("v_cvt_pk_bf16_f32", S + i, [S + 2 * i, S + 2 * i + 1]))
            L.append(lambda i=i: e.valu("v_cvt_pk_bf16_f32", D + i, [D + 2 * i, D + 2 * i + 1]))
        return L

    def loads(self):
        e = self.e
        G, GST = self.G, self.GST
        return [lambda: e.buffer_load(G[0], 4, A_G0, S_SRD0), lambda: e.buffer_load(G[1], 4, A_G1, S_SRD0),
                lambda: e.buffer_load(G[2], 4, A_H0, S_SRD1), lambda: e.buffer_load(G[3], 4, A_H1, S_SRD1),
                lambda: e.buffer_load(GST, 1, A_ST, S_SRD2),
                lambda: self.advance_srd(S_SRD0, "s%d" % S_STEP0), lambda: self.advance_srd(S_SRD1, "s%d" % S_STEP1),
                lambda: self.advance_srd(S_SRD2, "256")]

    def writes(self, buf):
        e = self.e
        G, GST = self.G, self.GST
        neg = ([lambda: e.valu("v_mul_f32", GST, ["=%28", GST]), lambda: e.valu("v_add_f32", GST, ["=%29", GST])] if self.DROP
               else [lambda: e.valu("v_xor_b32", GST, ["=0x80000000", GST])])
        return neg + [
                lambda: e.ds_write("ds_write_b128", W_T, G[0], 4, buf * BUF),
                lambda: e.ds_write("ds_write_b128", W_T, G[1], 4, buf * BUF + 32 * STR),
                lambda: e.ds_write("ds_write_b128", W_T, G[2], 4, buf * BUF + MAT),
                lambda: e.ds_write("ds_write_b128", W_T, G[3], 4, buf * BUF + MAT + 32 * STR),
                lambda: e.ds_write("ds_write_b32", W_ST, GST, 1, buf * BUF)]

    def prologue(self):
        e = self.e
        e.comment("dK/dV body: descriptors, lane offsets")
        e.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.descriptor(S_SRD0, "%18", "%21", "%23")
        self.descriptor(S_SRD1, "%19", "%22", "%23")
        e.salu("s_mov_b64 s[%d:%d], %%20" % (S_SRD2, S_SRD2 + 1))
        e.salu("s_lshl_b32 s%d, %%23, 2" % (S_SRD2 + 2))                     # lq * 4 bytes of statistics
        e.salu("s_mov_b32 s%d, 0x00020000" % (S_SRD2 + 3))
        e.salu("s_lshl_b32 s%d, %%21, 6" % S_STEP0)                          # bytes per 64-row tile
        e.salu("s_lshl_b32 s%d, %%22, 6" % S_STEP1)
        nt = "%24"
        if self.DROP:
            self.drop_setup("%26", "%27", "%24")
            e.salu("s_and_b32 s%d, %%24, 0xffff" % S_NT)
            nt = "s%d" % S_NT
        e.salu("s_sub_u32 s%d, %s, 1" % (S_CNT, nt))                          # tiles after the current one
        for f in self.loads()[5:]:                                            # tile 0 is in the staging registers: on to tile 1
            f()
        self.lane_addresses("%21", "%22", "%25")
        e.valu("v_lshlrev_b32", A_ST, ["=2", T2])                             # statistic of query `lane` of the tile
        e.valu("v_and_b32", W_ST, ["=127", self.TID])
        e.valu("v_lshlrev_b32", W_ST, ["=2", W_ST])
        e.valu("v_add_u32", W_ST, ["=%25", W_ST])
        e.valu("v_add_u32", W_ST, ["=%d" % STAT, W_ST])
        e.valu("v_lshlrev_b32", R_STAT, ["=4", T0])
        e.valu("v_add_u32", R_STAT, ["=%25", R_STAT])
        e._nop(5)                                                             # SALU write -> VMEM descriptor read

    def build(self):
        e = self.e
        A, B = SET_A, SET_B
        self.prologue()
        for f in self.writes(0):       # tile 0 (loaded by the C++ prologue) -> buffer 0; tile 1 -> staging registers
            f()
        for f in self.loads():
            f()
        e.wait_lds_writes()
        e.barrier()
        # M1(0) into set A; step 0 without M2: M1(1) into set B, V(0) on A, tile 1 -> buffer 1
        m10, m11 = self.m1(0, 0, A), self.m1(0, 1, B)
        self.gidx = 0
        self.prime(m10)
        self.run(m10, m11, [])
        self.inits_now(m11)
        self.run(m11, self.m2(0, 0, A), self.vlist(A, (0, 0)), extras=self.spread(self.writes(1)))
        e.wait_lds_writes()
        e.barrier()
        e.salu("s_cmp_eq_u32 s%d, 0" % S_CNT)
        self.leave("LASTODD0")
        e.salu("s_cbranch_scc1 %s" % e.lab("LASTODD0"))
        self.leave("ODD0")
        for p in range(3):
            q = (p + 1) % 3
            # odd step of tile t (t % 3 == p): X = B, Y = A.  M2(2t): buffer p block 0; M1(2t + 2): buffer q block 0
            self.enter("ODD%d" % p)
            self.run(self.m2(p, 0, A) + self.m1(q, 0, A), self.m2(p, 1, B), self.vlist(B, (p, 1)), extras=self.spread(self.loads()))
            e.salu("s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT))
            self.leave("EVEN%d" % q)
            # even step of tile t + 1 (buffer q): X = A, Y = B.  M2(2t + 1): buffer p block 1; M1(2t + 3): buffer q block 1
            self.enter("EVEN%d" % q)
            self.run(self.m2(p, 1, B) + self.m1(q, 1, B), self.m2(q, 0, A), self.vlist(A, (q, 0)), extras=self.spread(self.writes((q + 1) % 3)))
            e.wait_lds_writes()
            e.barrier()
            e.salu("s_cmp_eq_u32 s%d, 0" % S_CNT)
            self.leave("LASTODD%d" % q)
            e.salu("s_cbranch_scc1 %s" % e.lab("LASTODD%d" % q))
            self.leave("ODD%d" % q)
            if p == 2:
                e.salu("s_branch %s" % e.lab("ODD0"))
        for p in range(3):             # the last tile's odd step (no M1) and the final M2
            self.enter("LASTODD%d" % p)
            nxt = self.m2(p, 1, B)
            self.run(self.m2(p, 0, A), nxt, self.vlist(B, (p, 1)))
            self.run(nxt, [], [])
            self.leave("END")
            if p < 2:
                e.salu("s_branch %s" % e.lab("END"))
        self.enter("END")
        self.finish()
        return e


# ------------------------------------------------------------------------------------------------------------------
# dQ body.  operands (in / out): %0 %1 = dq[0..1]; %2..%5 = the staging registers of the streamed tiles (K rows 0-31 / 32-63, V
# likewise; tile 0 loaded by the C++ prologue); (in): %6..%9 = Q fragments * scale * log2 e, %10..%13 = dO fragments; %14 = -lse
# (16 equal registers), %15 = -delta; %16 = threadIdx.x; %17 = K base, %18 = V base; %19 = ldk * 2, %20 = ldv * 2; %21 = lk;
# %22 = tiles (64 keys each); %23 = LDS byte address of the ring
# ------------------------------------------------------------------------------------------------------------------
class DQ(Body):
    G = ["%2", "%3", "%4", "%5"]
    TID = "%16"
    DROP = False
    NT = "%22"
    # dropout variant (class DQDrop): %22 = tiles | thresh << 16, %24 = hash key (s), %25 = the lane's counter base
    # ((q >> 1) << 15) + 8 (lane & 1) + 2 hi + (b H + h) * 0x85ebca6b (v), %26 = -delta / keep-scale of the lane's query (v);
    # %14 then holds -lse + log2(keep-scale), %15 = -delta / keep-scale (the C++ prologue folds the scale in)
    DR_CJ, DR_STEP, DR_PRESHIFT, DR_E1 = [0, 1, 4, 5], 16, 16, 1

    def m2(self, buf, blk, Y):
        return [Mf("%%%d" % d, Y + 16 + 4 * hf, "%%%d" % d, tr_reads(buf, 0, blk, hf, d), tag="dQ") for hf in range(2) for d in range(2)]

    def m1(self, buf, blk, Y):
        out = []
        for mat, yoff, b0, c0 in ((0, 0, 6, "%14"), (1, 16, 10, "%15")):
            for t in range(4):
                out.append(Mf(Y + yoff, "%%%d" % (b0 + t), c0 if t == 0 else Y + yoff, nat_read(buf, mat, blk, t), tag="S" if mat == 0 else "dP"))
        if knob("DQ_CHAIN_IL", 0):       # (switch) the two accumulation chains interleaved
            out = [out[i // 2 + 4 * (i % 2)] for i in range(8)]
        return out

    def vlist(self, X, mask_blk=None):
        e = self.e
        S, D = X, X + 16
        L = [(lambda r=r: e.valu("v_exp_f32", S + r, [S + r], trans=True)) for r in range(16)]
        if mask_blk is not None:      # keys past the end: P = 0 (the exponential itself may be inf there)
            for r in range(16):
                row = 32 * mask_blk + (r & 3) + 8 * (r >> 2)
                L.append(lambda row=row: e.valu("v_cmp_lt_i32", None, ["=vcc", "=%d" % row, V_M]))
                L.append(lambda r=r: e.valu("v_cndmask_b32", S + r, ["=0", S + r, "=vcc"]))
        if self.DROP:
            # dS = P k (M ? dP - delta / k : -delta / k): the exponential already carries the keep-scale k (folded into -lse),
            # the gradient accumulator started at -delta / k; a dropped pair keeps that start value
            H = []
            self.drop_words(H)
            L = H + L          # the hashes need no matrix result: first
            for i in range(8):
                for r in (2 * i, 2 * i + 1):
                    self.drop_keep(L, r)
                    L.append(lambda r=r: e.valu("v_cndmask_b32", D + r, ["=%26", D + r, "=vcc"]))
                self.muls(L, S, D, i)
                L.append(lambda i=i: e.valu("v_cvt_pk_bf16_f32", D + i, [D + 2 * i, D + 2 * i + 1]))
            return L
        for i in range(8):
            self.muls(L, S, D, i)
            L.append(lambda i=i: e.valu("v_cvt_pk_bf16_f32", D + i, [D + 2 * i, D + 2 * i + 1]))
        return L

    def loads(self):
        e = self.e
        G = self.G
        return [lambda: e.buffer_load(G[0], 4, A_G0, S_SRD0), lambda: e.buffer_load(G[1], 4, A_G1, S_SRD0),
                lambda: e.buffer_load(G[2], 4, A_H0, S_SRD1), lambda: e.buffer_load(G[3], 4, A_H1, S_SRD1),
                lambda: self.advance_srd(S_SRD0, "s%d" % S_STEP0), lambda: self.advance_srd(S_SRD1, "s%d" % S_STEP1)]

    def writes(self, buf):
        e = self.e
        G = self.G
        return [lambda: e.ds_write("ds_write_b128", W_T, G[0], 4, buf * BUF),
                lambda: e.ds_write("ds_write_b128", W_T, G[1], 4, buf * BUF + 32 * STR),
                lambda: e.ds_write("ds_write_b128", W_T, G[2], 4, buf * BUF + MAT),
                lambda: e.ds_write("ds_write_b128", W_T, G[3], 4, buf * BUF + MAT + 32 * STR)]

    def prologue(self):
        e = self.e
        e.comment("dQ body: descriptors, lane offsets")
        e.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.descriptor(S_SRD0, "%17", "%19", "%21")
        self.descriptor(S_SRD1, "%18", "%20", "%21")
        e.salu("s_lshl_b32 s%d, %%19, 6" % S_STEP0)
        e.salu("s_lshl_b32 s%d, %%20, 6" % S_STEP1)
        nt = self.NT
        if self.DROP:
            self.drop_setup("%24", "%25", "%22")
            e.salu("s_and_b32 s%d, %%22, 0xffff" % S_NT)
            nt = "s%d" % S_NT
        e.salu("s_sub_u32 s%d, %s, 1" % (S_CNT, nt))
        e.salu("s_lshl_b32 s%d, s%d, 6" % (S_T1, S_CNT))                     # first key of the last tile
        e.salu("s_sub_u32 s%d, %%21, s%d" % (S_T1, S_T1))                    # keys of the last tile that exist
        for f in self.loads()[4:]:                                            # tile 0 is in the staging registers: on to tile 1
            f()
        self.lane_addresses("%19", "%20", "%23")
        e.valu("v_lshlrev_b32", T1, ["=2", T0])                               # 4 hi
        e.valu("v_sub_u32", V_M, ["=s%d" % S_T1, T1])                         # key row r of block blk exists iff 32 blk + row(r) < V_M
        e._nop(5)

    def even_step(self, p, q, mask):
        """even step of the tile in buffer q (the tile before it sits in buffer p): X = A, Y = B"""
        A, B = SET_A, SET_B
        nxt = self.m2(q, 0, A) + (self.m2(q, 1, B) if mask else self.m1((q + 1) % 3, 0, A)[:4])
        return self.run(self.m2(p, 1, B) + self.m1(q, 1, B), nxt, self.vlist(A, mask_blk=0 if mask else None),
                        extras=None if mask else self.spread(self.writes((q + 1) % 3)), defer_from=None if mask else 4)

    def last_odd(self, p):
        """the last tile's odd step (V masked, no M1) and the final M2 as ONE block of 8 MFMAs: their A fragments were all
        requested by the step before; V(last block) must be packed before the fifth MFMA reads it"""
        A, B = SET_A, SET_B
        self.run(self.m2(p, 0, A) + self.m2(p, 1, B), [], self.vlist(B, mask_blk=1), valu_to=4)

    def build(self):
        e = self.e
        A, B = SET_A, SET_B
        self.prologue()
        for f in self.writes(0):
            f()
        for f in self.loads():
            f()
        e.wait_lds_writes()
        e.barrier()
        m10, m11 = self.m1(0, 0, A), self.m1(0, 1, B)
        self.gidx = 0
        self.prime(m10)
        self.run(m10, m11, [])
        e.salu("s_cmp_eq_u32 s%d, 0" % S_CNT)
        self.leave("FIRSTLAST")
        e.salu("s_cbranch_scc1 %s" % e.lab("FIRSTLAST"))
        # step 0 (no M2): M1(1) into B, V(0) on A, tile 1 -> buffer 1; the first score fragments of tile 1 after the barrier
        deferred = self.run(m11, self.m2(0, 0, A) + self.m1(1, 0, A)[:4], self.vlist(A), extras=self.spread(self.writes(1)), defer_from=4)
        e.wait_lds_writes()
        e.barrier()
        for f in deferred:
            f()
        self.leave("ODD0")
        for p in range(3):
            q = (p + 1) % 3
            # odd step of tile t (buffer p): X = B, Y = A.  M2(2t): buffer p block 0; M1(2t + 2): buffer q block 0
            self.enter("ODD%d" % p)
            self.run(self.m2(p, 0, A) + self.m1(q, 0, A), self.m2(p, 1, B) + self.m1(q, 1, B)[:4], self.vlist(B), extras=self.spread(self.loads()))
            e.salu("s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT))
            e.salu("s_cmp_eq_u32 s%d, 0" % S_CNT)
            self.leave("EVENLAST%d" % q)
            e.salu("s_cbranch_scc1 %s" % e.lab("EVENLAST%d" % q))
            self.leave("EVEN%d" % q)
            self.enter("EVEN%d" % q)
            deferred = self.even_step(p, q, mask=False)
            e.wait_lds_writes()
            e.barrier()
            for f in deferred:
                f()
            self.leave("ODD%d" % q)
            if p == 2:
                e.salu("s_branch %s" % e.lab("ODD0"))
        for q in range(3):             # the last tile: even step with V masked (block 0), odd step (block 1, no M1), final M2
            self.enter("EVENLAST%d" % q)
            assert not self.even_step((q + 2) % 3, q, mask=True)
            self.last_odd(q)
            self.leave("END")
            e.salu("s_branch %s" % e.lab("END"))
        self.enter("FIRSTLAST")        # a one-tile problem
        self.run(m11, self.m2(0, 0, A) + self.m2(0, 1, B), self.vlist(A, mask_blk=0))
        self.last_odd(0)
        self.leave("END")
        self.enter("END")
        self.finish()
        return e


# (A forward stream in the same style - class FWD: scores of block n + 1 first in the step, exponentials of block n, P V and
# the row-sum MFMAs of block n - 1; bit-identical to st_attn64.hip's kernel - was generated and measured in round 4: 32.8 us
# against 33.3 us at the config-2 encoder shape, 86.5 against 81.7 us on 16 x 2048 uniform utterances.  With ten MFMAs and 24
# vector instructions per 32 x 32 block the compiler-scheduled kernel at three workgroups per CU is already at ~57 % of the
# matrix rate the chip sustains at the 1.5-1.7 GHz it clocks to under this load, and what is left of the launch is per-item
# prologue / epilogue and the granularity of 816 items on 512 or 768 slots - nothing an instruction order fixes.  Not kept.)

class DQDrop(DQ):
    DROP = True


class DKVDrop(DKV):
    DROP = True


def generate(cls):
    joins = None
    text = None
    for _ in range(6):
        body = cls("%=", joins)
        e = body.build()
        joins = body.merged()
        new = e.text()
        if new == text:
            return e
        text = new
    raise RuntimeError("the join states did not converge")


def main(out_dir=None):
    """out_dir: where the .inc files go (default: the product's csrc/; tests/test_abi_cpu.py generates into a scratch
    directory and compares with the committed files)"""
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = out_dir or os.path.join(os.path.dirname(here), "speech-tranformer-pytorch_amd", "csrc")
    for name, cls in (("dkv", DKV), ("dq", DQ), ("dkv_drop", DKVDrop), ("dq_drop", DQDrop)):
        e = generate(cls)
        head = ("// GENERATED by tools/gen_attn_bwd64.py - do not edit.  Instruction stream of the %s body of csrc/st_attn_bwd64.hip\n"
                "// (in the text: %d MFMA, %d VALU, %d LDS, %d VMEM, %d SALU, %d waits, %d hazard wait states)\n"
                % (name, e.stats["mfma"], e.stats["valu"], e.stats["lds"], e.stats["vmem"], e.stats["salu"], e.stats["wait"],
                   e.stats["nop_states"]))
        path = os.path.join(csrc, "st_attn_bwd64_%s.inc" % name)
        with open(path, "w") as f:
            f.write(head + e.text())
        print(path, e.stats, "lines", len(e.lines))
    # the registers the streams own (clobber lists of the asm statements; the dropout variants own seven more scalar registers)
    for fname, s_hi in (("st_attn_bwd64_clobbers.inc", 53), ("st_attn_bwd64_clobbers_drop.inc", 60)):
        with open(os.path.join(csrc, fname), "w") as f:
            f.write("// GENERATED by tools/gen_attn_bwd64.py - do not edit.  Registers the hand-scheduled streams use as scratch.\n")
            f.write('"memory", "scc", "vcc",\n')
            f.write(", ".join('"s%d"' % r for r in range(36, s_hi)) + ",\n")
            regs = ['"v%d"' % r for r in range(128, 256)]
            for i in range(0, len(regs), 16):
                f.write(", ".join(regs[i:i + 16]) + ("," if i + 16 < len(regs) else "") + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
