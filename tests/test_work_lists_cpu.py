"""Host logic of the attention work lists (st_amd.functional.attn_work): every (utterance, 128-row tile)
appears exactly once and the list is ordered by decreasing cost (longest-first list scheduling)."""
import torch

from st_amd.functional import Rows, attn_work


def _decode(w):
    return [(int(v) >> 16, int(v) & 0xffff) for v in w.tolist()]


def test_work_lists_cover_every_tile_once_and_sort_by_cost():
    lq = torch.tensor([50, 33, 1, 47, 129])
    lk = torch.tensor([1000, 517, 130, 64, 128])
    q_rows, k_rows = Rows.packed(lq, "cpu"), Rows.packed(lk, "cpu")
    for qr, kr, causal, a, b in ((k_rows, k_rows, False, lk, lk), (q_rows, q_rows, True, lq, lq),
                                 (q_rows, k_rows, False, lq, lk)):
        from st_amd import native as nv
        for H in (4, 3):          # 4 heads: two XCD-affinity groups; 3 heads: no grouping (one plain longest-first list)
            wf, wq, wk = attn_work(qr, kr, causal, 64, H)
            ng = 8 // H if H in (1, 2, 4, 8) else 1
            Rf, Rq, Rk = (nv.attn_tile_rows(w, 64, int(a.max()), int(b.max()), causal) for w in range(3))
            assert Rf in (128, 256) and Rq in (128, 256) and Rk in (128, 256)

            def cost_q(R):
                def f(i, t):
                    seen = min(int(b[i]), (t + 1) * R) if causal else int(b[i])
                    return (seen + 63) // 64
                return f

            def cost_k(i, t):
                q_begin = (t * Rk // 64) * 64 if causal else 0
                return (int(a[i]) - q_begin + 63) // 64

            for w, R, L, cf in ((wf, Rf, a, cost_q(Rf)), (wq, Rq, a, cost_q(Rq)), (wk, Rk, b, cost_k)):
                assert w.dtype == torch.int32 and w.numel() % ng == 0
                want = {(i, t) for i in range(len(L)) for t in range((int(L[i]) + R - 1) // R)}
                got = [it for it in _decode(w) if it[1] != 0xffff]              # (no-op padding entries: tile 0xffff)
                assert len(got) == len(want) and set(got) == want
                owner = {}
                for g in range(ng):      # group g holds the positions = g (mod ng): each sorted by cost, utterances disjoint
                    items = [it for it in _decode(w)[g::ng] if it[1] != 0xffff]
                    c = [cf(*it) for it in items]
                    assert c == sorted(c, reverse=True) and (not c or min(c) >= 1)
                    for i, _ in items:
                        assert owner.setdefault(i, g) == g, "an utterance is served by two XCD groups"
            # cached on the query layout
            assert attn_work(qr, kr, causal, 64, H)[0] is wf


def test_dropout_hash_statistics():
    """The counter-based dropout masks (tests/_emul mirrors csrc/st_common.cuh bit for bit; the GPU tests pin
    that): keep rate = 1 - round(256 p)/256, different salts / seeds give independent masks."""
    from tests import _emul as em
    seed = torch.tensor([42], dtype=torch.int32)
    for p in (0.1, 0.2, 0.5):
        d = em.Drop(seed, 7, p)
        keep = em.keep_rc(d, torch.arange(2000), torch.arange(256), 256)
        want = 1.0 - d.thresh / 256.0
        assert abs(keep.float().mean().item() - want) < 4e-3
        assert abs(d.scale * want - 1.0) < 1e-6                     # unbiased for the realised keep rate
        assert (keep.float().mean(0) - want).abs().max() < 0.05      # no dead / always-on columns
        assert (keep.float().mean(1) - want).abs().max() < 0.12      # ... or rows
        other = em.keep_rc(em.Drop(seed, 8, p), torch.arange(2000), torch.arange(256), 256)
        agree = (keep == other).float().mean().item()
        assert abs(agree - (want * want + (1 - want) ** 2)) < 6e-3   # independent masks
        kq = em.keep_qk(d, 3, 300, 500)
        assert abs(kq.float().mean().item() - want) < 6e-3
        assert abs((kq == em.keep_qk(d, 4, 300, 500)).float().mean().item() - (want * want + (1 - want) ** 2)) < 8e-3


def test_deferred_weight_gradients_split_wide_and_grouped_launches():
    """Host logic of the deferred weight gradients (st_amd.functional): decoder-sized problems leave with the flush at the
    end of the decoder's backward, encoder-sized ones stay pending until the final flush and go out as ONE wide-tile
    launch whose token splits keep it within one workgroup per CU; the results equal the direct launches."""
    from st_amd import functional as F_
    from st_amd import native as nv
    from tests import _emul as em
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g).bfloat16()
    big, small = F_._Deferred.WIDE_ROWS + 100, 300
    shapes = [(small, 256, 256), (big, 256, 512), (small, 256, 768), (big, 256, 1024), (big, 1024, 256), (big, 80, 256)]
    calls = []
    with em.emulated_kernels():
        orig = nv.wgrad_group

        def spy(problems, wide=False):
            problems = list(problems)
            calls.append((wide, [(p[0].shape[0], p[0].shape[1], p[5], p[4]) for p in problems]))
            return orig(problems, wide=wide)

        nv.wgrad_group = spy
        try:
            outs, refs = [], []
            with F_.deferred_wgrads(True):
                for i, (m, k, n) in enumerate(shapes):
                    x, dy = rnd(m, k), rnd(m, n)
                    gw, gb = torch.zeros(n, k), torch.zeros(n)
                    F_.wgrad(dy, x, gw, gB=gb)
                    outs.append((gw, gb))
                    refs.append((dy.float().t() @ x.float(), dy.float().sum(0)))
                    if i == 2:
                        F_.flush_deferred_wgrads(final=False)       # what EmbedFn.backward does
                        assert [c[0] for c in calls] == [False] and len(calls[0][1]) == 2
                        assert len(F_._Deferred.pending) == 1       # the encoder-sized one waits
            assert not F_._Deferred.pending
        finally:
            nv.wgrad_group = orig
    assert [c[0] for c in calls] == [False, True]
    wide = calls[1][1]
    assert sorted(p[0] for p in wide) == [big] * 4
    tiles = sum(-(-k // 256) * -(-n // 256) for _, k, n, _ in wide)
    # one split count per launch, the launch within one round of the 256 CUs, at least 256 tokens per item (functional._wide_plan)
    sp = wide[0][3]
    assert len({p[3] for p in wide}) == 1 and 1 <= sp and tiles * sp <= 256 and big // sp >= 256
    for (gw, gb), (rw, rb) in zip(outs, refs):
        assert torch.allclose(gw, rw, rtol=1e-3, atol=1e-2) and torch.allclose(gb, rb, rtol=1e-3, atol=1e-2)


def test_wide_weight_gradient_plan_follows_the_measured_optima():
    """functional._wide_plan's cost model (round 6) against what tools/dev/wgrad_small_m.py measured on the MI355X: the splits that
    finish first for one encoder layer's 12 tiles (a DP step flushes per finished layer) and for the whole encoder's 85, at a
    4-utterance shard and at the full batch; config 3's two launches as round 5 planned them."""
    import torch
    from st_amd import functional as F_
    def probs(rows, shapes):
        return [(torch.empty(rows, k, dtype=torch.bfloat16), torch.empty(rows, n, dtype=torch.bfloat16), None, None, 1, n) for (n, k) in shapes]
    layer = ((768, 256), (256, 256), (1024, 256), (256, 1024))
    enc = layer * 6 + ((512, 256),) * 6
    pick = lambda rows, shapes: [l[0][4] for l in F_._wide_plan(probs(rows, shapes), cus=256)]
    assert pick(3120, layer)[0] in (4, 5, 6) and pick(24060, layer)[0] in (10, 11, 12, 13, 14)      # (measured: 4 and 12; the curve is flat +-1)
    assert pick(3120, enc) == [2] and pick(24060, enc) == [3] and pick(12657, enc) == [3]
    c3 = ((1536, 512), (512, 512), (2048, 512), (512, 2048)) * 12          # 12 layers of width 512: 48 problems, 384 tiles
    assert pick(24060, c3) == [2]
