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
        wq, wk = attn_work(qr, kr, causal)
        assert wq.dtype == torch.int32 and wk.dtype == torch.int32
        want_q = {(i, t) for i in range(len(a)) for t in range((int(a[i]) + 127) // 128)}
        want_k = {(i, t) for i in range(len(b)) for t in range((int(b[i]) + 127) // 128)}
        got_q, got_k = _decode(wq), _decode(wk)
        assert len(got_q) == len(want_q) and set(got_q) == want_q
        assert len(got_k) == len(want_k) and set(got_k) == want_k

        def cost_q(i, t):
            seen = min(int(b[i]), (t + 1) * 128) if causal else int(b[i])
            return (seen + 63) // 64

        def cost_k(i, t):
            q_begin = (t * 128 // 64) * 64 if causal else 0
            return (int(a[i]) - q_begin + 63) // 64

        cq = [cost_q(*it) for it in got_q]
        ck = [cost_k(*it) for it in got_k]
        assert cq == sorted(cq, reverse=True) and ck == sorted(ck, reverse=True)
        assert min(cq) >= 1 and min(ck) >= 1
        # cached on the query layout
        assert attn_work(qr, kr, causal)[0] is wq
