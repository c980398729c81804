"""Feature front-end oracle (oracle/feature_oracle.py) against the import-generated golden
(tests/golden/features.npz: the reference's AudioDateset.cmvn / concat_frame / subsampling as written)."""
import os

import numpy as np

from oracle import feature_oracle as fo


def test_feature_oracle_matches_reference_golden(golden_dir):
    fx = dict(np.load(os.path.join(golden_dir, "features.npz")))
    for ci, (left, right, rate) in enumerate(fx["cases"]):
        feats, stats = fx["c%d/feats" % ci], fx["c%d/stats" % ci]
        assert np.allclose(fo.cmvn(feats, stats), fx["c%d/cmvn" % ci], rtol=1e-12, atol=0)
        stacked = fo.concat_frame(fx["c%d/cmvn" % ci].astype(np.float32), int(left), int(right))
        assert np.array_equal(stacked, fx["c%d/stacked" % ci])
        assert np.array_equal(fo.subsampling(stacked, int(rate)), fx["c%d/out" % ci])
        assert np.array_equal(fo.front_end(feats, stats, int(left), int(right), int(rate)), fx["c%d/out" % ci])


def _case(left, right, rate, with_stats, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    B, T, F = 3, 41, 8
    lens = torch.tensor([41, 17, 30])
    x = torch.randn(B, T, F, generator=g) * 2 + 0.5
    for b in range(B):
        x[b, lens[b]:] = 0
    stats = None
    if with_stats:
        cnt = torch.tensor([400.0, 500.0, 600.0])
        mean, var = torch.randn(B, F, generator=g) * 0.3, torch.rand(B, F, generator=g) + 0.5
        stats = torch.zeros(B, 2, F + 1)
        stats[:, 0, :-1], stats[:, 0, -1] = mean * cnt[:, None], cnt
        stats[:, 1, :-1] = (var + mean ** 2) * cnt[:, None]
    return x, lens, stats


def run_stack_frames(device):
    """st_amd.features.stack_frames (the st_feat_stack kernel, or its emulation on CPU) against the oracle, per
    utterance, for the context / frame-rate combinations of the golden."""
    import torch
    from st_amd.features import stack_frames
    for ci, (left, right, rate) in enumerate([(3, 0, 10), (3, 0, 30), (2, 2, 10), (2, 1, 20), (0, 0, 10)]):
        for with_stats in (False, True):
            x, lens, stats = _case(left, right, rate, with_stats, 10 + ci)
            rows_mat, rows = stack_frames(x.to(device), lens, left, right, rate,
                                          None if stats is None else stats.to(device))
            o = 0
            for b in range(x.shape[0]):
                want = fo.front_end(x[b, :lens[b]].numpy().astype(np.float64) if with_stats else x[b, :lens[b]].numpy(),
                                    None if stats is None else stats[b].numpy().astype(np.float64), left, right, rate)
                got = rows_mat[o:o + want.shape[0]].float().cpu().numpy()
                assert int(rows.lens_host[b]) == want.shape[0]
                err = np.abs(got - want).max()
                assert err <= 2 ** -8 * max(1.0, np.abs(want).max()), (left, right, rate, with_stats, b, err)   # bf16 storage
                o += want.shape[0]
            assert o == rows.total


def test_stack_frames_emulated():
    from tests._emul import emulated_kernels
    with emulated_kernels():
        run_stack_frames("cpu")
