"""Beam-search decode (transformer/Beam.py, transformer/Decode.py) against the oracle restatement
(oracle/beam_oracle.py; parity unpinned - the reference's decode cannot run, see its header).

Search is discontinuous in the scores, so parity is stated in a way that survives bf16 noise: every hypothesis
the HIP path returns must (a) be scored by the fp64 oracle (teacher forcing) at the score the HIP path reported,
and (b) be as good as the oracle's own best hypothesis up to that noise; on this model the token sequences also
coincide exactly for the best hypothesis."""
import pytest
import torch

import oracle as orc
from oracle import beam_oracle as bo
from tests._emul import emulated_kernels

H = 4


def _params(eos_boost=0.0, d=128, dff=256, n_enc=2, n_dec=2):
    p = orc.xavier_init_(orc.make_params(80, 30, d, dff, n_enc, n_dec, 100, 40, dtype=torch.float64), seed=1)
    p["tgt_word_proj.weight"] = p["tgt_word_proj.weight"] * 12.0        # peaky distributions: robust rankings
    if eos_boost:
        # make EOS competitive late in the sequence: its output row follows the positional encoding of step ~6
        pe = p["decoder.position_enc.pe"].reshape(-1, d)
        p["tgt_word_proj.weight"][bo.EOS] = eos_boost * (pe[6] - pe[1])
    return p


def _model(p, device, d=128, dff=256, n_enc=2, n_dec=2):
    import transformer.Models as M
    import transformer.Utils as U
    cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=40, num_enc_layer=n_enc,
                          num_dec_layer=n_dec, n_heads=H, d_k=d // H, d_v=d // H, d_model=d, d_inner_hid=dff, dropout=0.1,
                          vocab_size=30))
    m = M.Transformer(cfg)
    m.load_state_dict({k: v.float() for k, v in p.items()})
    return m.eval().to(device)


def test_beam_class_matches_oracle_beam():
    from transformer.Beam import Beam
    g = torch.Generator().manual_seed(0)
    for trial in range(20):
        size, V = 4, 11
        a, b = Beam(size, "cpu"), bo.Beam(size)
        for step in range(12):
            lk = torch.log_softmax(torch.randn(size, V, generator=g) * 3, -1)
            da, db = a.advance(lk), b.advance(lk.double())
            assert da == db
            assert torch.equal(a.next_ys[-1], b.next_ys[-1]) and torch.equal(a.prev_ks[-1], b.prev_ks[-1])
            assert torch.allclose(a.scores.double(), b.scores, atol=1e-5)
            if da:
                break
        for k in range(size):
            assert a.get_hypothesis(k) == b.get_hypothesis(k)
        assert a.get_current_state().tolist() == b.current_prefixes().tolist()


def test_beam_advance_batch_equals_per_beam_advance():
    from transformer.Beam import Beam
    g = torch.Generator().manual_seed(1)
    size, V, n = 3, 9, 4
    one, many = [Beam(size, "cpu") for _ in range(n)], [Beam(size, "cpu") for _ in range(n)]
    for step in range(6):
        lk = torch.log_softmax(torch.randn(n, size, V, generator=g) * 3, -1)
        d1 = [b.advance(lk[i]) for i, b in enumerate(one)]
        d2 = Beam.advance_batch(many, lk)
        assert d1 == d2
        for a, b in zip(one, many):
            assert torch.equal(a.scores, b.scores) and torch.equal(a.prev_ks[-1], b.prev_ks[-1])
            assert torch.equal(a.next_ys[-1], b.next_ys[-1]) and a.done == b.done
            assert [a.get_hypothesis(k) for k in range(size)] == [b.get_hypothesis(k) for k in range(size)]


def run_decode(device, eos_boost, max_steps, beam=4, shape=None, use_graph=None, proj_scale=12.0):
    """shape = (d_model, d_ff, n_enc, n_dec); BASELINE config 5 is beam 10 on (256, 1024, 6, 6)."""
    from transformer.Decode import Decode
    from transformer.Utils import AttrDict
    shape = shape or (128, 256, 2, 2)
    p = _params(eos_boost, *shape)
    if proj_scale != 12.0:
        p["tgt_word_proj.weight"] = p["tgt_word_proj.weight"] * (proj_scale / 12.0)
    batch = orc.synthetic_batch(5, 80, 10, 80, 30, seed=2, t_min=30, l_min=5)
    x, in_len = batch["x"], batch["in_len"]
    dec = Decode(AttrDict(dict(beam_size=beam, n_best=2, max_steps=max_steps, use_graph=use_graph)), device,
                 model=_model(p, device, *shape))
    hyps, scores = dec.decode_batch((x, in_len))
    ref_h, ref_s = bo.beam_search(p, x.double(), in_len, H, beam_size=beam, n_best=2, max_steps=max_steps)
    # bf16 logits of magnitude ~10 carry ~0.03 of absolute noise per step on the 2+2-layer model; a few steps dominate a
    # score.  The 6+6-layer, d_model 256 model (config 5) is three times as deep and its x12 output projection turns the
    # same relative noise into up to 0.4 on a 10-step score (tools/dev/decode_dbg.py: the error scales with the
    # projection, 0.04 at x1, 0.14 at x4, 0.40 at x12, and the teacher-forced TRAINING kernels show the same spread), so
    # the deep case runs at x4 with a wider bound.
    deep = shape[2] + shape[3] > 4
    # deep: up to 0.55 (5.6 %) measured, always in the SAME direction (reported score above the fp64 score): the search keeps
    # the hypotheses whose bf16 noise was favourable - the winner's curse of an arg-max over noisy scores on a random-weight
    # model with many near-ties (the logits themselves carry no bias: least-squares gain 0.9995, rel-L2 7e-3,
    # tools/dev/logit_scale.py; the teacher-forced TRAINING kernels score the same hypotheses within 0.04 of the decode path)
    tol = 0.6 if deep else 0.12
    lengths = set()
    for b in range(x.shape[0]):
        assert len(hyps[b]) == 2 and len(scores[b]) == 2
        for n in range(2):
            got = float(scores[b][n])
            truth = bo.score_hypothesis(p, x[b:b + 1].double(), in_len[b:b + 1], H, hyps[b][n])
            assert abs(got - truth) <= max(tol, (8e-2 if deep else 5e-2) * abs(truth)), (b, n, got, truth)      # (a)
        assert float(scores[b][0]) >= float(ref_s[b][0]) - max(tol, 5e-2 * abs(float(ref_s[b][0])))   # (b)
        if not deep or len(ref_s[b]) < 2 or float(ref_s[b][0]) - float(ref_s[b][1]) > 2 * tol:
            assert hyps[b][0] == ref_h[b][0], (b, hyps[b][0], ref_h[b][0])       # a near-tie may legitimately flip
        lengths.add(len(hyps[b][0]))
    return lengths


def test_decode_runs_to_the_step_limit():
    with emulated_kernels():
        assert run_decode("cpu", 0.0, 12) == {12}


def test_decode_with_early_finishers():
    """EOS reachable: utterances finish at different steps, leave the batch, and the rest keep their caches."""
    with emulated_kernels():
        lengths = run_decode("cpu", 3.0, 16)
    assert len(lengths) > 1 or min(lengths) < 16
