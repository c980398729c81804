"""CPU restatement of the reference's beam-search decode (transformer/Beam.py, transformer/Decode.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity: ``Beam`` is **pinned** - tests/golden/beam_trellis.npz is the
trellis the reference's own Beam class produces (tools/make_beam_goldens.py imports /root/reference/transformer/Beam.py and
applies repair R5, the floor division of line 65) and tests/test_decode_cpu.py holds this class, the product's
transformer/Beam.py and st_beam_advance to it (indices bit-exact, scores <= 1e-6).  ``beam_search`` is **pinned** since round 6:
`Decode.__init__` calls an obsolete `Transformer(...)` signature and `prob_projection` is undefined, but `decode_batch` itself
runs once it is handed the (repaired, imported) reference model behind an adapter - tools/make_search_goldens.py executes the
reference's own method text and writes tests/golden/beam_search.npz (five utterances, beam 4, n_best 2: four finish after 6-7
tokens and leave the batch one by one - the compaction of Decode.py:112-165 -, one runs all 100 steps); this function reproduces it
exactly (identical hypotheses, scores to 1e-9: tests/test_decode_cpu.py::test_search_driver_golden_pins_the_oracle).  What it restates:

* ``Beam`` (Beam.py:13-116): scores start at 0, ``next_ys[0] = [BOS]*size``; ``advance(word_lk)`` takes
  ``[beam, V]`` log-probabilities, first step uses row 0 only (:48-51), top-k over the flattened beam x V array
  (:55-58), back-pointer ``id // V`` (:65, the repair of D12) and token ``id - prev*V`` (:67), done when the
  top-of-beam token is EOS (:70-72).  Hypotheses that emitted EOS below the top keep being extended (as in
  the reference); ``sort_scores`` / ``get_hypothesis`` as :75-116.
* ``beam_search`` (Decode.py:48-179): encode once, repeat per beam (:57-66), at most 100 steps (:75), feed the
  whole prefix each step and take the last position (:96-98), ``log_softmax`` as the undefined
  ``prob_projection``, advance every unfinished beam (:103-110), drop finished utterances (:112-165), return
  the ``n_best`` hypotheses / scores per utterance (:168-177).
"""
from typing import List, Tuple

import torch

from . import speech_transformer_oracle as orc

PAD, UNK, BOS, EOS = orc.PAD, orc.UNK, orc.BOS, orc.EOS


class Beam:
    def __init__(self, size: int):
        self.size = size
        self.done = False
        self.scores = torch.zeros(size, dtype=torch.float64)
        self.prev_ks: List[torch.Tensor] = []
        self.next_ys = [torch.full((size,), BOS, dtype=torch.long)]

    def advance(self, word_lk: torch.Tensor) -> bool:
        num_words = word_lk.size(1)
        beam_lk = word_lk + self.scores.unsqueeze(1) if self.prev_ks else word_lk[0]
        best_scores, best_ids = beam_lk.reshape(-1).topk(self.size, 0, True, True)
        self.scores = best_scores
        prev_k = best_ids // num_words
        self.prev_ks.append(prev_k)
        self.next_ys.append(best_ids - prev_k * num_words)
        if int(self.next_ys[-1][0]) == EOS:
            self.done = True
        return self.done

    def sort_scores(self):
        return torch.sort(self.scores, 0, True)

    def get_hypothesis(self, k: int) -> List[int]:
        hyp = []
        for j in range(len(self.prev_ks) - 1, -1, -1):
            hyp.append(int(self.next_ys[j + 1][k]))
            k = int(self.prev_ks[j][k])
        return hyp[::-1]

    def current_prefixes(self) -> torch.Tensor:
        """[beam, len] decoder input: BOS + the hypothesis of every beam slot (Beam.py:83-95; the slots are
        already in score order because top-k returns sorted scores)."""
        if len(self.next_ys) == 1:
            return self.next_ys[0].unsqueeze(1)
        return torch.tensor([[BOS] + self.get_hypothesis(k) for k in range(self.size)], dtype=torch.long)


def beam_search(p: orc.Params, x: torch.Tensor, in_len: torch.Tensor, n_head: int, beam_size: int, n_best: int = 1,
                max_steps: int = 100) -> Tuple[List[List[List[int]]], List[torch.Tensor]]:
    """-> (all_hyp[b][n] = token list, all_scores[b] = tensor[n_best])."""
    bsz = x.shape[0]
    t_max = int(in_len.max())
    enc, _ = orc.encoder(p, x[:, :t_max], in_len, n_head)
    beams = [Beam(beam_size) for _ in range(bsz)]
    for step in range(max_steps):
        active = [b for b in range(bsz) if not beams[b].done]
        if not active:
            break
        prefixes = torch.cat([beams[b].current_prefixes() for b in active], 0)                # [n*beam, step+1]
        idx = torch.tensor(active).repeat_interleave(beam_size)
        tgt_len = torch.full((prefixes.shape[0],), step + 1, dtype=torch.long)
        t_act = int(in_len[idx].max())          # the masks are built for the longest ACTIVE utterance (Decode.py:135-165)
        dec, _, _ = orc.decoder(p, prefixes, tgt_len, in_len[idx], enc[idx][:, :t_act], n_head)
        logits = torch.nn.functional.linear(dec[:, -1], p["tgt_word_proj.weight"])
        word_lk = torch.log_softmax(logits, -1).view(len(active), beam_size, -1)
        for i, b in enumerate(active):
            beams[b].advance(word_lk[i])
    all_hyp, all_scores = [], []
    for b in range(bsz):
        scores, order = beams[b].sort_scores()
        all_scores.append(scores[:n_best])
        all_hyp.append([beams[b].get_hypothesis(int(k)) for k in order[:n_best]])
    return all_hyp, all_scores


def score_hypothesis(p: orc.Params, x: torch.Tensor, in_len: torch.Tensor, n_head: int, hyp: List[int]) -> float:
    """Teacher-forced log-probability of one hypothesis for a single utterance (x [1, T, F])."""
    enc, _ = orc.encoder(p, x[:, :int(in_len.max())], in_len, n_head)
    prefix = torch.tensor([[BOS] + hyp[:-1]], dtype=torch.long)
    dec, _, _ = orc.decoder(p, prefix, torch.tensor([prefix.shape[1]]), in_len, enc, n_head)
    lp = torch.log_softmax(torch.nn.functional.linear(dec[0], p["tgt_word_proj.weight"]), -1)
    return float(sum(lp[t, tok] for t, tok in enumerate(hyp)))
