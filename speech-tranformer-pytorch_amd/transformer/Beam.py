"""Per-utterance beam state for transformer/Decode.py - API-compatible with the reference's transformer/Beam.py.

What is kept from the reference is the public surface only (constructor, ``advance``, ``sort_scores``,
``get_hypothesis``, ``get_current_state`` / ``get_current_origin`` / ``get_tentative_hypothesis``,
``get_the_best_score_and_idx`` and the attributes ``size, done, scores, all_scores, prev_ks, next_ys``) and the
search rule it encodes:

* step 0 expands a single BOS row (all slots are identical), later steps add each slot's running score to its
  ``[beam, vocab]`` log-probabilities;
* the ``size`` best entries of the flattened ``beam x vocab`` table become the new slots: back-pointer =
  ``index // vocab`` (the reference divides with ``/`` - float indices on current PyTorch, its decode cannot run;
  SURVEY D12), token = ``index % vocab``;
* the beam is finished as soon as its best slot emits EOS; slots that emitted EOS further down keep being
  extended, exactly as in the reference.

The trellis lives on the device the scores live on.  ``advance`` needs one host read per call (the EOS test);
``advance_batch`` does the same update for many utterances with one top-k launch and one host read.
"""
import torch

import transformer.Constants as Constants


class Beam(object):
    def __init__(self, size, device):
        self.size, self.device = size, device
        self.done = False
        self.scores = torch.zeros(size, dtype=torch.float32, device=device)       # running log-probability per slot
        self.all_scores = []                                                        # score vector before every step
        self.prev_ks = []                                                           # back-pointers, one tensor per step
        self.next_ys = [torch.full((size,), Constants.BOS, dtype=torch.long, device=device)]   # tokens per step

    # ---- state update ------------------------------------------------------------------------------
    def _commit(self, best_scores, origin, token, finished):
        self.all_scores.append(self.scores)
        self.scores = best_scores
        self.prev_ks.append(origin)
        self.next_ys.append(token)
        if finished:
            self.done = True
            self.all_scores.append(self.scores)

    def advance(self, word_lk):
        """word_lk [beam, vocab] log-probabilities of the next token for every slot -> True once finished."""
        vocab = word_lk.size(1)
        table = word_lk[0] if not self.prev_ks else word_lk + self.scores.unsqueeze(1)
        best_scores, best_flat = table.reshape(-1).topk(self.size, 0, True, True)
        origin = best_flat // vocab
        token = best_flat - origin * vocab
        self._commit(best_scores, origin, token, token[0].item() == Constants.EOS)
        return self.done

    @staticmethod
    def advance_batch(beams, word_lk):
        """The same update for ``len(beams)`` utterances that are at the same step: word_lk [n, beam, vocab].
        One top-k launch and one device->host read in total; returns each beam's ``done`` flag."""
        n, size, vocab = word_lk.shape
        if beams[0].prev_ks:
            table = (word_lk + torch.stack([b.scores for b in beams]).unsqueeze(2)).reshape(n, -1)
        else:
            table = word_lk[:, 0]
        best_scores, best_flat = table.topk(size, 1, True, True)
        origin = best_flat // vocab                      # back-pointers and tokens of every utterance in two launches
        token = best_flat - origin * vocab
        finished = (token[:, 0] == Constants.EOS).tolist()
        for i, (b, sc, og, tk) in enumerate(zip(beams, best_scores.unbind(0), origin.unbind(0), token.unbind(0))):
            b._commit(sc, og, tk, finished[i])
        return [b.done for b in beams]

    # ---- read-out ----------------------------------------------------------------------------------
    def sort_scores(self):
        """(scores, slot indices) in decreasing score order."""
        return torch.sort(self.scores, 0, True)

    def get_the_best_score_and_idx(self):
        """Element [1] of the sorted scores / indices - the reference returns the runner-up here (its Beam.py:81)."""
        ordered, slots = self.sort_scores()
        return ordered[1], slots[1]

    def get_current_origin(self):
        """Back-pointers of the latest step: new slot j extends old slot get_current_origin()[j]."""
        return self.prev_ks[-1]

    def get_hypothesis(self, k):
        """Token list of slot k, rebuilt through the back-pointers (one device->host copy of the trellis)."""
        if not self.prev_ks:
            return []
        back = torch.stack(self.prev_ks).tolist()
        toks = torch.stack(self.next_ys[1:]).tolist()
        slot, out = int(k), []
        for step in reversed(range(len(back))):
            out.append(toks[step][slot])
            slot = back[step][slot]
        out.reverse()
        return out

    def get_tentative_hypothesis(self):
        """[beam, len] decoder input: BOS followed by every slot's tokens, best slot first."""
        if len(self.next_ys) == 1:
            return self.next_ys[0].unsqueeze(1)
        _, slots = self.sort_scores()
        rows = [[Constants.BOS] + self.get_hypothesis(s) for s in slots.tolist()]
        return torch.tensor(rows, dtype=torch.long, device=self.device)

    get_current_state = get_tentative_hypothesis
