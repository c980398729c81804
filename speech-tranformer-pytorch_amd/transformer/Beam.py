"""Beam bookkeeping: drop-in for the reference's transformer/Beam.py (OpenNMT-style beam).

Same class, same methods, same semantics - with the one repair the reference needs to run at all: the
back-pointer is ``id // num_words`` (Beam.py:65 uses true division, which yields float indices on current
PyTorch and fails at the first ``get_hypothesis``).  Everything stays on the device the scores live on; the only
host read per step is the "top of beam is EOS" test (Beam.py:70), as in the reference."""
import torch

import transformer.Constants as Constants


class Beam(object):
    ''' Store the necessary info for beam search. '''

    def __init__(self, size, device):
        self.size = size
        self.done = False
        self.device = device
        self.scores = torch.zeros(size, dtype=torch.float32, device=device)      # Beam.py:24
        self.all_scores = []
        self.prev_ks = []                                                          # back-pointers per step
        self.next_ys = [torch.full((size,), Constants.BOS, dtype=torch.long, device=device)]   # Beam.py:31-33

    def get_current_state(self):
        "Get the outputs for the current timestep."
        return self.get_tentative_hypothesis()

    def get_current_origin(self):
        "Get the backpointers for the current timestep."
        return self.prev_ks[-1]

    def advance(self, word_lk):
        "Update the status and check for finished or not.  word_lk: [beam, num_words] log-probabilities."
        num_words = word_lk.size(1)
        if len(self.prev_ks) > 0:
            beam_lk = word_lk + self.scores.unsqueeze(1).expand_as(word_lk)       # Beam.py:48-49
        else:
            beam_lk = word_lk[0]                                                   # all beams are BOS: use one row
        flat_beam_lk = beam_lk.reshape(-1)
        best_scores, best_scores_id = flat_beam_lk.topk(self.size, 0, True, True)
        self.all_scores.append(self.scores)
        self.scores = best_scores
        prev_k = best_scores_id // num_words                                       # Beam.py:65, repaired (D12)
        self.prev_ks.append(prev_k)
        self.next_ys.append(best_scores_id - prev_k * num_words)
        if self.next_ys[-1][0].item() == Constants.EOS:                            # Beam.py:70: top-of-beam is EOS
            self.done = True
            self.all_scores.append(self.scores)
        return self.done

    @staticmethod
    def advance_batch(beams, word_lk):
        """``advance`` for several beams that are at the same step, with ONE top-k launch and ONE host read (the
        per-beam version costs a launch chain and a sync each).  word_lk: [len(beams), beam, num_words].
        Returns the list of ``done`` flags; every beam ends up exactly as after ``beams[i].advance(word_lk[i])``."""
        n, size, num_words = word_lk.shape
        if len(beams[0].prev_ks) > 0:
            scores = torch.stack([b.scores for b in beams])
            beam_lk = (word_lk + scores.unsqueeze(2)).reshape(n, -1)
        else:
            beam_lk = word_lk[:, 0]
        best_scores, best_ids = beam_lk.topk(size, 1, True, True)
        prev_k = best_ids // num_words
        next_y = best_ids - prev_k * num_words
        done = (next_y[:, 0] == Constants.EOS).tolist()                           # the one synchronisation
        for i, b in enumerate(beams):
            b.all_scores.append(b.scores)
            b.scores = best_scores[i]
            b.prev_ks.append(prev_k[i])
            b.next_ys.append(next_y[i])
            if done[i]:
                b.done = True
                b.all_scores.append(b.scores)
        return [b.done for b in beams]          # sticky, as advance() returns self.done

    def sort_scores(self):
        "Sort the scores."
        return torch.sort(self.scores, 0, True)

    def get_the_best_score_and_idx(self):
        "Get the score of the best in the beam (the reference returns element [1], Beam.py:81; kept)."
        scores, ids = self.sort_scores()
        return scores[1], ids[1]

    def get_tentative_hypothesis(self):
        "Get the decoded sequence for the current timestep: [beam, len] with BOS in front."
        if len(self.next_ys) == 1:
            return self.next_ys[0].unsqueeze(1)
        _, keys = self.sort_scores()
        hyps = [[Constants.BOS] + self.get_hypothesis(k) for k in keys.tolist()]
        return torch.tensor(hyps, dtype=torch.long, device=self.device)

    def get_hypothesis(self, k):
        "Walk the back-pointers to rebuild hypothesis k (a list of token ids)."
        k = int(k)
        if not self.prev_ks:
            return []
        # one device->host copy of the whole trellis instead of two reads per step
        prev = torch.stack(self.prev_ks).tolist()
        ys = torch.stack(self.next_ys[1:]).tolist()
        hyp = []
        for j in range(len(prev) - 1, -1, -1):
            hyp.append(ys[j][k])
            k = prev[j][k]
        return hyp[::-1]
