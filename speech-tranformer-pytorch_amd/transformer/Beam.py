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
        hyp = []
        for j in range(len(self.prev_ks) - 1, -1, -1):
            hyp.append(int(self.next_ys[j + 1][k]))
            k = int(self.prev_ks[j][k])
        return hyp[::-1]
