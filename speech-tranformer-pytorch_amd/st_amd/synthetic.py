"""Synthetic 80-d fbank batches - the stand-in for the
reference's Kaldi ``AudioDateset`` (Dataset.py:34-51) with the batch contract of its
synthetic loader (tests/random_character_loader.py:78-104): zero-padded features,
``[BOS-side]`` targets, ``ground_truth = targets shifted left`` with a PAD tail."""
import torch

PAD = 0


def make_batch(bsz, t_max, l_max, feat, vocab, seed=0, t_min=None, l_min=None):
    """-> (inputs [B,T,F] fp32, targets [B,L] int64, input_lengths [B], target_lengths [B],
    ground_truth [B,L]) - the 5-tuple order train.py:25 unpacks.  Utterance 0 has the
    maximum lengths so the batch trims to (t_max, l_max)."""
    t_min = t_max // 2 if t_min is None else t_min
    l_min = l_max // 2 if l_min is None else l_min
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(bsz, t_max, feat, generator=g)
    in_len = torch.randint(t_min, t_max + 1, (bsz,), generator=g)
    in_len[0] = t_max
    tgt_len = torch.randint(l_min, l_max + 1, (bsz,), generator=g)
    tgt_len[0] = l_max
    tokens = torch.randint(4, vocab, (bsz, l_max), generator=g)
    x = x * (torch.arange(t_max).view(1, -1, 1) < in_len.view(-1, 1, 1))
    tokens = tokens * (torch.arange(l_max).view(1, -1) < tgt_len.view(-1, 1))
    gt = torch.roll(tokens, -1, dims=1)
    gt[:, -1] = PAD
    return x, tokens, in_len, tgt_len, gt
