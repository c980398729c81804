"""Host-side helpers the reference's ``transformer/Utils.py`` provides to the hot path.

Only what the training step touches is mirrored: the config AttrDict, the two
mask builders (kept for API parity - the HIP path consumes *lengths*, see
:class:`LengthMask`), parameter initialisation / counting and the Noam rate.
"""
import logging

import torch


class AttrDict(dict):
    """dict with attribute access; a missing key reads as ``None``
    (reference Utils.py:9-22 - the silent-None behaviour is part of the contract)."""

    def __getattr__(self, item):
        if item not in self:
            return None
        val = self[item]
        if type(val) is dict:
            val = self[item] = AttrDict(val)
        return val


class LengthMask(object):
    """What the reference's dense masks encode, without materialising them.

    ``padding_info_mask`` marks key j of utterance b iff ``j >= k_len[b]``
    (Utils.py:41-57); ``feature_info_mask`` marks ``j > i`` (Utils.py:60-70).  The
    HIP attention kernels take the lengths and a causal flag directly.
    """

    def __init__(self, q_len, k_len, causal=False):
        self.q_len, self.k_len, self.causal = q_len, k_len, bool(causal)

    def dense(self, device=None):
        m = padding_info_mask(self.q_len, self.k_len)
        if self.causal:
            m = m | feature_info_mask(self.q_len)
        return m.to(device) if device is not None else m


def padding_info_mask(seq_q_length, seq_k_length):
    """bool [B, max(q_len), max(k_len)], True where the key is padding (Utils.py:41-57)."""
    assert seq_q_length.dim() == 1 and seq_k_length.dim() == 1
    len_q, len_k = int(seq_q_length.max()), int(seq_k_length.max())
    cols = torch.arange(len_k, device=seq_k_length.device)
    mask = cols.view(1, 1, -1) >= seq_k_length.view(-1, 1, 1)
    return mask.expand(seq_k_length.size(0), len_q, len_k)


def feature_info_mask(seq_length):
    """bool [B, L, L], True strictly above the diagonal (Utils.py:60-70)."""
    assert seq_length.dim() == 1
    n = int(seq_length.max())
    tri = torch.ones(n, n, dtype=torch.bool, device=seq_length.device).triu(1)
    return tri.unsqueeze(0).expand(seq_length.size(0), n, n)


def lengths_from_mask(mask):
    """Recover (k_len[B], causal) from a dense [B, Lq, Lk] mask built by the two
    functions above; None if the mask is not of that family (MultiHeadAttention then
    takes st_amd.functional.DenseMhaFn).  A device sync: for callers that hand
    MultiHeadAttention a dense mask instead of a LengthMask."""
    mask = mask.bool()
    B, Lq, Lk = mask.shape      # (callers with a broadcastable mask - [B, 1, Lk] - expand it first: MultiHeadAttention.forward does)
    k_len = (~mask[:, -1, :]).sum(-1)
    pad = torch.arange(Lk, device=mask.device).view(1, 1, -1) >= k_len.view(-1, 1, 1)
    if torch.equal(mask, pad.expand(B, Lq, Lk)):
        return k_len, False
    if Lq == Lk:
        tri = torch.ones(Lq, Lk, dtype=torch.bool, device=mask.device).triu(1)
        if torch.equal(mask, pad | tri):
            return k_len, True
    return None          # a foreign mask: MultiHeadAttention takes its general (slow) path


def learn_rate(d_model, n_warmup_steps, current_step):
    """Noam schedule (Utils.py:73-77, Optim.py:36-45)."""
    return d_model ** -0.5 * min(current_step ** -0.5, n_warmup_steps ** -1.5 * current_step)


def count_parameters(model):
    """(total, encoder, decoder) element counts (Utils.py:89-98)."""
    enc = sum(p.nelement() for n, p in model.named_parameters() if 'encoder' in n)
    dec = sum(p.nelement() for n, p in model.named_parameters() if 'encoder' not in n and 'decoder' in n)
    return sum(p.nelement() for p in model.parameters()), enc, dec


def init_parameters(model):
    """xavier_normal on every parameter with >= 2 dims (Utils.py:101-104) - this
    also overwrites the zeroed padding_idx row of the embedding, as in the reference."""
    for _, param in model.named_parameters():
        if param.dim() >= 2:
            torch.nn.init.xavier_normal_(param)


def init_logger(log_file=None):
    """Root logger to console (+ file) (Utils.py:25-38)."""
    fmt = logging.Formatter("[%(asctime)s %(levelname)s] %(message)s")
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    handlers = [logging.StreamHandler()]
    if log_file:
        handlers.append(logging.FileHandler(log_file))
    for h in handlers:
        h.setFormatter(fmt)
    logger.handlers = handlers
    return logger
