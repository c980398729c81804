"""Encoder / decoder layers (drop-in for reference transformer/Layers.py:8-44)."""
import torch.nn as nn

from transformer.Attention import MultiHeadAttention
from transformer.SubLayers import PositionwiseFeedForward


class EncoderLayer(nn.Module):
    """self-attention -> feed-forward."""

    def __init__(self, d_model, d_inner_hid, n_head, d_k, d_v, dropout=0.1):
        super(EncoderLayer, self).__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner_hid, dropout=dropout)

    def forward_rows(self, x, rows):
        return self.pos_ffn.forward_rows(self.slf_attn.forward_rows(x, None, rows, rows, False))

    def forward(self, inputs, slf_attn_mask=None):
        a, w = self.slf_attn(inputs, inputs, inputs, mask=slf_attn_mask)
        return self.pos_ffn(a), w


class DecoderLayer(nn.Module):
    """masked self-attention -> encoder-decoder attention -> feed-forward."""

    def __init__(self, d_model, d_inner_hid, n_head, d_k, d_v, dropout=0.1):
        super(DecoderLayer, self).__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.enc_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner_hid, dropout=dropout)

    def forward_rows(self, y, enc, t_rows, in_rows, kv_acc=None):
        s = self.slf_attn.forward_rows(y, None, t_rows, t_rows, True)
        c = self.enc_attn.forward_rows(s, enc, t_rows, in_rows, False, kv_acc=kv_acc)
        return self.pos_ffn.forward_rows(c)

    def forward(self, inputs, enc_output, slf_attn_mask=None, dec_enc_attn_mask=None):
        s, w1 = self.slf_attn(inputs, inputs, inputs, mask=slf_attn_mask)
        c, w2 = self.enc_attn(s, enc_output, enc_output, mask=dec_enc_attn_mask)
        return self.pos_ffn(c), (w1, w2)
