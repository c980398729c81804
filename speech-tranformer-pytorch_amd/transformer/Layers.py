"""Encoder / decoder layers (drop-in for reference transformer/Layers.py:8-44)."""
import torch
import torch.nn as nn

from st_amd.functional import LnLink

from transformer.Attention import MultiHeadAttention
from transformer.SubLayers import PositionwiseFeedForward


def _links(n, pre=None):
    """LnLinks only when a backward pass will follow - and not when the stack's backward runs as row chains
    (st_amd.chains.ChainBackward does the LayerNorm-backward hand-over itself)."""
    on = torch.is_grad_enabled() and not (pre is not None and pre[0].bwd is not None)
    return [LnLink() if on else None for _ in range(n)]


class EncoderLayer(nn.Module):
    """self-attention -> feed-forward."""

    def __init__(self, d_model, d_inner_hid, n_head, d_k, d_v, dropout=0.1):
        super(EncoderLayer, self).__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner_hid, dropout=dropout)

    def forward_rows(self, x, rows, up=None, pre=None):
        """-> (output rows, LnLink the next layer may pass back as ``up``).  The links let each sublayer's backward
        run the previous sublayer's LayerNorm backward inside its last GEMM; they require that nothing but the next
        sublayer consumes the intermediate tensors, which holds inside the stacks.
        pre: this layer's (self-attention, feed-forward) st_amd.chains.SubPre when the forward values come from the
        fused launches (st_amd.chains.EncoderChains.forward)."""
        l1, l2 = _links(2, pre)
        pa, pf = pre if pre is not None else (None, None)
        a = self.slf_attn.forward_rows(x, None, rows, rows, False, up=up, down=l1, pre=pa)
        return self.pos_ffn.forward_rows(a, up=l1, down=l2, pre=pf), l2

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

    def forward_rows(self, y, enc, t_rows, in_rows, kv_acc=None, up=None, pre=None):
        """-> (output rows, LnLink for the next layer); see EncoderLayer.forward_rows.
        pre: this layer's (self-attention, encoder-decoder attention, feed-forward) st_amd.chains.SubPre when the
        forward values come from the fused decoder launches (st_amd.chains.DecoderChains.forward)."""
        l1, l2, l3 = _links(3, pre)
        pa, pb, pf = pre if pre is not None else (None, None, None)
        s = self.slf_attn.forward_rows(y, None, t_rows, t_rows, True, up=up, down=l1, pre=pa)
        c = self.enc_attn.forward_rows(s, enc, t_rows, in_rows, False, kv_acc=kv_acc, up=l1, down=l2, pre=pb)
        return self.pos_ffn.forward_rows(c, up=l2, down=l3, pre=pf), l3

    def forward(self, inputs, enc_output, slf_attn_mask=None, dec_enc_attn_mask=None):
        s, w1 = self.slf_attn(inputs, inputs, inputs, mask=slf_attn_mask)
        c, w2 = self.enc_attn(s, enc_output, enc_output, mask=dec_enc_attn_mask)
        return self.pos_ffn(c), (w1, w2)
