"""Host logic of st_amd/chains.py (no GPU): the block tables of the fragment streams, the chain plans and their limits."""
import torch

from st_amd import chains
from tests._emul import emulated_kernels

BF16 = torch.bfloat16


def _w(n, k):
    return torch.zeros(n, k, dtype=BF16)


def test_block_tables_spell_the_chain_in_consumption_order():
    with emulated_kernels():
        wo, w1, w2, wqkv = _w(256, 256), _w(1024, 256), _w(256, 1024), _w(768, 256)
        cs = chains.ChainSet("cpu")
        f = cs.add(chains.blocks_of(wo) + chains.ffn_blocks(w1, w2) + chains.blocks_of(wqkv))
        b = cs.add(chains.t_blocks(chains.blocks_of(wqkv)) + chains.ffn_blocks_bwd(w1, w2) + chains.t_blocks(chains.blocks_of(wo)))
        cs.finalize()
        t = cs.table.tolist()
        assert len(t) == 24 and cs.chain(f).n_blocks == 12 and cs.chain(b).n_blocks == 12
        depth, wave_frags = cs.depth, 12 * 16 + cs.depth
        # forward chain: Wo | (W1 rows c*256.., W2 columns c*256..) x 4 | Wqkv rows u*256..; not transposed; block i at fragment 16 i
        exp = [(wo, 0, 0)] + sum(([(w1, c, 0), (w2, 0, c)] for c in range(0, 1024, 256)), []) + [(wqkv, r, 0) for r in (0, 256, 512)]
        for i, (w, n0, k0) in enumerate(exp):
            src, ld, frag, dst = t[i]
            assert src == w.data_ptr() + 2 * (n0 * w.stride(0) + k0) and ld == w.stride(0) and frag == (16 * i) | (wave_frags << 32)
            assert dst == cs.buf.data_ptr()
        # backward chain: the same blocks transposed, reverse order of the sublayers; stored right behind the forward chain
        exp_b = [(wqkv, r, 0) for r in (0, 256, 512)] + sum(([(w2, 0, c), (w1, c, 0)] for c in range(0, 1024, 256)), []) + [(wo, 0, 0)]
        for i, (w, n0, k0) in enumerate(exp_b):
            src, ld, frag, dst = t[12 + i]
            assert src == w.data_ptr() + 2 * (n0 * w.stride(0) + k0) and ld == w.stride(0) | (1 << 32)
            assert dst == cs.buf.data_ptr() + 2 * 8 * wave_frags * 512
        assert cs.buf.numel() == 2 * 8 * wave_frags * 512 and cs.chain(f, True).next_blocks == 12 and cs.chain(b, True).next_blocks == 0
        assert depth == 16


def test_plans_decline_what_the_kernels_do_not_serve():
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd.arena import arena_of

    def model(d, h, dff):
        return M.Transformer(U.AttrDict(dict(feature_dim=80, max_inputs_length=64, max_target_length=16, num_enc_layer=1,
                                             num_dec_layer=2, n_heads=h, d_k=d // h, d_v=d // h, d_model=d, d_inner_hid=dff,
                                             dropout=0.0, vocab_size=30)))

    with emulated_kernels():
        m = model(256, 4, 512)
        a = arena_of(m)
        ec, dc = m.encoder.row_chains(a), m.decoder.row_chains(a)
        assert ec is not None and dc is not None and ec.use_bwd and dc.use_bwd
        assert [c.n_blocks for c in dc.f1] == [2, 2] and [c.n_blocks for c in dc.f2] == [1 + 4 + 3, 1 + 4]
        assert [c.n_blocks for c in dc.bwd2] == [3 + 4 + 1, 4 + 1] and [c.n_blocks for c in dc.bwd1] == [2, 2]
        assert len(chains.ChainHub.of(a).sets) == 4 and chains.ChainHub.of(a).table.shape[0] == sum(
            s.table.shape[0] for s in chains.ChainHub.of(a).sets)
        m.decoder.use_row_chains = False
        assert m.decoder.row_chains(a) is None
        m8 = model(256, 8, 512)                 # d_k 32: forward chains yes, backward chains no (the delta epilogue's heads are 64 wide)
        assert m8.encoder.row_chains(arena_of(m8)).use_bwd is False
        for bad in (model(128, 4, 256), model(256, 4, 384)):
            ab = arena_of(bad)
            assert bad.encoder.row_chains(ab) is None and bad.decoder.row_chains(ab) is None
        m5 = model(512, 8, 1024)                # config 3's width: forward and backward chains for the encoder (st_row_chain512[_bwd])
        a5 = arena_of(m5)
        e5 = m5.encoder.row_chains(a5)
        assert e5 is not None and not e5.use_bwd and [c.n_blocks for c in e5.e] == [4 + 16] and [c.n_blocks for c in e5.bwd] == [16 + 4]
        assert m5.decoder.row_chains(a5) is None
