// The backward row chain at d_model = 512 (BASELINE config 3's encoder, st_rowchain_pipe512.cuh's mirror image) - included by
// st_rowchain.hip behind ChainBwdArgs:
//
//   HEAD  dy = sum_u dP[:, 256 u ..] Wp_u + G  (Wp [1536, 512]: 12 blocks; head_blocks = 0: dy = G), ds_a = LayerNorm-backward(dy)
//   FFN   dH_c = mask(ds_a W2[:, c]) (2 blocks per chunk), dy = sum_c dH_c W1[c, :] (2 blocks per chunk) + ds_a, ds_b = LayerNorm-backward(dy)
//   TAIL  dctx = ds_b Wo (4 blocks), delta[head][row] = sum over the head's 64 columns of dctx (O + Ores)       (8 heads)
//
// 64-row workgroups, two [64][512] tiles.  The LayerNorm backward writes dx OVER its xhat tile (the column sums that need xhat run
// in front of that), the hidden-gradient chunks ping-pong in the column halves of the tile that is free, O and Ores pass through
// one tile one after the other.  Stream positions (all blocks read transposed; st_amd.chains.encoder512_blocks_bwd):
//   HEAD (h, u) -> 6 h + u  |  FFN nbh + 4 c + {B1 j = 0, 1; B2 h = 0, 1}  |  TAIL b_tail + 2 h + j
// (h: output-column half of dy / dctx, u: 256-column block of dP, j: input-column half of ds, c: hidden chunk).
#pragma once

namespace {

// column sums of a [64][512] tile pair for the LayerNorm backward: thread = (column pair, half of the rows); the upper half hands
// its sums over through `xch`, ONE atomic per column and quantity
template <bool DROP, bool WITH_X>
__device__ __forceinline__ void colsum512(const Ctx<2>& c, const bf16* t_v, const bf16* t_x, const Drop& d, float* xch, float* out_v,
                                          float* out_vx) {
  const int col = (c.tid & 255) * 2, half = c.tid >> 8;
  float sv0 = 0.f, sv1 = 0.f, sx0 = 0.f, sx1 = 0.f;
  for (int i = 0; i < 32; ++i) {
    const int row = half * 32 + i;
    const bf16x2 v2 = *reinterpret_cast<const bf16x2*>(t_v + row * PT5 + col);
    float v0 = (float)v2[0], v1 = (float)v2[1];
    if (DROP && d.on()) {
      const uint32_t bits = d.bits(drop_counter_rc(c.row0 + row, col & ~3, DM5));
      v0 = d.keep(bits, col & 3) ? v0 * d.scale : 0.f;
      v1 = d.keep(bits, (col & 3) + 1) ? v1 * d.scale : 0.f;
    }
    sv0 += v0; sv1 += v1;
    if (WITH_X) {
      const bf16x2 x2 = *reinterpret_cast<const bf16x2*>(t_x + row * PT5 + col);
      sx0 = fmaf(v0, (float)x2[0], sx0);
      sx1 = fmaf(v1, (float)x2[1], sx1);
    }
  }
  if (half) {
    xch[(c.tid & 255) * 4 + 0] = sv0; xch[(c.tid & 255) * 4 + 1] = sv1;
    if (WITH_X) { xch[(c.tid & 255) * 4 + 2] = sx0; xch[(c.tid & 255) * 4 + 3] = sx1; }
  }
  __syncthreads();
  if (!half) {
    if (out_v) { atomicAdd(out_v + col, sv0 + xch[c.tid * 4 + 0]); atomicAdd(out_v + col + 1, sv1 + xch[c.tid * 4 + 1]); }
    if (WITH_X && out_vx) { atomicAdd(out_vx + col, sx0 + xch[c.tid * 4 + 2]); atomicAdd(out_vx + col + 1, sx1 + xch[c.tid * 4 + 3]); }
  }
}

// LayerNorm backward over 512 columns.  acc: the GEMM result; t_aux: the addend, receives dy (bf16, one rounding, before the
// dropout mask) in place; t_xhat: the saved normalised values, receives dx IN PLACE.  Barriers: 4 inside (row sums | the column
// sums that need xhat, their exchange | dx | its column sums' exchange); on return t_xhat holds dx, complete.
template <bool DROP>
__device__ __forceinline__ void epi_lnbwd512_p(const Ctx<2>& c, f32x16 (&acc)[2][2], bf16* t_aux, bf16* t_xhat, const float (&rs)[2],
                                               const BiasRegs (&gamma)[2], const Drop& d, float* red2, float* dgamma, float* dbeta,
                                               float* dbias) {
  constexpr int MT = 2;
  bf16x4 xh[2][MT][4], ad[2][MT][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int at = (mt * 32 + c.r) * PT5 + h * 256 + c.wave * 32 + 8 * g + 4 * c.hi;
        xh[h][mt][g] = *reinterpret_cast<const bf16x4*>(t_xhat + at);
        ad[h][mt][g] = *reinterpret_cast<const bf16x4*>(t_aux + at);
      }
  float s1[MT], s2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = mt * 32 + c.r, jl = h * 256 + c.wave * 32 + 8 * g + 4 * c.hi;
        uint32_t bits = 0;
        if (DROP) bits = d.bits(drop_counter_rc(c.row0 + row, jl, DM5));
        bf16x4 dy4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dy4[e] = (bf16)(acc[h][mt][4 * g + e] + (float)ad[h][mt][g][e]);
          float v = (float)dy4[e];
          if (DROP && d.on()) v = d.keep(bits, e) ? v * d.scale : 0.f;
          const float gg = v * gamma[h].v[g][e];
          acc[h][mt][4 * g + e] = gg;
          a1 += gg;
          a2 = fmaf(gg, (float)xh[h][mt][g][e], a2);
        }
        *reinterpret_cast<bf16x4*>(t_aux + row * PT5 + jl) = dy4;
      }
    s1[mt] = a1;
    s2[mt] = a2;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    s1[mt] = wave_sum32(s1[mt]);
    s2[mt] = wave_sum32(s2[mt]);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    *reinterpret_cast<f32x2*>(red2 + (mt * 32 + c.r) * RED2_PITCH + 2 * c.wave) = f32x2{s1[mt], s2[mt]};
  __syncthreads();
  float m1[MT], m2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(red2 + (mt * 32 + c.r) * RED2_PITCH + 4 * w4);
      a1 += p[0]; a2 += p[1];
      a1 += p[2]; a2 += p[3];
    }
    m1[mt] = a1 * (1.f / DM5);
    m2[mt] = a2 * (1.f / DM5);
  }
  __syncthreads();        // every wave holds its row sums: red2 becomes the column sums' exchange buffer
  colsum512<DROP, true>(c, t_aux, t_xhat, d, red2, dbeta, dgamma);      // (one barrier inside: behind it nobody reads xhat any more)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 dx4;
#pragma unroll
        for (int e = 0; e < 4; ++e) dx4[e] = (bf16)(rs[mt] * (acc[h][mt][4 * g + e] - m1[mt] - (float)xh[h][mt][g][e] * m2[mt]));
        *reinterpret_cast<bf16x4*>(t_xhat + (mt * 32 + c.r) * PT5 + h * 256 + c.wave * 32 + 8 * g + 4 * c.hi) = dx4;
      }
  __syncthreads();        // dx complete; the first exchange's reads are done
  const Drop off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});
  colsum512<false, false>(c, t_xhat, nullptr, off, red2, dbias, nullptr);
}

template <bool DROP>
__global__ __launch_bounds__(512, 1) void row_chain512_bwd_kernel(ChainBwdArgs a) {
  constexpr int MT = 2, RB = 64, TE = RB * PT5;
  __shared__ __attribute__((aligned(16))) bf16 tiles[2 * TE];
  __shared__ __attribute__((aligned(16))) float red2[MT * 32 * RED2_PITCH];
  Ctx<MT> c;
  c.tid = threadIdx.x; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.l = c.tid & 63; c.hi = c.l >> 5; c.r = c.l & 31;
  c.row0 = blockIdx.x * RB; c.nvalid = min(RB, a.M - c.row0);
  const bf16x8* sbase = a.wfrag + (size_t)c.wave * a.wave_frags * 64;
  auto blk = [&](int b) { return sbase + (size_t)b * 16 * 64; };
  c.ws = sbase;
#pragma unroll
  for (int i = 0; i < 8; ++i) c.ring[i] = sbase[i * 64 + c.l];
  int touched;
  {
    const int nlines = NW * a.wave_frags * 8;
    const int ln = min(((int)blockIdx.x >> 3) * 512 + c.tid, nlines - 1);
    __builtin_amdgcn_sched_barrier(0);
    touched = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(a.wfrag) + (size_t)ln * 128);
    __builtin_amdgcn_sched_barrier(0);
  }
  bf16* TA = tiles; bf16* TB = tiles + TE;
  const Drop da = make_drop(a.drop_a), off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});
  const int nu = a.nb, nbh = 2 * nu, nc = a.nc, b_tail = nbh + 4 * nc;      // nu: 256-column blocks of dP (0 or 6)
  auto B1 = [&](int ch, int j) { return blk(nbh + 4 * ch + j); };
  auto B2 = [&](int ch, int h) { return blk(nbh + 4 * ch + 2 + h); };
  auto desc = [&](const bf16* t, bf16* g, int ld) { return tile_out_desc(c, t, g, ld, PT5); };
  auto rows_rstd = [&](const float* g, float (&rs)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rs[mt] = mt * 32 + c.r < c.nvalid ? g[c.row0 + mt * 32 + c.r] : 0.f;
  };
  auto bits_of = [&](int ch) { return a.relu_bits[((size_t)(blockIdx.x * nc + ch) * NW + c.wave) * 64 + c.l]; };
  NoSide ns;
  f32x16 acc2[2][MT];
  BiasRegs gam[2];
  float rs[MT];
  zero_acc(acc2[0]);
  zero_acc(acc2[1]);

  // ---- HEAD: dy = sum_u dP_u Wp_u + G.  The dP blocks ([64][256] each) pass through TA's column halves (ping-pong); G -> TB and
  // xhat_a -> TA arrive in registers under the last blocks
  Tile5Regs rg, rx;
  {
    TileRegs<MT> nxt;
    if (nu > 0) tile_load(c, a.dP, a.ldp, nxt);
    for (int u = 0; u < nu; ++u) {
      bf16* st = TA + (u & 1) * 256;
      // (store block u: its half was last read by block u - 2's MFMAs, two barriers ago)
#pragma unroll
      for (int p = 0; p < 2 * MT; ++p) {
        const int id = c.tid + p * 512, rr = id >> 5, cc = id & 31;
        *reinterpret_cast<bf16x8*>(st + rr * PT5 + cc * 8) = nxt.v[p];
      }
      __syncthreads();
      if (u + 1 < nu) tile_load(c, a.dP + (u + 1) * 256, a.ldp, nxt);
      if (u + 1 == nu) {
        if (a.G) tile5_load(c, a.G, a.ldg, rg);
        tile5_load(c, a.xhat_a, DM5, rx);
      }
      block_mma_pt<PT5>(c, blk(u), blk(6 + u), st, acc2[0], ns);                              // (h = 0, u)
      block_mma_pt<PT5>(c, blk(6 + u), u + 1 < nu ? blk(u + 1) : B1(0, 0), st, acc2[1], ns);   // (h = 1, u)
    }
    if (nu == 0) {
      if (a.G) tile5_load(c, a.G, a.ldg, rg);
      tile5_load(c, a.xhat_a, DM5, rx);
    }
    bias_load(c, a.gamma_a, gam[0]);
    bias_load(c, a.gamma_a + 256, gam[1]);
    rows_rstd(a.rstd_a, rs);
    __syncthreads();                      // every wave is past its MFMAs on the last dP block: TA takes xhat_a
    if (a.G) tile5_store(c, rg, TB);
    else {
#pragma unroll
      for (int p = 0; p < 8; ++p) *reinterpret_cast<bf16x8*>(TB + ((c.tid + p * 512) >> 6) * PT5 + ((c.tid + p * 512) & 63) * 8) = zero_bf8();
    }
    tile5_store(c, rx, TA);
    __syncthreads();
  }
  unsigned long long relu = bits_of(0);
  epi_lnbwd512_p<DROP>(c, acc2, TB, TA, rs, gam, da, red2, a.dgamma_a, a.dbeta_a, a.dbias_a);      // TB: G -> dy; TA: xhat_a -> ds_a
  __syncthreads();
  bf16* cur = TA;         // ds_a; TB is free

  // ---- FFN
  const int dff = nc * 256;
  f32x16 acc1[MT];
  zero_acc(acc2[0]);
  zero_acc(acc2[1]);
  zero_acc(acc1);
  {       // B1_0's two blocks carry the copy of ds_a (two 256-column sub-tiles)
    CopySide<MT, 1> cs0{c, {desc(cur, a.ds_a, DM5)}};
    block_mma_pt<PT5>(c, B1(0, 0), B1(0, 1), cur, acc1, cs0);
    CopySide<MT, 1> cs1{c, {desc(cur + 256, a.ds_a + 256, DM5)}};
    block_mma_pt<PT5>(c, B1(0, 1), nc > 1 ? B1(1, 0) : B2(0, 0), cur + 256, acc1, cs1);
  }
  {
    MaskSide<MT, PT5> m0{c, acc1, TB, a.mask_scale, (uint32_t)relu, (uint32_t)(relu >> 32)};
    m0.all();
  }
  if (nc > 1) relu = bits_of(1);
  __syncthreads();
  // dH chunk c lives in TB's column half (c & 1); its copy rides under B1_(c+1)
  for (int ch = 0; ch + 1 < nc; ++ch) {
    bf16* hc = TB + (ch & 1) * 256;
    bf16* hn = TB + ((ch + 1) & 1) * 256;
    zero_acc(acc1);
    {
      CopySide<MT, 1> cs{c, {desc(hc, a.dH + ch * 256, dff)}};
      block_mma_pt<PT5>(c, B1(ch + 1, 0), B1(ch + 1, 1), cur, acc1, cs);
    }
    block_mma_pt<PT5>(c, B1(ch + 1, 1), B2(ch, 0), cur + 256, acc1, ns);
    {
      MaskSide<MT, PT5> ms{c, acc1, hn, a.mask_scale, (uint32_t)relu, (uint32_t)(relu >> 32)};
      if (ch + 2 < nc) relu = bits_of(ch + 2);
      block_mma_pt<PT5>(c, B2(ch, 0), B2(ch, 1), hc, acc2[0], ms);
    }
    block_mma_pt<PT5>(c, B2(ch, 1), ch + 2 < nc ? B1(ch + 2, 0) : B2(ch + 1, 0), hc, acc2[1], ns);
    __syncthreads();
  }
  {
    bf16* hl = TB + ((nc - 1) & 1) * 256;
    tile5_load(c, a.xhat_b, DM5, rx);
    bias_load(c, a.gamma_b, gam[0]);
    bias_load(c, a.gamma_b + 256, gam[1]);
    rows_rstd(a.rstd_b, rs);
    {
      CopySide<MT, 1> cs{c, {desc(hl, a.dH + (nc - 1) * 256, dff)}};
      block_mma_pt<PT5>(c, B2(nc - 1, 0), B2(nc - 1, 1), hl, acc2[0], cs);
    }
    block_mma_pt<PT5>(c, B2(nc - 1, 1), blk(b_tail), hl, acc2[1], ns);
    __syncthreads();                      // every wave is past its MFMAs on and its copy of the last chunk: TB takes xhat_b
    tile5_store(c, rx, TB);
    __syncthreads();
  }
  epi_lnbwd512_p<false>(c, acc2, cur, TB, rs, gam, off, red2, a.dgamma_b, a.dbeta_b, a.dbias_b);      // TA: ds_a -> dy; TB: xhat_b -> ds_b
  __syncthreads();
  cur = TB;               // ds_b; TA is free

  // ---- TAIL: dctx = ds_b Wo (h, j), delta; ds_b leaves beside the blocks; O and Ores arrive in registers under them
  zero_acc(acc2[0]);
  zero_acc(acc2[1]);
  tile5_load(c, a.O, a.ldo, rg);
  if (a.Ores) tile5_load(c, a.Ores, a.ldo, rx);
  {
    CopySide<MT, 1> cs0{c, {desc(cur, a.ds_b, DM5)}};
    block_mma_pt<PT5>(c, blk(b_tail), blk(b_tail + 1), cur, acc2[0], cs0);
    CopySide<MT, 1> cs1{c, {desc(cur + 256, a.ds_b + 256, DM5)}};
    block_mma_pt<PT5>(c, blk(b_tail + 1), blk(b_tail + 2), cur + 256, acc2[0], cs1);
  }
  tile5_store(c, rg, TA);                 // O
  block_mma_pt<PT5>(c, blk(b_tail + 2), blk(b_tail + 3), cur, acc2[1], ns);
  block_mma_pt<PT5>(c, blk(b_tail + 3), blk(b_tail + 4), cur + 256, acc2[1], ns);
  __syncthreads();                        // O visible; every wave is past its MFMAs on and its copies of ds_b: TB stages dctx
  bf16x4 dc[2][MT][4];
  float part[2][MT];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float p = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int at = (mt * 32 + c.r) * PT5 + h * 256 + c.wave * 32 + 8 * g + 4 * c.hi;
        const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(TA + at);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (bf16)acc2[h][mt][4 * g + e];
          p = fmaf((float)o[e], (float)o4[e], p);
        }
        dc[h][mt][g] = o;
        *reinterpret_cast<bf16x4*>(cur + at) = o;
      }
      part[h][mt] = p;
    }
  if (a.Ores) {
    __syncthreads();                      // every wave has read O
    tile5_store(c, rx, TA);               // Ores
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(TA + (mt * 32 + c.r) * PT5 + h * 256 + c.wave * 32 + 8 * g + 4 * c.hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) part[h][mt] = fmaf((float)dc[h][mt][g][e], (float)r4[e], part[h][mt]);
        }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red2[((h * 8 + c.wave) * MT + mt) * 32 + c.r] = wave_sum32(part[h][mt]);      // half a head
  __syncthreads();
  tile_out_now(c, desc(cur, a.dctx, a.lddc));
  tile_out_now(c, desc(cur + 256, a.dctx + 256, a.lddc));
  for (int i = c.tid; i < 8 * RB; i += 512) {      // delta[head][row]: head = 4 h + (wave >> 1), 64 columns = two waves
    const int hd = i / RB, row = i % RB, mt = row >> 5, r = row & 31, h = hd >> 2, w0 = (hd & 3) * 2;
    if (row < c.nvalid)
      a.delta[(size_t)hd * a.M + c.row0 + row] = red2[((h * 8 + w0) * MT + mt) * 32 + r] + red2[((h * 8 + w0 + 1) * MT + mt) * 32 + r];
  }
  if (touched == 0x5a5a5a5a && a.M < 0) red2[0] = 1.f;      // (never true: keeps the warm-up load alive)
}

}  // namespace
