// The backward row chain (HEAD + FFN + TAIL: the encoder's shape) at 64- / 96-row workgroups, pipelined as the forward kernel
// of st_rowchain_pipe.cuh is - included by st_rowchain.hip behind ChainBwdArgs:
//   * the feed-forward's blocks run in the order B1_0, B1_1, B2_0, B1_2, B2_1, ..  (B1_c = ds x W2[:, c] -> the hidden gradient of
//     chunk c before the mask, B2_c = dH_c x W1[c, :] accumulating dy): the mask / scale / bf16 epilogue of B1_(c+1) is side work
//     of B2_c, the copy of dH chunk c to HBM side work of B1_(c+1) (both only read its tile);
//   * the normalised gradients ds_a / ds_b leave as side work of the block that multiplies them next;
//   * gamma, rstd, the ReLU bits and the xhat / O / Ores tiles are requested a block before they are used (a vector load asked for
//     where it is needed waits behind the whole in-order queue - ring and copies);
//   * the LayerNorm backward exchanges its two row sums once, without a branch, every LDS read of a pass in front of its
//     arithmetic; saved-tensor copies are write-through through range-checked buffer descriptors.
#pragma once

namespace {

// the mask / scale / bf16 epilogue of a B1 block as side work: dH = on ? bf16(acc * mask_scale) : 0 into the LDS tile
template <int MT, int PT = AS> struct MaskSide {
  const Ctx<MT>& c;
  const f32x16 (&acc)[MT];
  bf16* t;
  float scale;
  uint32_t lo, hi;        // this lane's ReLU bits of the chunk (bit 16 mt + 4 g + e)
  static constexpr int NU = 4 * MT;
  __device__ __forceinline__ void unit(int j) {
    const int mt = j / 4, g = j % 4;
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int b = mt * 16 + 4 * g + e;
      // (-1 / 0 from one v_bfe_i32, then an AND on the fp32 product: the rounding of a kept value is the plain kernels')
      const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe(b < 32 ? (int)lo : (int)hi, b & 31, 1);
      o[e] = (bf16)__builtin_bit_cast(float, f2u(acc[mt][4 * g + e] * scale) & m);
    }
    *reinterpret_cast<bf16x4*>(t + (mt * 32 + c.r) * PT + c.wave * 32 + 8 * g + 4 * c.hi) = o;
  }
  __device__ __forceinline__ void a(int) {}
  __device__ __forceinline__ void b(int k2) {
#pragma unroll
    for (int j = (k2 * NU) / 8; j < ((k2 + 1) * NU) / 8; ++j) unit(j);
  }
  __device__ __forceinline__ void all() {
#pragma unroll
    for (int j = 0; j < NU; ++j) unit(j);
  }
};

// LayerNorm backward as epi_lnbwd (st_rowchain.hip), gamma and the rows' rstd handed in as registers; dx is left in t_dx (the
// caller copies it out beside its next block).  One barrier for the row sums, one before the column pass (which uses red2 as
// its exchange buffer: one more barrier inside); the caller needs one more before t_aux / t_xhat are rewritten.
template <bool DROP, int MT>
__device__ __forceinline__ void epi_lnbwd_p(const Ctx<MT>& c, f32x16 (&acc)[MT], bf16* t_aux, const bf16* t_xhat, bf16* t_dx,
                                            const float (&rs)[MT], const BiasRegs& gamma, const Drop& d, float* red2, float* dgamma,
                                            float* dbeta, float* dbias) {
  const int j0 = c.wave * 32;
  bf16x4 xh[MT][4], ad[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int at = (mt * 32 + c.r) * AS + j0 + 8 * g + 4 * c.hi;
      xh[mt][g] = *reinterpret_cast<const bf16x4*>(t_xhat + at);
      ad[mt][g] = *reinterpret_cast<const bf16x4*>(t_aux + at);
    }
  float s1[MT], s2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = mt * 32 + c.r, jl = j0 + 8 * g + 4 * c.hi;
      uint32_t bits = 0;
      if (DROP) bits = d.bits(drop_counter_rc(c.row0 + row, jl, DM));   // the mask the forward drew on this LayerNorm's output
      bf16x4 dy4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dy4[e] = (bf16)(acc[mt][4 * g + e] + (float)ad[mt][g][e]);      // one rounding, as st_gemm_lnbwd
        float v = (float)dy4[e];
        if (DROP && d.on()) v = d.keep(bits, e) ? v * d.scale : 0.f;
        const float gg = v * gamma.v[g][e];
        acc[mt][4 * g + e] = gg;                                         // keep g = dy * gamma
        a1 += gg;
        a2 = fmaf(gg, (float)xh[mt][g][e], a2);
      }
      *reinterpret_cast<bf16x4*>(t_aux + row * AS + jl) = dy4;
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
  f32x4 pp[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) pp[mt][w4] = *reinterpret_cast<const f32x4*>(red2 + (mt * 32 + c.r) * RED2_PITCH + 4 * w4);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    // (summed in wave order 0 .. 7, as epi_lnbwd does)
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) {
      a1 += pp[mt][w4][0]; a2 += pp[mt][w4][1];
      a1 += pp[mt][w4][2]; a2 += pp[mt][w4][3];
    }
    const float m1 = a1 * (1.f / DM), m2 = a2 * (1.f / DM);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 dx4;
#pragma unroll
      for (int e = 0; e < 4; ++e) dx4[e] = (bf16)(rs[mt] * (acc[mt][4 * g + e] - m1 - (float)xh[mt][g][e] * m2));
      *reinterpret_cast<bf16x4*>(t_dx + (mt * 32 + c.r) * AS + j0 + 8 * g + 4 * c.hi) = dx4;
    }
  }
  __syncthreads();
  // column sums: thread = (column, half of the rows), the upper half hands its sums over through LDS: ONE atomic per column,
  // quantity and workgroup (see epi_lnbwd)
  {
    const int col = c.tid & 255, half = c.tid >> 8, rows = 16 * MT;
    float cg = 0.f, cb = 0.f, cx = 0.f;
    for (int i = 0; i < rows; ++i) {
      const int row = half * rows + i;
      float v = (float)t_aux[row * AS + col];
      if (DROP && d.on()) {
        const uint32_t bits = d.bits(drop_counter_rc(c.row0 + row, col & ~3, DM));
        v = d.keep(bits, col & 3) ? v * d.scale : 0.f;
      }
      cb += v;
      cg += v * (float)t_xhat[row * AS + col];
      cx += (float)t_dx[row * AS + col];
    }
    float* xch = red2;          // MT * 32 * RED2_PITCH >= 768 floats (the row sums were consumed before the barrier above)
    if (half) { xch[col] = cg; xch[256 + col] = cb; xch[512 + col] = cx; }
    __syncthreads();
    if (!half) {
      if (dgamma) atomicAdd(dgamma + col, cg + xch[col]);
      if (dbeta) atomicAdd(dbeta + col, cb + xch[256 + col]);
      if (dbias) atomicAdd(dbias + col, cx + xch[512 + col]);
    }
  }
}

template <bool DROP, int MT>
__global__ __launch_bounds__(512, 1) void row_chain_bwd_pipe_kernel(ChainBwdArgs a) {
  static_assert(MT >= 2, "pipelined backward chain: 64- / 96-row workgroups");
  constexpr int RB = 32 * MT, TE = RB * AS;
  __shared__ __attribute__((aligned(16))) bf16 tiles[3 * TE];
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
#if ST_PIPE_PRIO
  if (c.wave >= 4) __builtin_amdgcn_s_setprio(1);      // the later-dispatched half loses every arbitration against its SIMD partner otherwise
#endif
  bf16* T0 = tiles; bf16* T1 = tiles + TE; bf16* T2 = tiles + 2 * TE;
  const Drop da = make_drop(a.drop_a), off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});
  const int nbh = a.nb, nc = a.nc, b_tail = nbh + 2 * nc;      // stream: HEAD 0 .. | B1_c nbh + 2c, B2_c nbh + 2c + 1 | TAIL
  NoSide ns;
  auto rows_rstd = [&](const float* g, float (&rs)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rs[mt] = mt * 32 + c.r < c.nvalid ? g[c.row0 + mt * 32 + c.r] : 0.f;
  };
  auto bits_of = [&](int ch) { return a.relu_bits[((size_t)(blockIdx.x * nc + ch) * NW + c.wave) * 64 + c.l]; };

  // ---- HEAD: dy = sum_u dP_u Wp_u + G, LayerNorm backward a.  T0: the dP blocks, then ds_a; T1: G -> dy; T2: xhat_a
  f32x16 acc2[MT];        // (HEAD's accumulator, then the feed-forward's dy)
  zero_acc(acc2);
  BiasRegs gam;
  float rs[MT];
  {
    TileRegs<MT> nxt;
    {
      TileRegs<MT> rg, rx;
      if (a.G) tile_load(c, a.G, a.ldg, rg);
      tile_load(c, a.xhat_a, DM, rx);
      if (nbh > 0) tile_load(c, a.dP, a.ldp, nxt);
      bias_load(c, a.gamma_a, gam);
      rows_rstd(a.rstd_a, rs);
      if (a.G) tile_store(c, rg, T1);
      else {
#pragma unroll
        for (int p = 0; p < 2 * MT; ++p) *reinterpret_cast<bf16x8*>(T1 + ((c.tid + p * 512) >> 5) * AS + ((c.tid + p * 512) & 31) * 8) = zero_bf8();
      }
      tile_store(c, rx, T2);
    }
    for (int u = 0; u < nbh; ++u) {
      tile_store(c, nxt, T0);
      __syncthreads();
      if (u + 1 < nbh) tile_load(c, a.dP + (u + 1) * 256, a.ldp, nxt);
      block_mma_p(c, blk(u), blk(u + 1), T0, acc2, ns);
      __syncthreads();                           // every wave is past its MFMAs on this block of dP: T0 may be rewritten
    }
    if (nbh == 0) __syncthreads();               // (bare LayerNorm backward: the G / xhat tiles must be visible)
  }
  unsigned long long relu = bits_of(0);          // (requested in front of the LayerNorm, used behind it)
  epi_lnbwd_p<DROP>(c, acc2, T1, T2, T0, rs, gam, da, red2, a.dgamma_a, a.dbeta_a, a.dbias_a);
  __syncthreads();                               // the column pass has read T1 / T2: free from here
  bf16* cur = T0;         // the running gradient ds
  bf16 *X = T1, *Y = T2;

  // ---- FFN
  const int dff = nc * 256;
  f32x16 acc1[MT];
  zero_acc(acc2);
  zero_acc(acc1);
  {       // B1_0 with the copy of ds_a beside it
    CopySide<MT, 1> cs{c, {tile_out_desc(c, cur, a.ds_a, DM)}};
    block_mma_p(c, blk(nbh), nc > 1 ? blk(nbh + 2) : blk(nbh + 1), cur, acc1, cs);
  }
  {       // chunk 0's mask epilogue is the one nothing hides
    MaskSide<MT> m0{c, acc1, X, a.mask_scale, (uint32_t)relu, (uint32_t)(relu >> 32)};
    m0.all();
  }
  if (nc > 1) relu = bits_of(1);
  __syncthreads();
  // from here: dH chunk c lives in (c even ? X : Y); its copy rides under B1_(c+1)
  for (int ch = 0; ch + 1 < nc; ++ch) {
    bf16* hc = (ch & 1) ? Y : X;
    bf16* hn = (ch & 1) ? X : Y;
    zero_acc(acc1);
    {
      CopySide<MT, 1> cs{c, {tile_out_desc(c, hc, a.dH + ch * 256, dff)}};
      block_mma_p(c, blk(nbh + 2 * (ch + 1)), blk(nbh + 2 * ch + 1), cur, acc1, cs);
    }
    {
      MaskSide<MT> ms{c, acc1, hn, a.mask_scale, (uint32_t)relu, (uint32_t)(relu >> 32)};
      if (ch + 2 < nc) relu = bits_of(ch + 2);
      block_mma_p(c, blk(nbh + 2 * ch + 1), ch + 2 < nc ? blk(nbh + 2 * (ch + 2)) : blk(nbh + 2 * (ch + 1) + 1), hc, acc2, ms);
    }
    __syncthreads();
  }
  bf16* hl = ((nc - 1) & 1) ? Y : X;      // the last chunk
  bf16* tx = ((nc - 1) & 1) ? X : Y;      // free: takes xhat_b
  {
    TileRegs<MT> xr;
    tile_load(c, a.xhat_b, DM, xr);
    bias_load(c, a.gamma_b, gam);
    rows_rstd(a.rstd_b, rs);
    CopySide<MT, 1> cs{c, {tile_out_desc(c, hl, a.dH + (nc - 1) * 256, dff)}};
    block_mma_p(c, blk(nbh + 2 * (nc - 1) + 1), blk(b_tail), hl, acc2, cs);
    tile_store(c, xr, tx);
  }
  __syncthreads();                               // xhat_b visible; every wave is past its MFMAs on and its copy of the last chunk
  // dy = acc2 + ds in place over the ds tile, ds_b into the last chunk's tile
  epi_lnbwd_p<false>(c, acc2, cur, tx, hl, rs, gam, off, red2, a.dgamma_b, a.dbeta_b, a.dbias_b);
  __syncthreads();                               // the column pass has read cur / tx
  bf16* fa = cur;         // takes O
  bf16* fb = tx;          // takes Ores
  cur = hl;               // ds_b

  // ---- TAIL: dctx = ds_b Wo, delta; ds_b leaves beside the block
  {
    TileRegs<MT> ro, rr_;
    tile_load(c, a.O, a.ldo, ro);
    if (a.Ores) tile_load(c, a.Ores, a.ldo, rr_);
    zero_acc(acc1);
    CopySide<MT, 1> cs{c, {tile_out_desc(c, cur, a.ds_b, DM)}};
    block_mma_p(c, blk(b_tail), blk(b_tail + 1), cur, acc1, cs);
    tile_store(c, ro, fa);
    if (a.Ores) tile_store(c, rr_, fb);
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = mt * 32 + c.r;
    float part = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int at = row * AS + c.wave * 32 + 8 * g + 4 * c.hi;
      const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(fa + at);
      bf16x4 r4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      if (a.Ores) r4 = *reinterpret_cast<const bf16x4*>(fb + at);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (bf16)acc1[mt][4 * g + e];
        part += (float)o[e] * ((float)o4[e] + (float)r4[e]);
      }
      *reinterpret_cast<bf16x4*>(cur + at) = o;      // (dctx is staged in the ds_b tile: every wave is past its MFMAs and copies on it)
    }
    part = wave_sum32(part);
    red2[(c.wave * MT + mt) * 32 + c.r] = part;      // this wave's 32 columns of the row: half a head
  }
  __syncthreads();
  tile_out_now(c, tile_out_desc(c, cur, a.dctx, a.lddc));
  for (int i = c.tid; i < 4 * RB; i += 512) {      // delta[h][row]: heads are 64 columns = two waves
    const int h = i / RB, row = i % RB, mt = row >> 5, r = row & 31;
    if (row < c.nvalid)
      a.delta[(size_t)h * a.M + c.row0 + row] = red2[((2 * h) * MT + mt) * 32 + r] + red2[((2 * h + 1) * MT + mt) * 32 + r];
  }
  if (touched == 0x5a5a5a5a && a.M < 0) red2[0] = 1.f;      // (never true: keeps the warm-up load alive)
}

}  // namespace
