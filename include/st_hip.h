/* st_hip.h - C ABI of libst_hip.so, the MI355X (gfx950) kernels behind the
 * speech-transformer training step.
 *
 * The reference (ZhengkunTian/Speech-Tranformer-Pytorch) has no FFI: its hot
 * path is the nn.Module surface of transformer/{Attention,SubLayers,Layers,
 * Embedding,Models}.py.  The drop-in keeps those Python names (package
 * speech-tranformer-pytorch_amd/transformer) and binds THIS library through
 * ctypes (speech-tranformer-pytorch_amd/st_amd/native.py); INTEGRATION.md shows
 * the stub.  Every entry point below names the reference lines it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers;
 *  - every call is asynchronous on `stream` (hipStream_t passed as void*),
 *    holds no global state and never synchronises the device (it is called
 *    from the autograd engine thread as well as the main thread);
 *  - return 0 on success, a positive hipError_t if the launch failed, a
 *    negative value for an argument the kernels do not support (the Python
 *    wrapper raises RuntimeError / ValueError);
 *  - "bf16" buffers are IEEE bfloat16 (uint16 storage); activations are row
 *    matrices [rows, ld]; utterance b owns rows off[b] .. off[b]+len[b]-1
 *    (packed or padded - the kernels only see offsets and lengths);
 *  - leading dimensions are in ELEMENTS and must be multiples of 8.
 */
#ifndef ST_HIP_H
#define ST_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* st_stream_t; /* hipStream_t */

/* Library ABI version, bumped on any signature change (2: round 5's k_prescaled / dense attention / grad_scale arguments;
 * 3, 4: round 6 - 4 added the column-sum workspace of st_row_chain_bwd).  A host binding must refuse a library whose st_version() differs from the header it was written against:
 * through ctypes / dlsym a stale libst_hip.so would be called with shifted arguments (st_amd/native.py: ABI_VERSION). */
#define ST_ABI_VERSION 4
int st_version(void);

/* Re-read the development switches that choose between a specialised attention kernel and the general one
 * (ST_ATTN_FWD64=0, ST_ATTN_XS=0, ST_ATTN_BWD64=0|e).  They are read from the environment once, at the first
 * attention call; a host that changes one afterwards (same-process A/B runs in tests/ and tools/dev/) calls this. */
int st_env_refresh(void);

/* Epilogue selector of st_gemm. */
enum {
  ST_EPI_BF16 = 0,       /* D(bf16) = acc (+ bias)                                   */
  ST_EPI_BF16_RELU = 1,  /* D(bf16) = relu(acc + bias)      SubLayers.py:25          */
  ST_EPI_F32 = 2,        /* D(f32)  = acc (+ bias)          Models.py:151 (logits)   */
  ST_EPI_BF16_MASK = 3,  /* D(bf16) = acc * (aux > 0)       ReLU backward            */
  ST_EPI_BF16_ADD = 4,   /* D(bf16) = acc + aux             residual-gradient add    */
  ST_EPI_F32_ATOMIC = 5, /* D(f32) += acc (atomic, split-K)                          */
  ST_EPI_F32_ATOMIC_T = 6, /* D^T(f32)[j][i] += acc: weight gradients, coalesced atomics */
  ST_EPI_BF16_DELTA = 7  /* D(bf16) = acc, and delta[h][i] = sum_{j in head h} D(i,j) * (aux(i,j) + aux2(i,j)); aux2
                            (nullable, bf16, ld = ldaux) = st_attn_fwd's Ores: what rounding O to bf16 dropped */
};

/* Training-mode dropout (nn.Dropout in Attention.py:89, SubLayers.py:25,27,
 * Models.py:31).  Every entry point that can drop takes the same four
 * arguments: drop_seed (DEVICE pointer to a 32-bit seed; NULL = no dropout),
 * drop_salt (per-call-site constant), drop_thresh (round(256 p); an element
 * is kept iff its 8-bit draw >= thresh) and drop_scale (256 / (256 - thresh)).
 * Masks are a pure function of (*drop_seed, salt, element index): backward
 * entry points regenerate them from the same arguments, and a captured HIP
 * graph draws new masks on every replay once *drop_seed is advanced. */

/* D[i][j] = sum_c X(i,c) * Y(j,c), bf16 operands, fp32 accumulate (MFMA).
 * x_cmajor / y_cmajor: operand stored [c][rows] instead of [rows][c].
 * Replaces the three GEMMs of every nn.Linear on the path
 * (Attention.py:74-76,92; SubLayers.py:25-26; Models.py:145,151 and their
 * autograd backward): forward (0,0), dgrad (0,1), wgrad (1,1).
 * M rows of X, N rows of Y, Kc contraction length; `splits` > 1 only with
 * ST_EPI_F32_ATOMIC / ST_EPI_F32_ATOMIC_T (the latter stores the transposed
 * result: D is [N, ldd >= M]).
 * bias: fp32 [N], read and added to the accumulator - EXCEPT with
 * ST_EPI_F32_ATOMIC_T, where a non-NULL bias is the fp32 [N] bias-GRADIENT
 * accumulator: bias[j] += sum_c Y(j,c) (the column sums of dy, i.e. the
 * nn.Linear bias gradient, produced by the weight-gradient launch itself).
 * ST_EPI_BF16_DELTA (the dgrad that produces the attention backward's dO = d(context), aux = the forward
 * context O): `bias` is the fp32 OUTPUT delta [N / head, M] and `splits` carries the head width (32 or 64).
 * Dropout: ST_EPI_BF16_RELU drops after the ReLU (SubLayers.py:25);
 * ST_EPI_BF16_MASK multiplies the surviving (aux > 0) elements by drop_scale
 * (backward of the former; aux is the dropped activation). */
int st_gemm(st_stream_t stream, int x_cmajor, int y_cmajor, const void* X, int ldx, const void* Y, int ldy, void* D,
            int ldd, int M, int N, int Kc, float* bias, const void* aux, int ldaux, int epi, int splits,
            const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale, const void* aux2);

/* D (bf16 [M, N]) = X Y^T + bias (operands row-major as st_gemm's plain forward form) with the output columns
 * [col_lo, col_hi) - multiples of 32 - multiplied by `scale` in fp32 BEFORE their one rounding to bf16: the q | k | v
 * projection of Attention.py:74-76 with its key block pre-scaled by scale * log2(e) for st_attn_fwd / st_attn_bwd's
 * k_prescaled mode (the row chains do the same through st_row_chain's post_kscale). */
int st_gemm_kscale(st_stream_t stream, const void* X, int ldx, const void* Y, int ldy, void* D, int ldd, int M, int N, int Kc,
                   float* bias, int col_lo, int col_hi, float scale);

/* D (bf16 [M, N]) = X Y^T (y_cmajor as st_gemm) for FEW output tiles and a LONG contraction (the vocabulary projection's input
 * gradient, Models.py:151 backward: 1,206 x 256 over 4,344): the contraction is cut `splits` ways over workgroups, every
 * split leaves its fp32 partial tile in `work` (write-through) and the last one to finish a tile adds them in split order
 * and rounds once - no atomic adds; the result does not depend on arrival order (it differs from st_gemm's in fp32
 * summation order only).  work: device scratch of 4096 + tiles * splits * 65536 bytes (tiles = ceil(M / 128) *
 * ceil(N / 128) <= 1024) whose first 4096 bytes are zero before the first call (tickets; the launch leaves them zero). */
int st_gemm_splitk(st_stream_t stream, int y_cmajor, const void* X, int ldx, const void* Y, int ldy, void* D, int ldd, int M,
                   int N, int Kc, int splits, void* work, long long work_bytes);

/* st_gemm whose Y operand (and bias) is a stack of equally shaped blocks lying y_block_stride (bias_block_stride)
 * elements apart in memory - the same nn.Linear weight of consecutive identical layers as the parameter arena
 * lays them out.  Forward (0,0): N = blocks * y_block_rows output columns, block b = rows [b * y_block_rows, ..);
 * dgrad (0,1): Kc = blocks * y_block_rows, i.e. D = sum_b X[:, block b] W_b.  y_block_rows: a power of two >= 128
 * (0 = plain st_gemm).  Used for the decoder-encoder attention of ALL decoder layers at once: their key/value
 * projections of the encoder output (Attention.py:75-76, one launch instead of one per layer) and the matching
 * input gradient.  Not available for the split-K / DELTA epilogues. */
int st_gemm_stacked(st_stream_t stream, int x_cmajor, int y_cmajor, const void* X, int ldx, const void* Y, int ldy,
                    void* D, int ldd, int M, int N, int Kc, float* bias, const void* aux, int ldaux, int epi, int splits,
                    const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale, int y_block_rows,
                    long y_block_stride, long bias_block_stride);

/* Weight-stationary streaming GEMM for the short-contraction products (csrc/st_gemm_ws.hip): D (bf16) = epi(X W^T + bias)
 * with K = 256, N a multiple of 256, X [M, K] and W [N, K] natural.  epi: 0 = identity, 1 = ReLU (+ dropout, as
 * ST_EPI_BF16_RELU).  w_block_rows > 0: W / bias are stacks of equally spaced blocks as in st_gemm_stacked
 * (w_block_rows a power-of-two multiple of 256).  Replaces st_gemm(0, 0, ...) for Attention.py:74-76, SubLayers.py:25. */
int st_gemm_ws(st_stream_t stream, const void* X, int ldx, const void* W, int ldw, void* D, int ldd, int M, int N, int K,
               const float* bias, int epi, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh,
               float drop_scale, int w_block_rows, long w_block_stride, long bias_block_stride);

/* n weight-gradient problems in ONE launch (the decoder's are ~20 workgroups
 * each: launched one by one they are pure latency).  Problem q:
 *   dW[q][N_out, K_in] (f32, ld lddw) += dY[q]^T X[q],  db[q][N_out] += colsum(dY[q])   (db or db[q] may be NULL)
 * with X[q] bf16 [tokens, K_in] (ld ldx), dY[q] bf16 [tokens, N_out] (ld lddy),
 * i.e. st_gemm(1, 1, X, ldx, dY, lddy, dW, lddw, K_in, N_out, tokens, db, .., ST_EPI_F32_ATOMIC_T, splits[q]).
 * All arrays are HOST arrays of length n (the descriptors travel in the kernel arguments). */
int st_wgrad_group(st_stream_t stream, int n, const void* const* X, const int* ldx, const void* const* dY,
                   const int* lddy, float* const* dW, const int* lddw, float* const* db, const int* tokens,
                   const int* K_in, const int* N_out, const int* splits);

/* The same contract for ENCODER-sized token counts (st_wgrad.hip): 256 x 256 output tiles, one 8-wave workgroup per
 * CU.  splits[q] cuts problem q's token axis; the caller picks it so that the launch has about one workgroup per CU:
 * sum_q ceil(K_in/256) * ceil(N_out/256) * splits[q] ~ 256 (st_amd/functional.py: flush_deferred_wgrads). */
int st_wgrad_wide(st_stream_t stream, int n, const void* const* X, const int* ldx, const void* const* dY,
                  const int* lddy, float* const* dW, const int* lddw, float* const* db, const int* tokens,
                  const int* K_in, const int* N_out, const int* splits);

/* out = LayerNorm(act(X W^T + bias) + res) * gamma + beta (+ pe[pos[row]]),
 * N = d_model in {128, 256, 512}.  Replaces output_linear + residual +
 * layernorm (Attention.py:92-94), fc2 + residual + layernorm
 * (SubLayers.py:26-27) and, with relu=1 and the PE add, the encoder front-end
 * (Models.py:28-33,42-44).  Saves xhat (bf16 [M,N]) and rstd (f32 [M]) for the
 * backward; `pre` (optional) receives the pre-LN value (front-end ReLU mask).
 * drop_where: 1 = dropout on act(X W^T + bias) before the LayerNorm
 * (Models.py:31), 2 = dropout on the LayerNorm output (SubLayers.py:27). */
int st_gemm_ln(st_stream_t stream, const void* X, int ldx, const void* W, int M, int N, int K, const float* bias,
               const void* res, int ldres, const float* gamma, const float* beta, float eps, int relu,
               const float* pe, const int* pos, void* out, int ldo, void* xhat, float* rstd, void* pre,
               const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale, int drop_where);

/* Data-gradient GEMM + LayerNorm backward in one launch (the backward analogue of st_gemm_ln):
 *   dy = bf16(dY W (+ aux)),  dx = LayerNorm-backward(dy; xhat, rstd, gamma),  dgamma/dbeta += ..., dbias += colsum(dx)
 * dY bf16 [M, Kc], W bf16 [Kc, N] (ld ldw: an nn.Linear weight [out = Kc, in = N] as stored), N = d_model in
 * {128, 256, 512}; identical to st_gemm(0, 1, .., ST_EPI_BF16[_ADD]) followed by st_ln_bwd, without dy going to
 * HBM.  Used where a sublayer's input gradient is the previous sublayer's LayerNorm output gradient
 * (SubLayers.py:25-27 -> Attention.py:94 backward, and so on down the stack).  drop_*: the forward dropped the
 * LayerNorm output (st_gemm_ln with drop_where = 2, SubLayers.py:27); the same mask is regenerated here. */
int st_gemm_lnbwd(st_stream_t stream, const void* dY, int lddy, const void* W, int ldw, int M, int N, int Kc,
                  const void* aux, int ldaux, const void* xhat, const float* rstd, const float* gamma, void* dx,
                  int lddx, float* dgamma, float* dbeta, float* dbias, const unsigned* drop_seed, unsigned drop_salt,
                  int drop_thresh, float drop_scale);

/* ---- row chains for decoder-sized row counts (csrc/st_rowchain.hip) ---------------------------------------------
 * Everything the decoder does between two attention kernels is row-wise; one launch runs it for blocks of 32 rows:
 *   PRE   cur = LN(A Wo^T + bo + R) * g0 + be0               (Attention.py:92-94)   out0 / xhat0 / rstd0 as st_gemm_ln
 *                                                                                    (each may be NULL when nothing outside the chain reads it)
 *   FFN   H   = dropout1(relu(cur W1^T + b1))                (SubLayers.py:25)      [M, d_ff], saved for the backward
 *                                                                                    (H may be NULL: inference - it stays on the chip)
 *         cur = dropout2(LN(H W2^T + b2 + cur) * g1 + be1)   (SubLayers.py:26-27)   out1 / xhat1 / rstd1
 *   POST  P   = cur Wp^T + bp, Wp [256 post_blocks, 256]     (Attention.py:74-76 of the NEXT attention)
 * PRE is present iff R != NULL, FFN iff d_ff > 0, POST iff post_blocks > 0; without PRE the chain input is A.
 * d_model = 256, d_ff % 256 == 0.  The weights are read from per-wave fragment streams: every GEMM is cut into
 * 256 x 256 blocks, consumed in the order  Wo | (W1 rows c*256.., W2 columns c*256..) for c = 0.. | Wp rows u*256..;
 * n_blocks = their number.  st_wfrag_build lays the blocks out (table: 4 x int64 per block on the device - address of
 * the block's first element in a row-major bf16 matrix, its leading dimension (| 1 << 32: read the block transposed),
 * 16 * (position of the block in its chain) | wave stride (n_blocks * 16 + st_wfrag_depth()) << 32, address of the
 * chain's buffer); a chain's buffer holds 8 * (n_blocks * 16 + st_wfrag_depth()) * 512 bf16.  One table may fill any
 * number of chains.  Rebuild after every weight update.
 * next_blocks > 0: another chain of that many blocks is stored right behind this one and runs next - the launch warms the
 * L2 with its streams too (the streams are read once per step, from HBM).
 * Both dropout sites read the device seed *drop_seed (NULL = off) with their own salt / threshold / scale.
 * split_work (optional; split_bytes = its size): zero-initialised device scratch of (256 + 256 * 8192) * 4 bytes (256 tickets, then the partial sums; only the tickets need the zeroes).  With it, a
 * chain that has an FFN part and whose M / 32 row blocks x (d_ff / 256) fit 256 workgroups runs d_ff / 256 workgroups per row
 * block: each does PRE, ONE 256-wide chunk of the hidden dimension and leaves its partial of the second GEMM in the
 * scratch; the last of a row block's workgroups adds them (in chunk order: the result does not depend on arrival order, but
 * differs in rounding from the one-workgroup chain) and finishes the chain.  Launches that share the scratch must be
 * ordered on one stream. */
int st_wfrag_depth(void);
int st_wfrag_build(st_stream_t stream, const long long* table, int n_blocks);
int st_row_chain(st_stream_t stream, int M, const void* wfrag, int n_blocks, int next_blocks, float eps, const void* A,
                 int lda,
                 const void* R, int ldr, const float* bo, const float* g0, const float* be0, void* out0, void* xhat0,
                 float* rstd0, int d_ff, const float* b1, const float* b2, const float* g1, const float* be1, void* H,
                 unsigned long long* relu_bits, void* out1, void* xhat1, float* rstd1, const unsigned* drop_seed,
                 unsigned drop1_salt,
                 int drop1_thresh, float drop1_scale, unsigned drop2_salt, int drop2_thresh, float drop2_scale,
                 int post_blocks, const float* bp, void* P, int ldp, void* split_work, long long split_bytes,
                 float post_kscale);
/* post_kscale (post_blocks == 3: a q | k | v projection; 0 or 1 = plain): the KEY block leaves as (acc + bias) * post_kscale,
 * scaled in fp32 before its one rounding to bf16 - pass scale * log2(e) and hand the result to st_attn_fwd / st_attn_bwd /
 * st_attn_probs with k_prescaled = 1 (below). */

/* The same chain at d_model = 512 (BASELINE config 3's encoder; csrc/st_rowchain_pipe512.cuh): A, R, out0, xhat0, out1, xhat1 are
 * [M, 512] (leading dimension 512 for the outputs), the weight stream holds 256 x 256 blocks in the order
 *   Wo: (h, j) -> 2 h + j  |  per hidden chunk c of 256: W1 (c, j = 0, 1), W2 (h = 0, 1; c)  |  per 256-column block u of the
 *   projection: (u, j = 0, 1)          (h: output-column half, j: input-column half of the 512-wide matrices)
 * - st_amd.chains.encoder512_blocks spells it.  PRE and FFN are both required; post_blocks is 0 or 6 (a q | k | v projection
 * to 1,536 columns, key columns 512 .. 1023 scaled by post_kscale as st_row_chain does).  64-row workgroups; relu_bits:
 * st_row_chain512_mask_words(M, d_ff) words.  n_blocks = 4 + 4 d_ff / 256 + 2 post_blocks. */
int st_row_chain512(st_stream_t stream, int M, const void* wfrag, int n_blocks, int next_blocks, float eps, const void* A, int lda,
                    const void* R, int ldr, const float* bo, const float* g0, const float* be0, void* out0, void* xhat0,
                    float* rstd0, int d_ff, const float* b1, const float* b2, const float* g1, const float* be1, void* H,
                    unsigned long long* relu_bits, void* out1, void* xhat1, float* rstd1, const unsigned* drop_seed,
                    unsigned drop1_salt, int drop1_thresh, float drop1_scale, unsigned drop2_salt, int drop2_thresh,
                    float drop2_scale, int post_blocks, const float* bp, void* P, int ldp, float post_kscale);
int st_row_chain512_mask_words(int M, int d_ff);
/* ... and its backward (csrc/st_rowchain_pipe512_bwd.cuh; arguments as st_row_chain_bwd with 512-wide row matrices, HEAD + FFN +
 * TAIL all required; head_blocks 0 or 6; 8 heads of 64 columns: delta [8, M]).  The stream holds the TRANSPOSED blocks in the order
 *   (h, u) -> 6 h + u of Wp [1536, 512]  |  per hidden chunk c: W2^T (j = 0, 1), W1^T (h = 0, 1)  |  Wo^T (h, j) -> 2 h + j
 * (st_amd.chains.encoder512_blocks_bwd).  n_blocks = 2 head_blocks + 4 d_ff / 256 + 4. */
int st_row_chain512_bwd(st_stream_t stream, int M, const void* wfrag, int n_blocks, int next_blocks, int head_blocks, const void* dP,
                        int ldp, const void* G, int ldg, const void* xhat_a, const float* rstd_a, const float* gamma_a,
                        const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale, void* ds_a,
                        float* dgamma_a, float* dbeta_a, float* dbias_a, int d_ff, const unsigned long long* relu_bits,
                        float mask_scale, void* dH, const void* xhat_b, const float* rstd_b, const float* gamma_b, void* ds_b,
                        float* dgamma_b, float* dbeta_b, float* dbias_b, const void* O, const void* Ores, int ldo, void* dctx,
                        int lddc, float* delta);

/* relu_bits (st_row_chain: optional output, st_row_chain_bwd: input): which hidden values of the feed-forward sublayer are
 * > 0 after ReLU and dropout (the mask of SubLayers.py:25's backward), one bit per value in a layout private to the two
 * kernels; st_row_chain_mask_words(M, d_ff) 64-bit words.  The backward chain reads these instead of H. */
int st_row_chain_mask_words(int M, int d_ff);

/* The backward of a row chain in one launch; the chain's stream holds the TRANSPOSED weight blocks (st_wfrag_build table
 * entry [1] = leading dimension | 1 << 32) in the order HEAD | FFN | TAIL:
 *   HEAD  (xhat_a != NULL)  dy = sum_u dP[:, 256u..] Wp_u + G (G may be NULL; head_blocks may be 0: dy = G);  ds_a = LayerNorm-backward(dropout-backward(dy);
 *         xhat_a, rstd_a, gamma_a);  dgamma_a / dbeta_a / dbias_a accumulated atomically        (== st_gemm_lnbwd)
 *   FFN   (d_ff > 0)  dH = (ds W2) masked by H > 0, x mask_scale (== st_gemm ST_EPI_BF16_MASK);  ds_b = LayerNorm-backward(
 *         dH W1 + ds; xhat_b, rstd_b, gamma_b) and its column sums
 *   TAIL  (O != NULL)  dctx = ds Wo;  delta[h * M + i] = sum over head h's 64 columns of dctx (O + Ores)   (== ST_EPI_BF16_DELTA)
 * ds = the running gradient: ds_a after HEAD, ds_b after FFN, the input DS [M, 256] without HEAD.  Heads are 64 columns.
 * Reference lines: the backward of Attention.py:74-76,92-94 and SubLayers.py:24-28.
 * split_work / split_bytes: as st_row_chain - d_ff / 256 workgroups per row block at decoder-sized M (HEAD replicated, one
 * chunk of the hidden dimension each, the last arriver adds the partial dy in chunk order and runs the second LayerNorm
 * backward and TAIL). */
int st_row_chain_bwd(st_stream_t stream, int M, const void* wfrag, int n_blocks, int next_blocks, int head_blocks,
                     const void* dP, int ldp, const void* G, int ldg, const void* xhat_a, const float* rstd_a,
                     const float* gamma_a, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale,
                     void* ds_a, float* dgamma_a, float* dbeta_a, float* dbias_a, const void* DS, int d_ff,
                     const unsigned long long* relu_bits, float mask_scale, void* dH, const void* xhat_b, const float* rstd_b, const float* gamma_b, void* ds_b,
                     float* dgamma_b, float* dbeta_b, float* dbias_b, const void* O, const void* Ores, int ldo, void* dctx,
                     int lddc, float* delta, void* split_work, long long split_bytes, float* colsum_ws, long long colsum_bytes);
/* colsum_ws (optional; ABI 4): encoder-sized launches (HEAD + FFN + TAIL at M > 8192: st_row_chain_bwd_colsum_rows(..) > 0 workgroups) write
 * their six column sums per workgroup - [rows][dgamma_a, dbeta_a, dbias_a, dgamma_b, dbeta_b, dbias_b][256] floats, every float of
 * it - instead of adding them atomically (251 workgroups adding to the same 768 floats queue for ~4 us per LayerNorm: 66 -> 58 us per
 * launch at 24,060 rows); the d* pointers are then left alone and st_colsum_fold adds the rows to them: any time before the
 * gradients are read, several workspaces per launch.  Other launches ignore colsum_ws and add atomically. */
int st_row_chain_bwd_colsum_rows(int M, int has_head, int d_ff, int has_tail);
int st_colsum_fold(st_stream_t stream, int n, const float* const* ws, const int* rows, float* const* dst /* 6 n, NULL = skip */);

/* LayerNorm backward: dx, and atomically accumulated dgamma / dbeta / dbias
 * (dbias = column sum of dx = bias gradient of the Linear feeding the LN).
 * mask (optional bf16 [M,N]): dx is zeroed where mask <= 0 (front-end ReLU,
 * Models.py:28-33) and scaled by mask_scale elsewhere (1/(1-p) of the dropout
 * between that ReLU and the LN).  drop_*: the forward dropped the LayerNorm
 * OUTPUT (st_gemm_ln drop_where = 2): dy is masked / rescaled first. */
int st_ln_bwd(st_stream_t stream, const void* dy, int lddy, const void* xhat, const float* rstd, const float* gamma,
              const void* mask, void* dx, int lddx, float* dgamma, float* dbeta, float* dbias, int M, int N,
              const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale, float mask_scale);

/* Rows per work-list tile of the attention kernel that serves a problem of
 * this shape (the kernels are chosen by shape: 128-row workgroups, except
 * the forward for <= 64 queries against >= 256 keys with 64-wide heads -
 * the decoder-encoder attention - which runs 32-query workgroups).
 * which: 0 = st_attn_fwd (query tiles), 1 = st_attn_bwd work_q (query tiles),
 * 2 = st_attn_bwd work_k (key tiles). */
int st_attn_tile_rows(int which, int d_k, int max_q, int max_k, int causal);

/* Fused masked attention forward: softmax(Q K^T * scale, keys >= k_len[b]
 * and (causal) keys > query masked) V, per head; replaces Attention.py:82-90
 * and the dense masks of Utils.py:41-70.  lse (f32 [H, q_rows_total], log2
 * domain) is saved for the backward.  d_k in {32, 64, 128}.
 * work (optional, device int32 [n_work]): the (utterance, query tile) pairs
 * to run, packed (b << 16) | tile and sorted by decreasing cost, so the
 * ragged batch is list-scheduled longest-first; a tile is
 * st_attn_tile_rows(0, ...) query rows; NULL = enumerate every tile of every
 * utterance up to max_q.  drop_*: dropout on the attention
 * probabilities (Attention.py:89); pass the same values to st_attn_bwd.
 * k_prescaled: K holds scale * log2(e) * (key projection) - scaled once, in the fp32 epilogue of the GEMM that produced it
 * (st_row_chain's post_kscale) - so q . K is the score in the log2 domain and no kernel multiplies per score; forward,
 * backward and st_attn_probs of one sublayer must agree on it.  Results are those of k_prescaled = 0 on the unscaled keys
 * up to the one rounding of K; dK (st_attn_bwd) is the gradient of the UNSCALED key projection either way. */
int st_attn_fwd(st_stream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O,
                int ldo, void* Ores, float* lse, const int* q_off, const int* q_len, const int* k_off, const int* k_len, int B,
                int H, int d_k, int max_q, int max_k, int q_rows_total, int causal, float scale, const int* work,
                int n_work, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale, int k_prescaled);

/* The decoder-encoder attention of Layers.py:41 TOGETHER with what stands between it and the self-attention in front of it:
 * cur = LN(ctxA Wo^T + bo + R) (the self-attention's output_linear + residual + layernorm, Attention.py:92-94; out0 / xhat0 /
 * rstd0 as st_row_chain's PRE writes them) and q = cur Wq^T + bq (Attention.py:74; Qout [rows, 256]), then st_attn_fwd on
 * that q - one launch instead of st_row_chain(PRE + POST) + st_attn_fwd.  wfrag: the two-block fragment stream of that chain
 * (st_wfrag_build), n_blocks == 2, next_blocks as st_row_chain.  Only for the shapes st_attn_f1_applicable reports
 * (d_model 256, d_k 64, max_q <= 64, max_k >= 256: the few-queries forward); otherwise -10 - callers then issue the two
 * launches.  Results are those of the two launches (same arithmetic, same order). */
int st_attn_f1_applicable(int d_model, int d_k, int max_q, int max_k);
int st_attn_f1_fwd(st_stream_t stream, const void* ctxA, int lda, const void* R, int ldr, const void* wfrag, int n_blocks,
                   int next_blocks, float eps, const float* bo, const float* g0, const float* be0, void* out0, void* xhat0,
                   float* rstd0, const float* bq, void* Qout, int ldq, const void* K, int ldk, const void* V, int ldv, void* O,
                   int ldo, void* Ores, float* lse, const int* q_off, const int* q_len, const int* k_off, const int* k_len,
                   int B, int H, int d_k, int max_q, int max_k, int q_rows_total, float scale, const int* work, int n_work,
                   const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale);

/* st_attn_f1_fwd with the decoder's causal SELF-attention (Layers.py:39) computed in front of the chain stage, by the same launch:
 * qkv_q / qkv_k / qkv_v = the layer's q | k | v projection (leading dimension ld_qkv, head h at columns h * 64; queries and keys
 * are the utterance's own <= 64 target positions: q_off / q_len), Os / Oress / lses = what st_attn_fwd(causal = 1) writes
 * for it (Os and lses bit for bit, Oress - may be NULL - to its own precision), its dropout (Attention.py:89) with its own salt / threshold / scale on the shared device seed.
 * The context goes from there straight into output_linear + LayerNorm + q projection and the decoder-encoder attention: the
 * three launches st_attn_fwd + st_row_chain + st_attn_fwd as one.  Same shapes as st_attn_f1_fwd (-10 otherwise); H <= 8. */
int st_attn_sf1_fwd(st_stream_t stream, const void* qkv_q, const void* qkv_k, const void* qkv_v, int ld_qkv, void* Os, void* Oress,
                    int ldos, float* lses, unsigned sdrop_salt, int sdrop_thresh, float sdrop_scale, const void* R, int ldr,
                    const void* wfrag, int n_blocks, int next_blocks, float eps, const float* bo, const float* g0,
                    const float* be0, void* out0, void* xhat0, float* rstd0, const float* bq, void* Qout, int ldq, const void* K,
                    int ldk, const void* V, int ldv, void* O, int ldo, void* Ores, float* lse, const int* q_off,
                    const int* q_len, const int* k_off, const int* k_len, int B, int H, int d_k, int max_q, int max_k,
                    int q_rows_total, float scale, const int* work, int n_work, const unsigned* drop_seed, unsigned drop_salt,
                    int drop_thresh, float drop_scale);

/* Attention backward (autograd of Attention.py:82-90): dQ, dK, dV from
 * Q, K, V, O, dO, lse; `delta` is f32 [H, q_rows_total] scratch.  Two kernels:
 * parts & 1 = dQ (also writes delta), parts & 2 = dK/dV (reads delta); 3 = both.
 * O == NULL: delta is an input (st_gemm ST_EPI_BF16_DELTA wrote it with dO) and parts == 3 runs as ONE launch.
 * work_q / work_k: optional work lists (see st_attn_fwd) over query tiles
 * (dQ kernel) and key tiles (dK/dV kernel) of st_attn_tile_rows(1 / 2, ...) rows. */
int st_attn_bwd(st_stream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                const void* O, int ldo, const void* dO, int lddo, const float* lse, float* delta, void* dQ, int lddq,
                void* dK, int lddk, void* dV, int lddv, const int* q_off, const int* q_len, const int* k_off,
                const int* k_len, int B, int H, int d_k, int max_q, int max_k, int q_rows_total, int causal,
                float scale, int parts, const int* work_q, int n_work_q, const int* work_k, int n_work_k,
                const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale, int k_prescaled,
                void* split_work, long long split_bytes);
/* split_work (optional; ABI 4): st_attn_bwd_split_kib(..) KiB of scratch for the merged launch (parts = 3, O = NULL) of a few-queries /
 * many-keys problem (max_q <= 64, max_k >= 256, not causal: the decoder-encoder attention) - every dQ item's key tiles are then cut over
 * up to four workgroups, fp32 partials merged by the last arriver in part order (deterministic).  Its first 16 KiB (tickets) must be
 * zero before the first launch (the kernel leaves them zero); launches on one stream may share it.  0 KiB: the shape does not split. */
int st_attn_bwd_split_kib(int B, int H, int d_k, int max_q, int max_k, int causal);

/* row_pos[off[b]+t] = t (and row_seq[...] = b if non-null), t < len[b]:
 * the per-row position the PE add needs (Embedding.py:21-29). */
int st_row_index(st_stream_t stream, const int* off, const int* len, int B, int max_len, int* row_pos, int* row_seq);

/* Padded fp32 features [B,T,F] -> bf16 row matrix (train.py:33, Models.py:42). */
int st_pack_rows(st_stream_t stream, const float* x, int B, int T, int F, const int* off, const int* len, void* out);
/* Feature front-end fused with the ragged pack (reference Dataset.py: cmvn :89-92, concat_frame :121-143,
 * subsampling :145-153): raw padded fp32 features x [B,T,F] (in_len[b] valid frames) -> bf16 row matrix
 * out [sum(out_len), ld >= F*(1+left+right)]; output row r of utterance b is frame r*interval with its left /
 * right context blocks exactly where the reference writes them (right blocks indexed with the RIGHT width,
 * :139-141; right <= left).  stats: per-utterance Kaldi CMVN statistics f32 [B, 2, F+1] or NULL. */
int st_feat_stack(st_stream_t stream, const float* x, int B, int T, int F, const int* in_len, const float* stats,
                  int left, int right, int interval, const int* out_off, const int* out_len, int max_out_len,
                  void* out, int ld);
/* bf16 row matrix -> padded fp32 [B,T,D], zero past len[b] (Encoder/Decoder return value). */
int st_unpack_rows(st_stream_t stream, const void* x, int ld, int B, int T, int D, const int* off, const int* len,
                   float* out);
/* Backward of st_unpack_rows: padded fp32 gradient -> bf16 row matrix. */
int st_pack_grad(st_stream_t stream, const float* g, int B, int T, int D, const int* off, const int* len, void* out,
                 int ld);

/* Decoder input: out[off[b]+t] = emb[tok[b,t]] + pe[t]  (Models.py:84,87 with repair R3).  emb has V rows; a token id
   outside [0, V) traps the kernel (nn.Embedding's device assert). */
int st_embed_pe_fwd(st_stream_t stream, const long long* tok, int B, int L, const float* emb, int V, const float* pe,
                    int D, const int* off, const int* len, void* out);
/* demb[tok] += dy (f32 atomics); row pad_idx receives nothing (Models.py:74); ids outside [0, V) trap. */
int st_embed_bwd(st_stream_t stream, const long long* tok, int B, int L, const void* dy, int ld, int D,
                 const int* off, const int* len, int pad_idx, float* demb, int V);

/* One beam-search step's decoder input (Models.py:84,87): out bf16 [n, D] = emb[tokens[i]] + pe[*step] (emb f32 [V, D],
   pe f32 [>= *step + 1, D], *step a device scalar); ids outside [0, V) trap. */
int st_embed_step(st_stream_t stream, const long long* tokens, const float* emb, int V, const float* pe, const long long* step,
                  void* out, int n, int D);

/* Decode-shaped self-attention of one beam-search step (Decode.py:96-98 with a KV cache): one query per hypothesis.
   qkv bf16 [n, ldq] = this step's q | k | v (3 * H * 64 columns); cache bf16 [n][S][2 * H * 64] of ONE layer; appends k | v
   at position *step (device scalar) and writes ctx [n, ldc] = softmax(q K^T * scale) V over positions 0 .. *step.
   d_k = 64, S <= 128.  `anc`: NULL, or the lineage table int32 [n][S] that st_beam_advance maintains - position p < *step of
   hypothesis i is then read from cache row anc[i][p] (the slot the ancestor that produced it sat in), the step's own K | V
   still goes to row i: the cache rows never move and st_cache_reorder is not needed.  Every entry of the table must be a
   valid row at all times (initialise it to anc[i][p] = i). */
int st_decode_self_attn(st_stream_t stream, const void* qkv, int ldq, void* cache, const long long* step, const int* anc,
                        void* ctx, int ldc, int n, int S, int H, int d_k, float scale);

/* Beam.advance (Beam.py:43-74) for all B utterances in one launch, from the raw vocabulary logits f32 [B * beam, ldl] (V
   valid columns): log-softmax per hypothesis (Decode.py:102), the `beam` best of score + log-probability over beam x V
   (best first; Beam.py:53-57), back-pointer = flat / V, token = flat % V (Beam.py:63-66), done once the best hypothesis emits
   `eos` (Beam.py:70-72; a done utterance is frozen: identity back-pointers, scores / tokens unchanged).  Device state,
   updated in place: scores f32 [B, beam], tokens i64 [B * beam], done u8 [B], lengths i64 [B], the trellis hist_scores f32 /
   back i64 / toks i64 [S, B, beam] at row *step (device scalar), order i64 [B * beam] = the cache rows the hypotheses
   inherit (input of st_cache_reorder).  beam <= 16.  `work`: NULL, or ZEROED device scratch of B * beam * beam + 1 + B 8-byte words - with
   it (and V <= 5120) the step runs over B * beam workgroups instead of one per utterance (the `beam` best of every hypothesis
   row; the wave of an utterance's row that finishes last merges them - a ticket per utterance, no second launch); same
   results.  Only with `work`:
   `anc`: NULL, or the lineage table int32 [B * beam][S] of st_decode_self_attn (S <= 128): the hypothesis placed in slot s
   takes over positions 0 .. *step - 1 of its origin's row of the table and gets the origin's slot as position *step.
   `step_next`: NULL, or `step` itself: the launch also advances the device step counter, *step += 1, once every utterance
   has used the old value.
   `x_next`: NULL, or bf16 [B * beam, D]: the NEXT step's decoder input, bf16(emb[token] + pe[*step + 1]) for the tokens
   just chosen (st_embed_step's arithmetic; emb f32 [emb_rows, D], pe f32 [pe_rows, D]; skipped when *step + 1 == pe_rows). */
int st_beam_advance(st_stream_t stream, const float* logits, int ldl, int V, int beam, int B, const long long* step, int eos,
                    float* scores, long long* tokens, unsigned char* done, long long* lengths, float* hist_scores,
                    long long* back, long long* toks, long long* order, void* work, int* anc, int S, long long* step_next,
                    const float* emb, int emb_rows, const float* pe, int pe_rows, void* x_next, int D);

/* Beam-search decode (transformer/Decode.py with a KV cache): cache bf16 [L][n][S][W]; for every layer, position
   t <= *step (device scalar) and utterance (beam consecutive hypothesis rows), row u*beam+s <- row order[u*beam+s]
   (device int64 [n], a row of the same utterance: Beam.py:65 back-pointers), in place. */
int st_cache_reorder(st_stream_t stream, void* cache, const long long* order, const long long* step, int L, int n, int S,
                     int W, int beam);

/* fp32 master parameters -> bf16 shadow, n a multiple of 8. */
int st_cast_bf16(st_stream_t stream, const float* src, void* dst, long long n);

/* clip_grad_norm_ + Adam over the flat fp32 parameter buffer in one pass (train.py:45-46, transformer/Optim.py:
 * Adam(betas = (beta1, beta2), eps)): g *= min(1, max_norm / (*gnorm + 1e-6)) (gnorm NULL: no clipping), then the
 * update of torch.optim.Adam at step count *step (already incremented) and learning rate *lr - all three device
 * scalars.  n: elements, a multiple of 4; p, g, m (exp_avg), v (exp_avg_sq): fp32 [n].
 * grad_scale: the buffer holds gradient / grad_scale - 1 / world behind a SUMMING all-reduce (train_multi.py:161-163:
 * Horovod averages), 1 otherwise; the factor is applied together with the clip coefficient (g is left holding the
 * clipped, averaged gradient), *gnorm must be the norm of the scaled gradient (st_grad_norm with the same grad_scale). */
int st_adam_clip(st_stream_t stream, long long n, float* p, float* g, float* m, float* v, const float* lr,
                 const float* step, const float* gnorm, float max_norm, float beta1, float beta2, float eps,
                 float grad_scale);

/* Zero the unassigned tail rows of a list of row matrices in one launch (packed bucket layouts: the rows behind the
 * batch's last utterance belong to nobody; the kernels never write them, so they must read as zeros).  table (device,
 * int64 [n_max][4]): base address, bytes per row (% 16 == 0), capacity in rows, address of a device int32 holding the
 * number of valid rows; entries with a null base address are skipped. */
int st_zero_tails(st_stream_t stream, const long long* table, int n_max);
/* zero_grad of a flat buffer (train.py:37; ABI 4): `bytes` (a multiple of 16) at the 16-byte aligned `ptr` set to zero with wide stores. */
int st_zero(st_stream_t stream, void* ptr, long long bytes);

/* total_norm of clip_grad_norm_ (train.py:45) over the flat fp32 gradient buffer g [n] (n % 4 == 0) as one launch:
 * *gnorm = grad_scale * ||g||_2 (partials added in a fixed order, fp64), and - when step is not NULL - *step += 1, the optimiser's
 * step count st_adam_clip then reads.  scratch: st_grad_norm_blocks() + 1 floats owned by the caller, the last one
 * zero before the first call (the kernel leaves it zero). */
int st_grad_norm_blocks(void);
int st_grad_norm(st_stream_t stream, const float* g, long long n, float* scratch, float* gnorm, float* step,
                 float grad_scale);

/* Cross-entropy over ragged logits rows (train.py:40,120: nn.CrossEntropyLoss(ignore_index = 0), mean over the
 * non-ignored tokens).  logits f32 [R, ldl] (V valid columns; padding columns holding -1e30 may be counted as valid),
 * target i64 [R], or - with target_index (i64 [R]) - any i64 vector read as target[target_index[r]] (the padded
 * ground truth of train.py:40 addressed through the ragged rows' positions: no gather launch).  st_ce_fwd: lse[r] = logsumexp(logits[r, :V]) (f32 [R], kept for the backward); row_loss[r] (f32 [R],
 * scratch) = lse[r] - logits[r, target[r]] on the non-ignored rows; sums (f32 [3]): [0] = their sum, [1] = their number,
 * [2] = the loss sums[0] / sums[1].  st_ce_bwd: dlogits (bf16 [R, ldd], ldd % 8 == 0, columns >= V zero) = (softmax - onehot) * *grad_out /
 * sums[1] on the non-ignored rows, 0 elsewhere - the operand of the vocabulary projection's backward GEMMs. */
int st_ce_fwd(st_stream_t stream, const float* logits, int ldl, int R, int V, const long long* target,
              const long long* target_index, int ignore_index, float* lse, float* row_loss, float* sums);
int st_ce_bwd(st_stream_t stream, const float* logits, int ldl, int R, int V, const long long* target,
              const long long* target_index, int ignore_index, const float* lse, const float* sums, const float* grad_out,
              void* dlogits, int ldd);

/* CTC head of the joint CTC + attention objective (BASELINE config 4; transformer/Loss.py:CTCAttentionLoss - the reference's
 * train_attn_and_ctc.py is empty, the head is the standard hybrid one).  ctc_loss itself stays in PyTorch-ROCm; these two
 * kernels stand where the [T, B, V] log-softmax tensor and its gradient would be.  logits f32 [R, ldl]: the ragged rows of
 * the encoder-side vocabulary projection (V valid columns; padding columns at -1e30 may be counted); rowmap i64 [R]: row r is
 * frame t of utterance b with rowmap[r] = b * T + t (negative: the row belongs to nobody); cols i32 [B, C]: the vocabulary
 * column of class k of utterance b (class 0 = blank, classes 1.. = the utterance's labels as ctc_loss is given them).
 * st_ctc_gather: lse[r] = logsumexp(logits[r, :V]); lp (f32 [B, T, C]) [b][t][k] = logits[r][cols[b][k]] - lse[r] on the
 * rows that exist (the rest of lp is left untouched: ctc_loss does not read frames past an input length).
 * st_ctc_dlogits: dlogits (bf16 [R, ldd], ldd % 8 == 0) = *grad_out * (roww[b] * softmax(logits[r]) everywhere, columns >= V
 * zero; then column scat[b][k] (i32 [B, C], -1 = none) overwritten with gsmall[b][t][k]) - gsmall (f32 [B, T, C]) being
 * ctc_loss's gradient with respect to lp (which already contains the softmax term at those columns, Graves eq. 16) and
 * roww[b] the weight of utterance b's loss in the reduction (0 for an utterance whose loss is infinite: zero_infinity). */
int st_ctc_gather(st_stream_t stream, const float* logits, int ldl, int R, int V, const long long* rowmap, int T, const int* cols,
                  int C, float* lse, float* lp);
int st_ctc_dlogits(st_stream_t stream, const float* logits, int ldl, int R, int V, const float* lse, const long long* rowmap, int T,
                   const float* roww, const int* scat, int C, const float* gsmall, const float* grad_out, void* dlogits, int ldd);

/* Attention probabilities of ONE attention sublayer, materialised (reference transformer/Attention.py:89,96: the `attns`
 * MultiHeadAttention.forward returns; Models.py:53-54,107-109 collect them under return_attns): P (f32 [B, H, Lq, Lk],
 * dense) = softmax over keys of scale * Q K^T with keys >= k_len[b] (and, when causal, keys > the query) masked out -
 * zeros there and in the rows of query positions >= q_len[b].  Q, K: bf16 row matrices (utterance b owns rows
 * off[b] .. off[b] + len[b] - 1, head h columns h * d_k ..), d_k % 8 == 0.  A diagnostic path: the fused attention kernels
 * never write these maps. */
int st_attn_probs(st_stream_t stream, const void* Q, int ldq, const void* K, int ldk, float* P, const int* q_off,
                  const int* q_len, const int* k_off, const int* k_len, int B, int H, int d_k, int Lq, int Lk, int causal,
                  float scale, int k_prescaled);

/* Attention under an ARBITRARY dense mask / with keys and values from different tensors - the general form of
 * Attention.py:82-90 that no reference call site uses (they all pass a key-padding or key-padding | causal mask and k == v,
 * which st_attn_fwd serves through lengths): a slow path (one workgroup per (utterance, head, query) / key, fp32 FMAs) that
 * closes the module boundary.  Padded layout: utterance b owns query rows b Lq .. of Q / O / dO / dQ and key rows b Lk .. of
 * K / V / dK / dV; head h = columns h d_k ..; d_k % 8 == 0.  mask: uint8 [B, Lq, Lk], nonzero = masked (NULL: none); a row
 * with every key masked yields a zero context and zero gradients (the reference: NaN).  lse / delta: f32 [H, B Lq] (natural
 * log); P (nullable, f32 [B, H, Lq, Lk]): the probabilities before dropout (Attention.py:96).  drop_*: as everywhere. */
int st_attn_dense_fwd(st_stream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                      const unsigned char* mask, void* O, int ldo, float* lse, float* P, int B, int H, int d_k, int Lq, int Lk,
                      float scale, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale);
int st_attn_dense_bwd(st_stream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                      const unsigned char* mask, const void* dO, int lddo, const float* lse, float* delta, void* dQ, int lddq,
                      void* dK, int lddk, void* dV, int lddv, int B, int H, int d_k, int Lq, int Lk, float scale,
                      const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale);

/* Hardware probes used by tests to pin the MFMA / transposing-LDS-read layouts. */
int st_probe_tr16(st_stream_t stream, const void* in, void* out);
int st_probe_mfma(st_stream_t stream, const void* A, const void* Bt, float* D);

/* The shader clock this part sustains under dense matrix work (the normaliser of bench.py's MFMA roofline: the nominal 2.4 GHz is
 * not what an MI355X holds under chip-wide MFMA load).  n_wg workgroups of 256 threads (give one per CU) each run `iters` rounds of
 * four independent 32 x 32 x 16 bf16 MFMAs per wave on non-trivial operands and leave {shader-clock ticks (s_memtime), wall ticks
 * (s_memrealtime, 100 MHz)} of wave 0 in out[2 wg], out[2 wg + 1]: MHz = 100 ticks / wall.  ~2 ms at iters = 30000. */
int st_clock_probe(st_stream_t stream, long long* out, int n_wg, int iters);

#ifdef __cplusplus
}
#endif
#endif /* ST_HIP_H */
