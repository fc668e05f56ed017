/*
 * dynamorph_hip.h -- C ABI of libdynamorph_hip.so (MI355X / gfx950 only).
 *
 * The reference (mehta-lab/dynamorph) has NO native code and no FFI: its
 * VQ-VAE latent-encoding path is a sequence of PyTorch ATen ops issued from
 *   HiddenStateExtractor/vq_vae.py:52-84   VectorQuantizer.forward
 *   HiddenStateExtractor/vq_vae.py:276-298 VQ_VAE.enc / .dec (nn.Sequential)
 *   HiddenStateExtractor/vq_vae.py:300-338 VQ_VAE.forward (losses)
 *   run_training.py:404-408, :485          backward + Adam
 *   pipeline/patch_VAE.py:445-452          per-sample enc -> vq loop.
 * Each entry point below replaces the ATen op(s) named in its comment, so the
 * binding a reference maintainer would add is "call this instead of the
 * nn.Module's ATen kernel" (ctypes stubs: INTEGRATION.md).
 *
 * Conventions
 *  - every pointer is DEVICE memory owned by the caller (PyTorch allocates all
 *    tensors and workspaces); the library never allocates, frees or retains.
 *  - tensors are fp32, contiguous NCHW; indices are int64.
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*)
 *    and returns; no hidden synchronisation, no global mutable state.
 *  - return 0 = ok; < 0 = argument/shape error detected on the host before any
 *    launch (message via dm_last_error()); > 0 = hipError_t of a failed launch.
 */
#ifndef DYNAMORPH_HIP_H
#define DYNAMORPH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM_VERSION 128

/* ---- on-load operand transform ------------------------------------------
 * A kernel never reads a bare activation: BatchNorm-apply, ReLU and the
 * BatchNorm-backward formula are folded into the load of the consuming kernel.
 *   IDENT        v = p0
 *   RELU         v = max(p0, 0)
 *   AFFINE       v = c0*p0 + c2                  (BatchNorm apply)
 *   AFFINE_RELU  v = max(c0*p0 + c2, 0)          (BatchNorm + ReLU)
 *   AFFINE2      v = c0*p0 + c1*p1 + c2          (BatchNorm backward: p0 = dy, p1 = conv output)
 * coef is [C][4] floats (c0,c1,c2,unused), or [B][C][4] when coef_bstride != 0
 * (per-sample BatchNorm statistics = pipeline/patch_VAE.py batch-of-one calls). */
enum { DM_LOAD_IDENT = 0, DM_LOAD_RELU = 1, DM_LOAD_AFFINE = 2, DM_LOAD_AFFINE_RELU = 3, DM_LOAD_AFFINE2 = 4 };

typedef struct dm_operand {
    const float *p0;
    const float *p1;
    const float *coef;
    int64_t coef_bstride;   /* floats between consecutive samples in coef; 0 = shared */
    int32_t mode;           /* DM_LOAD_* */
    int32_t ones_channel;   /* 1: append a synthetic channel that is 1 inside the image, 0 in the zero padding */
} dm_operand;

/* ---- weight view ----------------------------------------------------------
 * Logical weight W[n][c][ky][kx] of the GEMM a kernel runs, addressed as
 *   w[off + n_ch*sn + c*sc + ky*sky + kx*skx]
 * so one kernel serves forward convs (PyTorch layout [co][ci][k][k]), their
 * data gradients (transposed / flipped views) and ConvTranspose2d layouts
 * ([ci][co][k][k]) without materialising a re-laid-out copy. */
typedef struct dm_weight_view {
    const float *w;
    int64_t off, sn, sc, sky, skx;
    float *scratch;          /* device scratch of >= dm_conv*_scratch_floats(...) floats, or NULL.  Channel counts without a */
    int64_t scratch_floats;  /* register-resident kernel run an implicit GEMM that first re-lays the weights into it; without
                                scratch those shapes take the (much slower) generic kernel.  Contents are don't-care. */
} dm_weight_view;

/* ---- epilogue ---------------------------------------------------------------
 *   v = acc + bias[n]; if relu v = max(v,0);
 *   if mask.p0: v = (load(mask) > 0) ? v : 0          (ReLU backward)
 *   if resid:   v += resid                             (residual-branch join)
 *   out = v;  if stats: partial sums (sum v, sum v*q), q = stat_q ? stat_q : v,
 *   written as doubles to stats[workgroup][n][2]; dm_conv*_num_blocks(..., stats_per_tile) says how many
 *   slabs that is (reduced by dm_bn_finalize / dm_bn_backward_finalize / dm_sum_slabs: deterministic). */
typedef struct dm_epilogue {
    const float *bias;
    const float *bias_border; /* dm_conv4x4s2 only, [3][3][NOUT] (row class, column class, channel) with classes
                                 first / interior / last output row or column: replaces `bias` by a per-position bias
                                 (the enc.0 bias seen through enc.1's zero padding, dm_e1_compose); NULL: plain bias */
    int32_t relu;
    int32_t stats_per_tile;   /* 1: one workgroup per tile, every stats slab holds that tile's sums (per-sample
                                 BatchNorm statistics); 0: persistent workgroups, most slabs are zero */
    dm_operand mask;
    const float *resid;
    const float *stat_q;
    double *stats;
} dm_epilogue;

const char *dm_last_error(void);
int dm_version(void);

/* ===== VectorQuantizer (vq_vae.py:52-116) ==================================== */

/* Bytes of workspace dm_vq_forward needs for a K x D codebook. */
size_t dm_vq_workspace_bytes(int K, int D);

/* Replaces vq_vae.py:65-82: distances + argmax(-dist) + embedding gather +
 * straight-through value + squared-error sum + code histogram.
 *   z (B,D,H,W); codebook (K,D); idx (B,H,W) int64 [may be NULL];
 *   out (B,D,H,W) = z + (q - z) [may be NULL];
 *   sse_slabs: one double per workgroup (dm_vq_num_blocks of them);
 *   hist: K int32 counters (zeroed by the call).
 * Distances are summed in the reference's order (blocks of 16 along D,
 * sequential, no FMA) so indices are bit-identical to the CPU path. */
int dm_vq_num_blocks(int64_t positions);
int dm_vq_forward(const float *z, const float *codebook, int64_t *idx, float *out,
                  double *sse_slabs, int32_t *hist, int B, int D, int K, int H, int W,
                  void *workspace, size_t workspace_bytes, void *stream);

/* The same with the kernel chosen by the caller (tests and measurements; dm_vq_forward = DM_VQ_AUTO):
 *   DM_VQ_EXACT  every distance in the reference's arithmetic (3*K*D non-fused operations per position);
 *   DM_VQ_MFMA   |e|^2 - 2 z.e on the matrix pipe as a filter, the exact arithmetic only for positions whose two best
 *                scores are closer than the proven error bound (csrc/vq.hip) -- identical indices; needs
 *                embedding_dim 8/16/32/64, H*W a multiple of 64 and 16-byte aligned tensors (else a negative return);
 *   DM_VQ_BF16   the same filter on v_mfma_f32_16x16x32_bf16 with both operands split into a bf16 head and remainder
 *                (a quarter of the matrix cycles, a wider proven tolerance, more positions on the exact path) -- identical
 *                indices; embedding_dim 16/32/64;
 *   DM_VQ_AUTO   DM_VQ_BF16 where it applies (DM_VQ_FILTER=f32 in the environment: DM_VQ_MFMA), else DM_VQ_MFMA, else
 *                DM_VQ_EXACT.
 * After a filtered call WITH hist != NULL the first int32 of `workspace` holds the number of positions that took the exact
 * path.  With hist = NULL and at most 64 codes the count stays in column 64 of the per-workgroup counter rows (it is summed
 * into workspace[0] only by the counter reduction that hist != NULL launches) and workspace[0] reads 0. */
enum { DM_VQ_AUTO = 0, DM_VQ_EXACT = 1, DM_VQ_MFMA = 2, DM_VQ_BF16 = 3 };
int dm_vq_forward_variant(const float *z, const float *codebook, int64_t *idx, float *out,
                          double *sse_slabs, int32_t *hist, int B, int D, int K, int H, int W,
                          void *workspace, size_t workspace_bytes, int variant, void *stream);

/* dm_vq_forward with the encoder's LAST RESIDUAL JOIN in its load path (ResidualBlock.forward, vq_vae.py:222-224, followed
 * by VectorQuantizer.forward, vq_vae.py:52-84): the latents are formed on the way in,
 *     z = fma(coef[d][0], rb, coef[d][2]) + h_in        (dm_apply's arithmetic: BatchNorm of the block's last conv + skip)
 * quantised exactly as dm_vq_forward quantises a stored z, and written to z_out (the backward pass, the time-matching
 * term and the callers of `model.enc` need them) -- one launch and one round trip of the latents fewer than dm_apply +
 * dm_vq_forward.  coef: the [D][4] table of dm_bn_finalize (batch statistics).  Built for the one-launch form: at most 64
 * codes, embedding_dim 16 (the reference's num_hiddens), H*W a multiple of 64 (dm_vq_forward_join_supported; else use dm_apply + dm_vq_forward).
 * z_out must not alias rb or h_in.  hist NULL: counters stay in the workspace (dm_vq_loss_finalize), as in dm_vq_forward. */
int dm_vq_forward_join_supported(int D, int K, int H, int W);
int dm_vq_forward_join(const float *rb, const float *h_in, const float *coef, float *z_out, const float *codebook,
                       int64_t *idx, float *out, double *sse_slabs, int32_t *hist, int B, int D, int K, int H, int W,
                       void *workspace, size_t workspace_bytes, void *stream);

/* Measurement entry (bench.py, tools/vqbench.py): the same launches, the distance / argmin kernel `repeats` times in a row
 * between ONE preparation and ONE counter reduction, so that T(repeats = n + 1) - T(repeats = 1) times n launches of that
 * kernel alone, with events on the caller's stream and no synchronisation inside the library.  idx / out / sse_slabs are the
 * single call's (the kernel overwrites them); hist and the re-check counter accumulate over the repeats. */
int dm_vq_forward_repeat(const float *z, const float *codebook, int64_t *idx, float *out,
                         double *sse_slabs, int32_t *hist, int B, int D, int K, int H, int W,
                         void *workspace, size_t workspace_bytes, int variant, int repeats, void *stream);

/* vq_vae.py:105-116 decode_inputs: q[b,d,h,w] = codebook[idx[b,h,w], d]. */
int dm_vq_decode(const int64_t *idx, const float *codebook, float *q,
                 int B, int D, int K, int H, int W, void *stream);

/* Reduces the slabs/histogram to the three scalars the module returns
 * (vq_vae.py:74-82): scalars[0] = loss = mse + cc*mse, [1] = perplexity, [2] = mse. */
int dm_vq_finalize(const double *sse_slabs, int nslabs, const int32_t *hist, int K,
                   int64_t positions, int D, float commitment_cost, float *scalars, void *stream);

/* dm_vq_forward with hist = NULL leaves the code counters in the workspace's replicas (no reduction launch); this entry
 * then produces the training step's four scalars -- (recon, commitment, total, perplexity), the arithmetic of
 * dm_vq_finalize followed by dm_loss_finalize -- in ONE launch from the sse slabs, the workspace of that very
 * dm_vq_forward call and the reconstruction-loss slabs (vq_vae.py:74-82, 320-323, 333-336). */
int dm_vq_loss_finalize(const double *sse_slabs, int nslabs, const void *workspace, int K, int D,
                        int64_t positions, float commitment_cost, const double *loss_slabs, int nloss,
                        int64_t count, float weight_recon, float weight_commitment, float *scalars_out, void *stream);
/* ... with the pairwise time-matching term in the same launch: tm_slabs = the loss_slabs of dm_time_matching_forward (ntm
 * pairs of doubles); scalars_out has FIVE entries: (recon, commitment, total + weight_matching * tm, perplexity, tm)
 * (vq_vae.py:324-336; vae.py:456-470). */
int dm_vq_loss_finalize_tm(const double *sse_slabs, int nslabs, const void *workspace, int K, int D, int64_t positions,
                           float commitment_cost, const double *loss_slabs, int nloss, int64_t count, float weight_recon,
                           float weight_commitment, const double *tm_slabs, int ntm, float weight_matching,
                           float *scalars_out, void *stream);

/* Autograd of vq_vae.py:71-76 for upstream (g_out, g_loss):
 *   dz = g_out + g_loss*2*cc*(z-q)/N   [g_out may be NULL = 0]
 *   dw[k] += g_loss * sum_{idx=k} 2*(q-z)/N   (dw must be zeroed by the caller)
 * g_loss is read from device memory (g_loss_dev[0]) so no host sync is needed. */
int dm_vq_backward(const float *z, const float *codebook, const int64_t *idx,
                   const float *g_out, const float *g_loss_dev, float commitment_cost,
                   float *dz, float *dw, int B, int D, int K, int H, int W, void *stream);

/* The same with the codebook gradient as per-workgroup slabs [dm_vq_backward_num_slabs(P, K, D)][K*D] instead of global
 * float atomics: dm_reduce_slabs / dm_reduce_slabs_multi adds them in a fixed order and nothing has to be zeroed
 * first.  K <= 64 with D in {16, 32, 64}, H*W % 64 == 0 and 16-byte aligned tensors: the gradient is a one-hot
 * (codes x positions) . (positions x D) product on the matrix cores, accumulated in registers and combined in wave
 * order -- bit-reproducible.  Otherwise positions are added into an LDS window with float atomics (hardware order
 * inside a workgroup: low bits may vary); codebooks above the 128 KB window (512 x 64, 4096 x 16) are processed in
 * windows of codes over grid.y. */
int dm_vq_backward_num_slabs(int64_t positions, int K, int D);   /* <= 512, fewer for large codebooks (slab tensor <= 32 MB) */
int dm_vq_backward_slabs(const float *z, const float *codebook, const int64_t *idx,
                         const float *g_out, const float *g_loss_dev, float commitment_cost,
                         float *dz, float *dw_slabs, int B, int D, int K, int H, int W, void *stream);

/* ===== convolutions (nn.Conv2d / nn.ConvTranspose2d, vq_vae.py:276-298) ======= */

/* 4x4, stride 2, padding 1 implicit-GEMM convolution on MFMA f32 16x16x4.
 * in: (B,CIN,H,W) [CIN counts the synthetic ones channel if requested];
 * out: (B,NOUT,H/2,W/2).  Replaces aten::convolution for enc.1/enc.4/enc.7 and
 * aten::convolution_backward(data) of dec.0/dec.2/dec.4. */
int dm_conv4x4s2(const dm_operand *in, const dm_weight_view *w, float *out, const dm_epilogue *ep,
                 int B, int CIN, int NOUT, int H, int W, void *stream);
int dm_conv4x4s2_num_blocks(int B, int CIN, int NOUT, int H, int W, int per_tile);
/* floats of dm_weight_view.scratch this call wants (0: none).  fallback != 0: the input is AFFINE2 or the border-bias
 * table comes with side inputs, which the register-resident kernels do not take. */
int64_t dm_conv4x4s2_scratch_floats(int CIN, int NOUT, int H, int W, int fallback);

/* 3x3 stride 1 padding 1 (taps = 9) or 1x1 (taps = 1) convolution on MFMA.
 * pixel_shuffle = 1 turns it into a ConvTranspose2d(4, stride 2, padding 1):
 * logical output channel n = co*4 + py*2 + px is written to out[co][2y+py][2x+px],
 * the weight view addresses the 4x4 transposed-conv kernel ([c][co][ky][kx]
 * through sc/sn/sky/skx) and the 3x3 neighbourhood tap (ty,tx) of phase (py,px)
 * uses kernel element ky = py+3-2*ty, kx = px+3-2*tx when that lies in 0..3.
 * Replaces aten::convolution (enc.10, residual convs), aten::conv_transpose2d
 * (dec.0/2/4) and aten::convolution_backward(data) of every encoder conv. */
int dm_conv3x3(const dm_operand *in, const dm_weight_view *w, float *out, const dm_epilogue *ep,
               int B, int CIN, int NOUT, int H, int W, int taps, int pixel_shuffle, void *stream);
int dm_conv3x3_num_blocks(int B, int CIN, int NOUT, int H, int W, int taps, int pixel_shuffle, int per_tile);
int64_t dm_conv3x3_scratch_floats(int CIN, int NOUT, int H, int W, int taps, int pixel_shuffle, int per_tile);

/* Backward of a Conv2d(CX -> CD, kernel 4, stride 2, padding 1) in ONE kernel (csrc/conv_mfma.hip, kernel D) -- the
 * aten::convolution_backward of enc.4 (vq_vae.py:281): data gradient and weight gradient from one staging of its operands.
 *   dy      the output gradient on the (H, W) grid, (B, CD, H, W); normally AFFINE2 (BatchNorm backward: A*dy + B*a_out + C)
 *   in      the layer input as the forward read it, (B, CX, 2H, 2W); normally AFFINE_RELU (BatchNorm + ReLU on load)
 *   w       the layer's weight [CD][CX][4][4] viewed for the transposed convolution: sn = 16, sc = CX * 16, sky = 4, skx = 1
 *   dx      out: the data gradient (B, CX, 2H, 2W), zero where ep->mask (= the layer input, IDENT or AFFINE) is <= 0;
 *           ep->stats (dm_conv_bwd_s2_fused_num_blocks slabs of [CX][2] doubles): (sum dx, sum dx * input) per channel
 *           (ep->stat_q = the layer input or NULL); ep->bias / relu / resid must be unset
 *   w_slabs out: num_blocks slabs of CD*CX*16 floats, each in the weight's own layout; dm_reduce_slabs(_multi) adds them
 * Built for CD = 16, CX = 8, H % 8 == 0, W % 32 == 0 (dm_conv_bwd_s2_fused_supported); batch-statistics coefficients
 * (coef_bstride = 0) only. */
/* Arithmetic of the backward matrix products: 0 = the f32-input matrix instruction, bit for bit the fp32 multiply-add
 * chain -- the only arithmetic the library is built with.  mode 1 (split-bf16 operands, an opt-in of earlier rounds) is
 * retired: it was slower than the exact path on these layer widths; asking for it returns -1 (dm_last_error).  Any other
 * value only queries; returns the current setting (0). */
int dm_backward_precision(int mode);
int dm_conv_bwd_s2_fused_supported(int CD, int CX, int H, int W);
int dm_conv_bwd_s2_fused_num_blocks(int B, int CD, int CX, int H, int W);
int dm_conv_bwd_s2_fused(const dm_operand *dy, const dm_operand *in, const dm_weight_view *w, float *dx,
                         const dm_epilogue *ep, float *w_slabs, int B, int CD, int CX, int H, int W, void *stream);

/* Weight gradient:  R[cs][ct][ky][kx] = sum_{b,y,x} S[b,cs,y,x] * T[b,ct,y*s+ky-p,x*s+kx-p]
 * (k,s,p) in {(4,2,1),(3,1,1),(1,1,0)}.  For a Conv2d: S = output gradient,
 * T = layer input, R = dW [co][ci][k][k].  For a ConvTranspose2d: S = layer
 * input, T = output gradient, R = dW [ci][co][k][k].  Partial results go to
 * `slabs` (nblocks x CS*CT*k*k floats, nblocks = dm_wgrad_num_blocks) and are
 * summed in slab order into `dst` (deterministic). Replaces
 * aten::convolution_backward(weight).
 * T may be an AFFINE2 operand (two tensors: a BatchNorm backward folded into the load) only where
 * dm_wgrad_t_affine2_supported says so (the wide decoder's first transposed convolution: S plain, 64 x 32 channels, k = 4);
 * elsewhere the caller materialises it (dm_apply). */
int dm_wgrad_num_blocks(int B, int CS, int CT, int Hs, int Ws, int k);
int dm_wgrad_t_affine2_supported(int CS, int CT, int Hs, int Ws, int k);
int dm_wgrad(const dm_operand *S, const dm_operand *T, float *slabs, float *dst,
             int B, int CS, int CT, int Hs, int Ws, int k, void *stream);   /* dst = NULL: leave the slabs unreduced */

/* Backward of a 1x1 convolution whose output feeds a train-mode BatchNorm (the second convolution of a ResidualBlock
 * layer, vq_vae.py:207-209): aten::convolution_backward for input AND weight from ONE staging of its operands,
 *   dy     the output gradient as an operand (AFFINE2: BatchNorm's backward folded into the load; or IDENT), CD channels
 *   x      the layer input RAW (CX channels), xcoef [CX][4] = (c0, -, c2, -): the forward saw t = relu(c0 x + c2)
 *   w      [CD][CX] (the Conv2d weight)
 *   dx     [B][CX][H][W] out = (c0 x + c2 > 0) * sum_co w[co][ci] dy[co]     (the ReLU's mask applied)
 *   stats  dm_conv1x1_bwd_fused_num_blocks slabs of [CX][2] doubles: (sum dx, sum dx * x) -- what dm_bn_backward_finalize
 *          of the layer below needs
 *   wslabs the same number of slabs of CD*CX floats: partial dW[co][ci] = sum dy[co] * t[ci]; dm_reduce_slabs(_multi) adds them
 * Built for 32 -> 16 channels (num_residual_hiddens 32, num_hiddens 16) on grids of a multiple of 256 positions
 * (dm_conv1x1_bwd_fused_supported); other shapes: dm_conv3x3 (taps = 1) + dm_wgrad. */
int dm_conv1x1_bwd_fused_supported(int CD, int CX, int H, int W);
int dm_conv1x1_bwd_fused_num_blocks(int B, int CD, int CX, int H, int W);
int dm_conv1x1_bwd_fused(const dm_operand *dy, const float *x, const float *xcoef, const float *w, float *dx,
                         double *stats, float *wslabs, int B, int CD, int CX, int H, int W, void *stream);

/* Backward of a 3x3 convolution (padding 1) on a 16 x 16 latent grid whose output feeds a train-mode BatchNorm (enc.10 and
 * the first convolution of a ResidualBlock layer, vq_vae.py:287, 205): aten::convolution_backward for input AND weight from
 * ONE staging of a whole patch,
 *   dy     the output gradient as an operand (AFFINE2: BatchNorm's backward folded into the load; or IDENT), CD channels
 *   x      the layer input RAW (16 channels); xcoef [16][4] = (c0, -, c2, -): the forward saw t = relu(c0 x + c2); NULL: relu(x)
 *   w      [CD][16][3][3] (the Conv2d weight)
 *   resid  NULL or [B][16][16][16], added to dx after the mask (the residual join's other branch)
 *   q      NULL or [B][16][16][16]: second factor of the statistics (the raw output of the convolution below)
 *   dx     [B][16][16][16] out = (t > 0) * sum_{co,ky,kx} dy[co][y+1-ky][x+1-kx] w[co][ci][ky][kx]  (+ resid)
 *   stats  NULL or dm_conv3x3_bwd_fused_num_blocks slabs of [16][2] doubles: (sum dx, sum dx * q)   (q NULL: sum dx^2)
 *   wslabs the same number of slabs of CD*144 floats: partial dW; dm_reduce_slabs(_multi) adds them
 * Built for CD = 16 or 32 output channels, 16 input channels, H = W = 16 (a whole patch per workgroup) and, as bands of 8 rows
 * with halo rows, CD = 32 on H = W = 32 (the residual layers on the latents of 256-pixel patches; tensors then [B][..][32][32])
 * (dm_conv3x3_bwd_fused_supported); other shapes: dm_conv3x3 + dm_wgrad. */
int dm_conv3x3_bwd_fused_supported(int CD, int CX, int H, int W);
int dm_conv3x3_bwd_fused_num_blocks(int B, int CD, int CX, int H, int W);
int dm_conv3x3_bwd_fused(const dm_operand *dy, const float *x, const float *xcoef, const float *w, const float *resid,
                         const float *q, float *dx, double *stats, float *wslabs, int B, int CD, int CX, int H, int W,
                         void *stream);

/* Backward of Conv2d(16 -> 16, 4, stride 2, padding 1) from a 32 x 32 to a 16 x 16 grid whose output feeds a train-mode
 * BatchNorm (enc.7, vq_vae.py:284): aten::convolution_backward for input AND weight from ONE staging of a whole patch,
 *   dy     the output gradient as an operand (AFFINE2 or IDENT), [B][16][16][16]
 *   x      the layer input RAW [B][16][32][32]; xcoef [16][4] = (c0, -, c2, -): the forward saw t = relu(c0 x + c2)
 *   w      [16][16][4][4];   dx [B][16][32][32] out = (t > 0) * (transposed convolution of dy with w)
 *   stats  dm_conv4x4s2_bwd_fused_num_blocks slabs of [16][2] doubles: (sum dx, sum dx * x);  wslabs: as many slabs of 4096 floats
 * H, W are the OUTPUT grid of the convolution (16, 16: dm_conv4x4s2_bwd_fused_supported); other shapes: dm_conv3x3
 * (pixel shuffle) + dm_wgrad, or dm_conv_bwd_s2_fused for 8 -> 16 channels. */
int dm_conv4x4s2_bwd_fused_supported(int CD, int CX, int H, int W);
int dm_conv4x4s2_bwd_fused_num_blocks(int B, int CD, int CX, int H, int W);
int dm_conv4x4s2_bwd_fused(const dm_operand *dy, const float *x, const float *xcoef, const float *w, float *dx, double *stats,
                           float *wslabs, int B, int CD, int CX, int H, int W, void *stream);

/* Backward of a thin ConvTranspose2d(CI -> CO, 4, stride 2, padding 1) (dec.0, dec.2: vq_vae.py:291-296):
 * aten::convolution_backward for input AND weight from ONE staging of
 *   S      the layer input [B][CI][H][W] as the forward multiplied it (post-ReLU where the layer below has one)
 *   G      the output gradient [B][CO][2H][2W]
 *   w      [CI][CO][4][4] (the ConvTranspose2d weight)
 *   gin    [B][CI][H][W] out = sum_{co,ky,kx} G[co][2y+ky-1][2x+kx-1] w[ci][co][ky][kx], times [S > 0] when mask_relu
 *   stats  NULL, or dm_convt_bwd_fused_num_blocks slabs of [CI][2] doubles: (sum gin, 0) -- the bias gradient of the layer below
 *   wslabs the same number of slabs of CI*CO*16 floats: partial dW; dm_reduce_slabs(_multi) adds them
 * Built for (CI, CO) = (8, 4) with W % 32 == 0 and (16, 8) with W % 16 == 0, H % 8 == 0 (dm_convt_bwd_fused_supported);
 * other shapes: dm_conv4x4s2 + dm_wgrad. */
int dm_convt_bwd_fused_supported(int CI, int CO, int H, int W);
int dm_convt_bwd_fused_num_blocks(int B, int CI, int CO, int H, int W);
int dm_convt_bwd_fused(const float *S, const float *G, const float *w, float *gin, double *stats, float *wslabs,
                       int mask_relu, int B, int CI, int CO, int H, int W, void *stream);

/* ===== BatchNorm2d in training mode (vq_vae.py:206,209,279-288) =============== */

/* Forward finalize: reduce stats slabs (sum x, sum x^2) -> batch mean / biased
 * variance -> coef[c] = (gamma*invstd, 0, beta - mean*gamma*invstd, 0),
 * saved[c] = (mean, invstd); running stats updated with momentum and the
 * unbiased variance, num_batches_tracked incremented (nn.BatchNorm2d defaults).
 * per_sample = 1: slabs are grouped per sample (slabs_per_group each), one
 * (mean, invstd, coef) per (sample, channel); running stats are updated once per
 * sample in order, as B successive batch-of-one calls would. */
int dm_bn_finalize(const double *stats, int nslabs, int slabs_per_group, int C, int64_t count_per_group,
                   const float *gamma, const float *beta, float *running_mean, float *running_var,
                   int64_t *num_batches_tracked, float momentum, float eps,
                   float *coef, float *saved, int per_sample, void *stream);

/* Per-sample mode only: running_mean = running_var = num_batches_tracked = NULL leaves the module state alone;
 * dm_bn_running_replay then brings the running statistics of up to 16 layers up to date in ONE launch from the same
 * slabs (the inference path needs a layer's coefficients before the next convolution, its running statistics never:
 * patch_VAE.py:445-452 only reads the latents). */
typedef struct dm_bn_replay_seg {
    const double *stats;        /* the slabs handed to dm_bn_finalize(per_sample = 1) */
    int32_t nslabs, slabs_per_group, C;
    int64_t count_per_group;
    float *running_mean, *running_var;      /* (C) each, or both NULL */
    int64_t *num_batches_tracked;           /* or NULL */
    float momentum;
} dm_bn_replay_seg;
int dm_bn_running_replay(const dm_bn_replay_seg *segs, int nseg, void *stream);

/* ----- the 16 x 16 part of the encoder, per-sample statistics, one kernel (csrc/latent_tail.hip) ----------------------
 * Replaces vq_vae.py:287-289 (enc.10 Conv2d 3x3, enc.11 BatchNorm2d, enc.12 ResidualBlock = :203-224) as
 * pipeline/patch_VAE.py:445-452 runs them: batch-of-one calls in train mode, i.e. every BatchNorm normalises a patch with
 * that patch's own statistics.  A workgroup takes a patch from a3 (raw output of enc.7) to z (the encoder's output) with
 * the activations in LDS / registers; the per-patch sums of the five (1 + 2 nres) BatchNorm inputs are written out as one
 * slab per patch and layer for dm_bn_running_replay (slabs_per_group = 1, count_per_group = 256).
 *   a3 (B,16,16,16); coef3 (B,16,4): per-sample coefficients of enc.8 from dm_bn_finalize(per_sample = 1);
 *   w10 (16,16,3,3), b10 (16) or NULL; gamma4 / beta4 (16) or NULL (1 / 0); stats4 (B,16,2) doubles;
 *   per residual layer: wa (32,16,3,3), ba (32), gamma_a / beta_a (32), stats_a (B,32,2); wb (16,32[,1,1]), bb (16),
 *   gamma_b / beta_b (16), stats_b (B,16,2);  z (B,16,16,16).
 * Built for num_hiddens 16, num_residual_hiddens 32, a 16 x 16 latent grid (128 x 128 patches), <= 4 residual layers:
 * dm_latent_tail_supported() tells; other shapes take the layer-by-layer kernels. */
typedef struct dm_latent_tail_res {
    const float *wa, *ba, *gamma_a, *beta_a;
    double *stats_a;
    const float *wb, *bb, *gamma_b, *beta_b;
    double *stats_b;
    float eps_a, eps_b;
} dm_latent_tail_res;
typedef struct dm_latent_tail_args {
    const float *a3, *coef3, *w10, *b10, *gamma4, *beta4;
    double *stats4;
    float *z;
    float eps4;
    int32_t B, C, CR, H, W, nres;
    dm_latent_tail_res res[4];
    /* Start one layer earlier (all NULL / 0: start at a3): a2 (B,16,32,32) = raw output of enc.4 with coef2 (B,16,4), the
     * per-sample coefficients of enc.5; then enc.7 (w7 (16,16,4,4), b7) and enc.8 (gamma3, beta3, eps3) run in the kernel
     * too, a3 / coef3 are ignored and stats3 (B,16,2) receives the per-patch sums of enc.7's output. */
    const float *a2, *coef2, *w7, *b7, *gamma3, *beta3;
    double *stats3;
    float eps3;
} dm_latent_tail_args;
int dm_latent_tail_supported(int C, int CR, int H, int W, int nres);
int dm_latent_tail_forward(const dm_latent_tail_args *args, void *stream);

/* Backward finalize: slabs hold (sum dy, sum dy*a).  Writes dgamma, dbeta and the
 * AFFINE2 coefficients (A,B,C) with da = A*dy + B*a + C.  count == 0: fixed statistics (eval() mode; saved = running mean and
 * 1 / sqrt(running_var + eps)): da = gamma * invstd * dy, dgamma / dbeta as always. */
int dm_bn_backward_finalize(const double *stats, int nslabs, int C, int64_t count,
                            const float *gamma, const float *saved, float *dgamma, float *dbeta,
                            float *coef_bwd, void *stream);

/* out = load(in) [+ resid];  used for h = BN(a) and h' = h + BN(r) (vq_vae.py:224). */
int dm_apply(const dm_operand *in, const float *resid, float *out, int B, int C, int H, int W, void *stream);

/* stats[block][c] = (sum p, sum p*q) over a (B,C,H,W) pair; blocks = dm_channel_stats_num_blocks. */
int dm_channel_stats_num_blocks(int B, int C, int H, int W);
int dm_channel_stats(const float *p, const float *q, double *stats, int B, int C, int H, int W, void *stream);

/* dst[n] = scale * sum_over_slabs stats[slab][n][0]   (bias gradients from epilogue stats). */
int dm_sum_slabs(const double *stats, int nslabs, int N, float scale, float *dst, void *stream);
/* Same sums scattered to up to 8 destinations: entries [end[k-1], end[k]) go to dst[k] (one launch instead of a
 * sum plus one copy per parameter). */
typedef struct dm_scatter {
    int32_t nseg;
    int32_t end[8];
    float *dst[8];
} dm_scatter;
int dm_sum_slabs_scatter(const double *stats, int nslabs, int N, float scale, const dm_scatter *sc, void *stream);

/* ===== decoder head + reconstruction loss (vq_vae.py:298, 320-323) ============ */

/* decoded = Conv2d(C4 -> NIN, 1x1)(d4) + bias; loss partials
 * sum ((decoded*m - x*m)^2 / channel_var[c]) as one double per workgroup.
 * mask: (B,MC,H,W) with MC in {1, NIN}, or NULL (= ones).
 * x = NULL: decoder-only call (VQ_VAE.dec(z)), no loss partials are written.
 * Built for C4 = num_hiddens//4 in {4, 8, 16} and NIN 1..4 (dm_head_supported tells); other widths run the same
 * arithmetic as dm_conv3x3(taps=1) + dm_recon_loss + dm_wgrad. */
int dm_head_supported(int C4, int NIN);
int dm_head_num_blocks(int B, int H, int W);
int dm_head_forward(const float *d4, const float *w6, const float *b6, const float *x, const float *mask,
                    int mask_channels, const float *channel_var, float *decoded, double *loss_slabs,
                    int B, int C4, int NIN, int H, int W, void *stream);

/* Gradient of the head: g_dec = gscale[0]*2*(dec*m - x*m)*m/(var*N) with
 * N = B*NIN*H*W (skipped when gscale_dev is NULL) plus gdec_ext (an upstream
 * gradient w.r.t. decoded, may be NULL); g4 = (W6^T g_dec) * (d4 > 0); partial
 * sums per workgroup, laid out as dm_sum_slabs expects ([block][n][2]):
 * n in [0, NIN*C4) = dW6, then NIN entries db6, then C4 entries sum g4 (bias grad of dec.4). */
int dm_head_backward(const float *decoded, const float *x, const float *mask, int mask_channels,
                     const float *channel_var, const float *d4, const float *w6, const float *gscale_dev,
                     const float *gdec_ext, float *g4, double *part_slabs, int B, int C4, int NIN, int H, int W,
                     void *stream);

/* ----- fused decoder tail (dec.4 + dec.5 + dec.6 + loss, vq_vae.py:296-298, 320-323) ---------------------
 * The 4 x 2H2 x 2W2 tensor d4 = relu(dec.4(d2)) and its gradient are never written to HBM.
 * Built for num_hiddens//4 = 4 channels, 1..4 input channels, H2 a multiple of 8 and W2 a multiple of 4 (W2 = 64, the
 * 128 x 128 patches, has its own instantiation: one tile spans the row; other widths run 64-lane tiles with 56 owned
 * columns); dm_dec_tail_supported() tells. */
int dm_dec_tail_supported(int C2, int NIN, int H2, int W2);
int dm_dec_tail_num_blocks(int B, int H2, int W2);
/* d2 (B,4,H2,W2) post-ReLU input of dec.4; w4 (4,4,4,4) ConvTranspose2d weight [ci][co][ky][kx]; w6 (NIN,4).
 * decoded (B,NIN,2H2,2W2); loss_slabs: dm_dec_tail_num_blocks doubles (x = NULL: decoder-only, no loss). */
int dm_dec_tail_forward(const float *d2, const float *w4, const float *b4, const float *w6, const float *b6,
                        const float *x, const float *mask, int mask_channels, const float *channel_var,
                        float *decoded, double *loss_slabs, int B, int C2, int NIN, int H2, int W2, void *stream);
/* Backward of the same block for d(total)/d(recon_loss) = gscale_dev[0]:
 *   g2 (B,4,H2,64) = gradient w.r.t. the PRE-ReLU output of dec.2 (already masked by d2 > 0);
 *   part_slabs [nblocks][NIN*4 + NIN + 4 + 4][2] doubles = partial sums of dW6 | db6 | db4 | db2 (dm_sum_slabs);
 *   w_slabs [nblocks][256] floats = partial dW4 in the parameter's layout (dm_reduce_slabs). */
int dm_dec_tail_backward(const float *d2, const float *w4, const float *b4, const float *w6,
                         const float *decoded, const float *x, const float *mask, int mask_channels,
                         const float *channel_var, const float *gscale_dev, float *g2, double *part_slabs,
                         float *w_slabs, int B, int C2, int NIN, int H2, int W2, void *stream);
/* Training pass of the decoder tail: dm_dec_tail_forward's loss and dm_dec_tail_backward's gradients in ONE kernel
 * (run_training.py:404-406: model(batch) -> total_loss.backward()).  The reconstruction-loss gradient needs no
 * global reduction, so `decoded` (vq_vae.py:298,319) is formed per tile in LDS, used for the loss partials and for
 * g_dec, and never written: 262 144 B/patch (d2 + x read, g2 written) instead of 720 896 for the two kernels.
 *   loss_slabs [nblocks] doubles as in dm_dec_tail_forward; the other outputs as in dm_dec_tail_backward. */
int dm_dec_tail_train(const float *d2, const float *w4, const float *b4, const float *w6, const float *b6,
                      const float *x, const float *mask, int mask_channels, const float *channel_var,
                      const float *gscale_dev, float *g2, double *part_slabs, float *w_slabs, double *loss_slabs,
                      int B, int C2, int NIN, int H2, int W2, void *stream);
/* dst[e] = sum over slabs of slabs[slab][e], fixed order (bitwise reproducible). */
int dm_reduce_slabs(const float *slabs, int nslabs, int E, float *dst, void *stream);
/* The same for up to 32 (slabs, dst) pairs in ONE launch: the weight gradients of a whole backward pass. */
typedef struct dm_reduce_seg {
    const float *slabs;
    float *dst;
    int32_t nslabs;
    int32_t E;
    int32_t stride;             /* elements from one slab to the next; 0 = E (dense float slabs) */
    int32_t pairs_of_doubles;   /* 1: `slabs` points at (value, -) double pairs of a statistics epilogue (bias gradients,
                                   dm_sum_slabs' input): the first of each pair is summed in double; stride counts pairs */
} dm_reduce_seg;
int dm_reduce_slabs_multi(const dm_reduce_seg *segs, int nseg, void *stream);

/* scalars_out = (recon, commitment, total, perplexity) from the loss slabs and
 * the dm_vq_finalize scalars: recon = sum/N, total = w_recon*recon + w_commit*commitment. */
int dm_loss_finalize(const double *loss_slabs, int nslabs, int64_t count, const float *vq_scalars,
                     float weight_recon, float weight_commitment, float *scalars_out, void *stream);

/* ===== plain reconstruction loss (VQ_VAE_z32, vae.py:450-452: no 1x1 head to fuse with) ==== */
/* loss_slabs[dm_recon_loss_num_blocks] = partial sums of (dec*m - x*m)^2 / var[c]  (dm_loss_finalize divides by N). */
int dm_recon_loss_num_blocks(int B, int NIN, int H, int W);
int dm_recon_loss(const float *decoded, const float *x, const float *mask, int mask_channels,
                  const float *channel_var, double *loss_slabs, int B, int NIN, int H, int W, void *stream);
/* g_decoded = gscale * 2/N * (dec*m - x*m) * m / var[c]; bias_slabs[num_blocks][NIN][2] = per-channel sums of g_decoded
 * (the gradient of the last layer's bias; dm_sum_slabs). */
int dm_recon_loss_backward(const float *decoded, const float *x, const float *mask, int mask_channels,
                           const float *channel_var, const float *gscale_dev, float *g_decoded,
                           double *bias_slabs, int B, int NIN, int H, int W, void *stream);

/* ===== time-matching loss (vq_vae.py:324-332, vae.py:322-336) ================= */

/* sim[i][j] = mean_d (z[i][d] - z[j][d])^2 for the B flattened latents z (B, n) -- the reference's
 * pow(z.reshape(1,B,n) - z.reshape(B,1,n), 2).mean(2) without its (B, B, n) intermediate.  The weighting, hinge and
 * mean over the (B, B) matrix stay with the caller (B*B elements). */
int dm_pair_msd(const float *z, float *sim, int B, int n, void *stream);
/* Its backward: dz[i] = (2/n) * sum_j (g_sim[i][j] + g_sim[j][i]) * (z[i] - z[j]). */
int dm_pair_msd_backward(const float *z, const float *g_sim, float *dz, int B, int n, void *stream);

/* The whole term on the matrix pipe (csrc/pairwise.hip): sim from the Gram matrix Z Z^T (f32 MFMA, K split into chunks added
 * in double), the loss form of either model family, and S_ij = dloss/dsim_ij + dloss/dsim_ji for the backward, which is a
 * second MFMA GEMM: dz_i = (2/n) (rowsum(S)_i z_i - sum_j S_ij z_j).
 * Near pairs -- Gram distance below 1/16 of |z_i|^2 + |z_j|^2, where the Gram form has cancelled: adjacent frames of one
 * cell, the pairs the term exists for -- are re-evaluated from differences as the reference does (vae.py:441-455), forward
 * (sim) and backward (S_ij (z_i - z_j)); S is therefore TWO (B, B) planes: [0] the far pairs' S for the GEMM, [1] the near
 * pairs' (zero elsewhere).
 *   mode 0  vq_vae.py:330-331   loss = sum_ij sim_ij * tm_ij
 *   mode 1  vae.py:327-336      w = {2: w_a, 1: w_t, 0: w_n}[tm]; v = sim * w; tm == 0: v = max(v + margin, 0); loss = mean v
 * z (B, n) contiguous with n % 32 == 0 (dm_time_matching_supported; other lengths: dm_pair_msd); tm (B, B) float32;
 * workspace: dm_time_matching_workspace_floats(B, n) floats of scratch; S (2, B, B) out; loss_slabs: dm_time_matching_num_slabs(B)
 * pairs of doubles, (partial loss, 0) each -- dm_sum_slabs(loss_slabs, nslabs, 1, 1, loss) gives the scalar.
 * dm_time_matching_backward: dz = scale * g_loss_dev[0] * d loss / d z  (g_loss_dev NULL: 1). */
int dm_time_matching_supported(int B, int n);
int64_t dm_time_matching_workspace_floats(int B, int n);
int dm_time_matching_num_slabs(int B);
int dm_time_matching_forward(const float *z, const float *tm, int B, int n, int mode, float w_a, float w_t, float w_n,
                             float margin, float *workspace, int64_t workspace_floats, float *S, double *loss_slabs,
                             void *stream);
int dm_time_matching_backward(const float *z, const float *S, const float *g_loss_dev, float scale, float *dz, int B, int n,
                              void *stream);
/* The same with another gradient of the same latents added on the way out: dz = add + scale * g * dloss/dz (`add` may be
 * `dz` itself).  The training step adds the pairwise term's gradient to the quantiser's without an elementwise pass. */
int dm_time_matching_backward_add(const float *z, const float *S, const float *g_loss_dev, float scale, const float *add,
                                  float *dz, int B, int n, void *stream);
/* The same pair with a STATE block the two calls share (dm_time_matching_state_ints(B) int32 of caller memory, written by
 * the forward call and read by the kernels of both).  (i) mode 0 (vq_vae.py:331: sum of sim * time_matching_mat) only needs
 * the pairs with a nonzero entry, and a batch's relation matrix holds a handful per row: the forward call counts them on
 * the device, and at up to 32 per row the Gram product is skipped -- every related pair is evaluated from differences
 * (exactly, as the reference does) and its gradient added row by row.  (ii) In every mode the forward call marks which
 * (64 rows x 32 columns) blocks of S hold a nonzero, and the gradient product multiplies only those: in the z16 / z32 form
 * an unrelated pair beyond the hinge's margin has no gradient, so S is as sparse as the relation matrix once such pairs
 * lie apart.  No host decision, the same launches either way (a captured step stays valid whatever the matrix holds), and
 * the same sums to the bit as the stateless calls.  `add` may be NULL. */
int dm_time_matching_state_ints(int B);
int dm_time_matching_forward_state(const float *z, const float *tm, int B, int n, int mode, float w_a, float w_t, float w_n,
                                   float margin, float *workspace, int64_t workspace_floats, float *S, double *loss_slabs,
                                   int32_t *state, void *stream);
int dm_time_matching_backward_state(const float *z, const float *S, const float *g_loss_dev, float scale, const float *add,
                                    float *dz, int B, int n, const int32_t *state, void *stream);

/* ===== enc.0 o enc.1 composition (vq_vae.py:277-278) ========================== */

/* enc.1(enc.0(x)) is linear in (x, 1): Weff[c1][ci][ky][kx] = sum_c W1[c1][c][ky][kx]*W0[c][ci]
 * for ci < NIN and Weff[c1][NIN][ky][kx] = sum_c W1[c1][c][ky][kx]*b0[c] (the
 * ones channel carries enc.0's bias through enc.1's zero padding exactly). */
int dm_e1_compose(const float *w0, const float *b0, const float *w1, float *weff,
                  int NIN, int C0, int C1, void *stream);
/* The same plus the forward pass's shortcut for the ones channel: out = conv(x, Weff[:, :NIN]) + bias_border where
 * bias_border[ry][rx][c1] = b1[c1] + sum over the taps (ky,kx) that stay inside the image for an output position of
 * row class ry / column class rx of Weff[c1][NIN][ky][kx] (4x4, stride 2, padding 1: the first row misses ky = 0,
 * the last row ky = 3, likewise for columns).  dm_conv4x4s2 then runs K = 16*NIN instead of 16*(NIN+1). */
int dm_e1_compose_border(const float *w0, const float *b0, const float *w1, const float *b1, float *weff,
                         float *bias_border, int NIN, int C0, int C1, void *stream);
/* Chain rule back to the stored parameters from dWeff. */
int dm_e1_chain(const float *dweff, const float *w0, const float *b0, const float *w1,
                float *dw0, float *db0, float *dw1, int NIN, int C0, int C1, void *stream);

/* ===== optimizer (run_training.py:485: Adam(lr, betas=(.9,.999)), eps 1e-8) ==== */
/* step_dev[0] holds the 1-based step count as float (kept on device so the call is graph-capturable). */
int dm_adam(float *param, const float *grad, float *m, float *v, int64_t n,
            float lr, float beta1, float beta2, float eps, const float *step_dev, void *stream);

/* The same keeping the count itself: steps_done[0] = completed steps, this call is step steps_done[0] + 1 and writes
 * that to steps_done_next[0] (a different word; callers alternate two counters), so the step needs no "+= 1" launch. */
int dm_adam_counted(float *param, const float *grad, float *m, float *v, int64_t n, float lr, float beta1, float beta2,
                    float eps, const float *steps_done, float *steps_done_next, void *stream);
/* The same on grad[i] * grad_scale: data parallel, the bucket holds the all-reduced SUM of the ranks' gradients and
 * grad_scale = 1 / world makes it the mean inside the optimizer's load (no separate scaling launch behind the collective). */
int dm_adam_counted_scaled(float *param, const float *grad, float *m, float *v, int64_t n, float lr, float beta1,
                           float beta2, float eps, float grad_scale, const float *steps_done, float *steps_done_next,
                           void *stream);

/* ===== per-patch z-score (pipeline/train_utils.py:252-274, applied at patch_VAE.py:413-419) ============ */
/* out[plane] = float((in[plane] - mean) / (std + eps)), population std over the HW elements of each of the `planes`
 * (patch, channel) planes, arithmetic in double; `in` is float64 (in_is_f64 = 1, what the reference z-scores) or
 * float32. */
int dm_zscore_patch(const void *in, int in_is_f64, float *out, int planes, int HW, void *stream);
/* Dataset-wide per-channel z-score with given statistics, then the cast to float32 (pipeline/train_utils.py:228-250 as
 * run_training.py:880 applies it to the pickled patches before train()): out[n][c][.] = float((in - mean[c]) / denom[c]),
 * in: (N, C, H*W) float64 or float32, mean / denom: C doubles on the device, denom = std + eps as numpy forms it.
 * diff_f64 / quot_f64: the types numpy's expression gives the difference and the quotient (a float32 dataset against
 * Python-float statistics: float32 difference; its quotient by `std + np.finfo(float).eps` is float64 under NumPy 2 and
 * float32 under NumPy 1).  Bit-equal to the numpy expression. */
int dm_zscore_channels(const void *in, int in_is_f64, int diff_f64, int quot_f64, float *out, const double *mean,
                       const double *denom, int64_t N, int C, int64_t HW, void *stream);

/* ===== on-device augmentation (run_training.py:396-403) ======================= */
/* out[b] = rot90(flip(in[b], flip_code[b]), k = rot_code[b]) on square (C,H,H) patches;
 * flip_code 0 none / 1 flip H / 2 flip W, rot_code 0..3 (counter-clockwise, dims [1,2]). */
int dm_augment(const float *in, float *out, const int32_t *flip_code, const int32_t *rot_code,
               int B, int C, int H, void *stream);

/* ===== feeding the step from a dataset resident in HBM (run_training.py:504-532) ======================= */
/* out[b] = rot90(flip(src[ids[b]], flip_code[b]), rot_code[b]): `dataset[ids][0].to(device)` (run_training.py:512) and the
 * per-sample augmentation loop (run_training.py:396-403) as ONE launch that writes the batch where the step reads it.
 * src is the whole dataset (n_src, C, H, H) fp32 in HBM; ids NULL = the first B samples in order (a batch that was
 * copied over PCIe already); flip_code / rot_code NULL = no augmentation.  An id outside [0, n_src) yields zeros. */
int dm_gather_augment(const float *src, int64_t n_src, const int32_t *ids, const int32_t *flip_code,
                      const int32_t *rot_code, float *out, int B, int C, int H, void *stream);
/* out[b] = src[ids[b]] for rows of row_floats floats (a multiple of 4; 16-byte aligned pointers): the mask planes of a
 * batch, `mask[ids][0][:, 1:2]` of run_training.py:371 with the {-1,1} -> {0,1} map applied once at upload. */
int dm_gather_rows(const float *src, int64_t n_src, const int32_t *ids, float *out, int B, int64_t row_floats,
                   void *stream);
/* out (B, B) = relation_mat[ids, :][:, ids].todense() (run_training.py:348-351) from the CSR arrays of the (n, n)
 * matrix in HBM (duplicates summed, float32 values).  pos: n int64 of caller scratch, zeroed once; stamp: a number in
 * [1, 2^31) that differs from the previous call's on the same scratch (entries of older calls are recognised by it). */
int dm_csr_block(const int64_t *indptr, const int32_t *indices, const float *data, int64_t n, const int32_t *ids, int B,
                 int64_t *pos, int64_t stamp, float *out, void *stream);
/* HOST function (no device work): the flip / rotation codes of n samples parsed from `raw`, a run of 32-bit words of
 * numpy's legacy generator, exactly as n interleaved np.random.choice([0,1,2]) / np.random.choice([0,1,2,3]) calls
 * (run_training.py:399-402) consume them.  Returns the number of words consumed, -1 if `raw` is too short. */
int64_t dm_augment_codes(const uint32_t *raw, int64_t n_raw, int64_t n, int32_t *flip_code, int32_t *rot_code);
/* HOST function: the sample order of reorder_with_trajectories (run_training.py:97-140) for n samples -- random picks
 * among the remaining ids, each followed by its trajectory over the ADJACENT (value 2) pairs, breadth first -- from a run
 * of the legacy generator's 32-bit words, in O((n + pairs) log n) instead of the reference's O(n^2).  adj_ptr (n + 1) /
 * adj_idx: the value-2 pairs as CSR over their first id, in the relation dict's order.  order: n ids out.  Returns the
 * words consumed; -1 `raw` too short; -2 / -3 the reference's KeyError cases (a reached sample without an adjacency row /
 * already taken), *err_id naming the sample; -4 bad argument or no memory. */
int64_t dm_reorder_with_trajectories(const uint32_t *raw, int64_t n_raw, int64_t n, const int64_t *adj_ptr,
                                     const int64_t *adj_idx, int64_t *order, int64_t *err_id);

#ifdef __cplusplus
}
#endif
#endif /* DYNAMORPH_HIP_H */
