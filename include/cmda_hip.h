/* cmda_hip.h -- C ABI of libcmda_hip.so, the MI355X (gfx950) kernel library under the CMDA hot path.
 *
 * The reference (XiaRho/CMDA) has no native code: every "kernel" is a stock torch op reached from the Python
 * modules listed in SURVEY.md section 8(a).  Each entry point below names the reference call site it replaces
 * (paths relative to the reference repo root).  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add inside those modules.
 *
 * Conventions
 *  - plain pointers + sizes only; the caller owns every buffer (inputs, outputs, workspaces, saved statistics);
 *  - every call only enqueues work on `stream` (a hipStream_t): no allocation, no synchronisation, re-entrant;
 *  - returns 0 (CMDA_OK) or a negative error code; never throws;
 *  - activations are NHWC / NLC ("tokens x channels"); `dtype` selects the activation storage type
 *    (CMDA_F32 = exact-fp32 parity mode, CMDA_BF16 = speed mode, fp32 accumulate); parameters, statistics,
 *    logits and parameter gradients are always fp32; parameter gradients are ACCUMULATED (+=).
 */
#ifndef CMDA_HIP_H_
#define CMDA_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMDA_OK 0
#define CMDA_ERR_SHAPE -1
#define CMDA_ERR_DTYPE -2
#define CMDA_ERR_HIP -3
#define CMDA_ERR_UNSUPPORTED -4

#define CMDA_F32 0
#define CMDA_BF16 1
/* cmda_gemm_params_t.dtype only: fp32 storage like CMDA_F32, but the contraction runs on SPLIT-bf16 operands (x = hi + lo, three bf16
 * MFMAs per k-step with fp32 accumulate: ~16 mantissa bits per product, ~1e-5 relative) instead of the exact-fp32 matrix
 * instruction, which runs at 1/16 of the bf16 rate -- the parity mode that meets the 1e-3 logit tolerance at 3/16 of that cost */
#define CMDA_F32X3 2

int cmda_abi_version(void);

/* ---- GEMM / implicit-GEMM convolution -------------------------------------------------------------------
 * Operand view V(r,c): plain row-major matrix (conv=0: element r*ld+c) or im2col view of an NHWC tensor
 * (conv=1: r=(b,oh,ow), c=(kh,kw,ci)).  Replaces nn.Linear / nn.Conv2d / torch.matmul at
 * mmseg/models/backbones/mix_transformer.py:31-44,80-102,169-183; decode_heads/segformer_head.py:25-28;
 * decode_heads/daformer_head.py:63-79; decode_heads/decode_head.py:563-586; cyclegan/cyclegan_model.py:339-374. */
typedef struct cmda_view_t {
  const void* ptr;
  int64_t ld;           /* plain: elements between consecutive r */
  int64_t R, Cc;        /* extent of r and c */
  int64_t batch_stride; /* elements between batch entries (grid z, outer) */
  int64_t batch2_stride; /* elements between inner batch entries (e.g. attention heads) */
  int32_t conv;         /* 0 plain, 1 im2col view, 2 im2col view of a kernel == stride, pad 0, dil 1 convolution with H = OH*stride,
                           W = OW*stride (non-overlapping patches: same matrix as 1, filled like a plain operand) */
  int32_t H, W, C;      /* conv: input height/width/channels (NHWC) */
  int32_t OH, OW;       /* conv: output height/width */
  int32_t KH, KW, stride, pad, dil;
  int32_t in_dil;       /* conv: >1 = input zero-insertion (transposed conv) */
  int32_t reflect;      /* conv: reflection padding instead of zeros */
  int32_t vec_ok;       /* 16-byte chunk loads along c are legal (alignment + divisibility) */
} cmda_view_t;

typedef struct cmda_gemm_params_t {
  cmda_view_t A, B;     /* C[m,n] = epi(alpha * sum_k A(m,k) B(n,k)) */
  int32_t a_kstrided;   /* 0: A view is (r=m, c=k); 1: (r=k, c=m) */
  int32_t b_kstrided;   /* 0: B view is (r=n, c=k); 1: (r=k, c=n) */
  void* C;
  int64_t ldc, c_batch_stride, c_batch2_stride;
  int32_t M, N, K, batch, batch2, splits; /* grid z = (batch*batch2)*splits; splits <= 0 with atomic=1: auto */
  float alpha, beta;
  const float* bias;    /* [N] or NULL */
  int32_t act;          /* 0 none, 1 ReLU, 2 GELU(erf), 3 tanh */
  const void* res;      /* residual, activation dtype, or NULL */
  int64_t ldres, res_batch_stride, res_batch2_stride;
  const float* rowscale; /* per-sample drop-path scale or NULL; index m / rows_per_scale */
  int32_t rows_per_scale;
  int32_t out_f32;      /* C is fp32 regardless of dtype */
  int32_t atomic;       /* C += via fp32 atomics (split-K / gradient accumulation) */
  int32_t dtype;
  int32_t c_vec_ok;     /* C, res and bias may be accessed 4 elements at a time (pitches, offsets % 4 == 0, 16-B bases) */
  float* colsum;        /* optional, A K-strided only: colsum[m] += sum_k A(m,k) (bias gradient fused into wgrad) */
  /* optional output re-layout "un-patchify" (data gradient of a kernel == stride convolution, the spatial-reduction conv
   * of mix_transformer.py:70-75): row m = (b*OH + oh)*OW + ow, column n = (kh*KW + kw)*Ci + ci is stored at
   * (((b*OH + oh)*KH + kh) * OW*KW + ow*KW + kw) * Ci + ci of C, i.e. straight into the NHWC input gradient instead of a
   * column buffer.  Off when c_patch_ow == 0; ldc / batch strides / residual are ignored in this mode. */
  int32_t c_patch_ow, c_patch_kh, c_patch_kwci;
  /* 0: the library's tile heuristics; -1: register-staged kernel instead of the LDS-DMA one.  > 0 (tuning sweeps, tests of the rarely
   * chosen kernels): low 4 bits 1..4 force tile 128x128 / 128x64 / 64x64 / 256x256; bits 4-7 LDS stages (4 = four);
   * bit 8 no tile-group walk; bit 9 gemm_glds_kernel instead of the ping-pong / weight-gradient 256x256 kernels; bit 10 ping-pong
   * kernel; bit 11 general DMA address path; bit 12 (32, 4) weight-gradient configuration; bit 13 general kernel instead of the lean
   * one; bit 14 lean kernel on four waves. */
  int32_t tile_hint;
  /* atomic stores only, c_perm_ci > 0: GEMM column n = cell * c_perm_ci + ci (cell = kh*KW + kw, the im2col column order) is
   * stored at column ci * c_perm_cells + cell -- a convolution's weight gradient accumulated straight into the parameter's
   * [Co][Ci][KH][KW] gradient. */
  int32_t c_perm_ci, c_perm_cells;
  /* the residual `res` is fp32 whatever `dtype` says (with out_f32: the fp32 residual stream of the bf16 mode, see
   * cmda_layernorm_fwd2) */
  int32_t res_f32;
  /* colstats != NULL: column statistics of the OUTPUT fused into the epilogue -- the BatchNorm / InstanceNorm behind a convolution
   * (mmcv ConvModule conv -> norm, daformer_head.py:46-62, sep_aspp_head.py:18-27; cyclegan_model.py:339-434) no longer re-reads the
   * activation for its batch statistics.  For every stored element v = C[m][n] (the general epilogue and the dilated depthwise walk take the value as computed, before rounding to
   * the storage type; the 256 x 256 ping-pong kernel takes the rounded bf16 value it stores -- the statistics differ by one rounding of the inputs):
   * colstats[g][slot][0][n] += v and colstats[g][slot][1][n] += v * v with g = m / colstats_rows and an arbitrary slot < 32, i.e. the
   * layout of the BatchNorm workspace (groups x cmda_bn_ws_floats(N) floats, ZERO on entry); cmda_bn_train_fwd / fwd2 with
   * ws_has_stats = 1 finish the normalisation from it.  Requires: forward form (A, B not K-strided outputs of atomics: atomic == 0,
   * splits <= 1), act == 0, no rowscale, no c_patch_*, batch == batch2 == 1, N % 4 == 0, colstats_rows % 256 == 0.  The library then
   * runs a kernel with the general epilogue (CMDA_ERR_UNSUPPORTED where none takes the problem). */
  int32_t colstats_rows;
  float* colstats;
} cmda_gemm_params_t;

int cmda_gemm(const cmda_gemm_params_t* p, void* stream);


/* K x K convolution with ONE output channel, stride 1, reflection (or zero) padding `pad`, NHWC x [B,H,W,C] and khwc weights
 * [K*K*C] in the activation dtype, out fp32 [B,H,W] = act(bias[0] + sum): the last layer of the Motion-Extractor generator
 * (ReflectionPad2d(3) + Conv2d(64,1,7) + Tanh, cyclegan/cyclegan_model.py:366-369).  Built for K = 7, C = 64, pad = 3 in bf16 (packed
 * dot products) and fp32 storage (both parity modes); CMDA_ERR_UNSUPPORTED otherwise (the caller then uses cmda_gemm). */
int cmda_conv_co1(const void* x, const void* w, const float* bias, float* out, int B, int H, int W, int C, int K, int pad,
    int reflect, int act, int dtype, void* stream);

/* GROUPED launch of n GEMMs that are independent of each other (no problem reads another's output, and outputs shared between
 * problems are ACCUMULATED -- atomic != 0 -- so any order is correct: the grouped buckets are launched first, the problems that
 * cannot be grouped afterwards, NOT in list order): the
 * DEFERRED weight gradients of a backward pass -- `dW += dY^T X` of every nn.Linear / nn.Conv2d the pass walked
 * (mix_transformer.py:31-44,62-76,169-173; decode_heads/segformer_head.py:25-28 under torch autograd) -- as ONE grid per (tile,
 * operand mode) instead of n latency-bound launches.  Problems in weight-gradient form (bf16, atomic fp32 output, both operands
 * K-strided, splits <= 0, LDS-DMA-able views) are grouped; every other problem is launched by itself through cmda_gemm.
 * ws_host: PINNED host memory, ws_dev: device memory, both >= cmda_gemm_grouped_ws_bytes(params, n) bytes, 16-byte aligned, owned
 * by the caller.  upload != 0: the parameter table / block map are (re)built in ws_host and copied to ws_dev by a kernel on
 * `stream` (ws_host must stay untouched until that kernel has run -- for as long as a captured graph replays it); upload == 0:
 * ws_dev still holds the table of an earlier call with IDENTICAL params. */
int64_t cmda_gemm_grouped_ws_bytes(const cmda_gemm_params_t* params, int n);
int cmda_gemm_grouped(const cmda_gemm_params_t* params, int n, void* ws_host, void* ws_dev, int64_t ws_bytes, int upload, void* stream);

/* ---- LayerNorm -- nn.LayerNorm at mmseg/models/backbones/mix_transformer.py:76 (sr norm, eps 1e-5), :123,:136 (Block, 1e-6),
 * :175 (patch embed, 1e-5), :270-318 (stage norms).  bwd: dx = [dres +] LN'(dy); dgamma/dbeta accumulated.
 * ws: cmda_layernorm_bwd_ws_floats() floats, ZERO on entry; the call leaves it zeroed again (reusable without a memset). */
int cmda_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
    int64_t rows, int C, float eps, int dtype, void* stream);
int64_t cmda_layernorm_bwd_ws_floats(int64_t rows, int C);
int cmda_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
    void* dx, float* dgamma, float* dbeta, float* ws, int64_t rows, int C, const float* out_scale, int64_t rows_per_scale,
    void* dx_scaled, int dtype, void* stream);
/* dgamma == NULL (deferred): the per-block partial sums stay in `ws` -- then a zero-initialised workspace owned by this LayerNorm
 * layer alone -- until cmda_layernorm_fold_batch folds MANY layers' workspaces into their dgamma / dbeta in one launch
 * (desc: DEVICE array of n 32-byte records {float* ws; float* dgamma; float* dbeta; int32 C; int32 nslots}; nslots =
 * cmda_layernorm_slots(); max_c = largest C).  The workspaces are left zeroed. */
/* Mixed storage types -- the fp32 RESIDUAL STREAM of the bf16 mode: `x + drop_path(...)` of Block.forward (mix_transformer.py:134,146)
 * accumulates over up to 52 blocks, so x stays fp32 while everything the GEMMs read is bf16.  fwd2: x in x_dtype -> y in y_dtype
 * (any of the four combinations); bwd2: dy / dres / dx / dx_scaled in `dtype`, the saved input x in x_dtype (fp32 or `dtype`). */
int cmda_layernorm_fwd2(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype, float* mean, float* rstd,
    int64_t rows, int C, float eps, void* stream);
int cmda_layernorm_bwd2(const void* dy, const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd,
    const void* dres, void* dx, float* dgamma, float* dbeta, float* ws, int64_t rows, int C, const float* out_scale,
    int64_t rows_per_scale, void* dx_scaled, int dtype, void* stream);
int cmda_layernorm_fold_batch(const void* desc, int n, int max_c, void* stream);
int cmda_layernorm_slots(void);
/* out_scale / rows_per_scale / dx_scaled (all or none): second output dx * out_scale[row / rows_per_scale] -- the per-sample
 * DropPath factor (timm DropPath, mix_transformer.py:145-146) of the residual branch that consumes this gradient next. */

/* ---- Row softmax of attention scores -- `attn.softmax(dim=-1)` mix_transformer.py:97-98 (in place, alpha = head_dim^-0.5);
 * bwd writes dS = alpha * P * (dP - sum P dP) over dP. */
int cmda_softmax_fwd(void* s, int64_t rows, int L, float alpha, int dtype, void* stream);
int cmda_softmax_bwd(const void* p, void* dp, int64_t rows, int L, float alpha, int dtype, void* stream);

/* ---- Fused attention core (bf16, head_dim 64, Nk <= 256) -- Attention.forward mix_transformer.py:86-103:
 * softmax(q k^T * scale) v per (batch, head) without materialising the [N, Nk] scores.  q [B*N, C] (head h = columns
 * 64h..), kv [B*Nk, 2C] (K at column 64h, V at C + 64h), o [B*N, C].  bwd recomputes the probabilities; dq [B*N, C] is
 * written, dkv32 [B*Nk, 2C] fp32 is ACCUMULATED into (atomics; caller zeroes).  CMDA_ERR_UNSUPPORTED outside these
 * limits (the caller then uses the GEMM + cmda_softmax path). */
int cmda_attention_fwd(const void* q, const void* kv, void* o, int B, int N, int Nk, int heads, int C, float scale,
    int dtype, void* stream);
int64_t cmda_attention_bwd_ws_floats(int B, int N, int heads);   /* size of `stats` (per-query log-sum-exp and D) */
/* dK | dV: accumulated into dkv32 (fp32 [B*Nk, 2C], zero on entry) -- or, when cmda_attention_bwd_direct(...) is 1 and dkv16 is
 * given, stored as bf16 [B*Nk, 2C] into dkv16 by one block per key slice (no workspace, no atomics; dkv32 may then be NULL). */
int cmda_attention_bwd_direct(int B, int N, int Nk, int heads);
int cmda_attention_bwd(const void* q, const void* kv, const void* d_o, void* dq, float* dkv32, void* dkv16, float* stats, int B,
    int N, int Nk, int heads, int C, float scale, int dtype, void* stream);
/* (ABI 8) the same three kernels for the SPLIT-bf16 mode (fp32 storage; every product as three bf16 MFMAs on hi / lo operands, the
 * arithmetic of cmda_gemm with dtype CMDA_F32X3): mix_transformer.py:97-101 of the tolerance-meeting mode without the [N, Nk]
 * probabilities in HBM.  q / o / d_o / dq fp32 [B*N, C]; kv_hi / kv_lo = cmda_split_bf16 of the fp32 kv [B*Nk, 2C] (split once by the
 * caller: every query block of a (batch, head) reads the same K / V); dK | dV fp32: accumulated into dkv32 (zero on entry) or, in the
 * direct mode of cmda_attention_bwd_direct, stored into dkv_direct.  head_dim 64, Nk <= 256, C % 4 == 0. */
int cmda_attention_fwd_x3(const float* q, const void* kv_hi, const void* kv_lo, float* o, int B, int N, int Nk, int heads, int C,
    float scale, void* stream);
int cmda_attention_bwd_x3(const float* q, const void* kv_hi, const void* kv_lo, const float* d_o, float* dq, float* dkv32,
    float* dkv_direct, float* stats, int B, int N, int Nk, int heads, int C, float scale, void* stream);

/* ---- Depthwise 3x3 convolution, NHWC -- DWConv(+GELU) of MixFFN mix_transformer.py:37-44,443-455 and the dilated depthwise
 * half of the sep-ASPP decode_heads/sep_aspp_head.py:18-27.  `w` is tap-major fp32 [9][C]; dw (gradient) is [C][9]. */
int cmda_dwconv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int dil,
    int act, int dtype, void* stream);
/* cmda_dwconv3x3_fwd (no activation, dil >= 2: the sep-ASPP's dilated depthwise convolutions, sep_aspp_head.py:18-27) with the batch
 * statistics of the BatchNorm behind it taken on the way: column sums / sums of squares of y per group of `imgs_per_group`
 * consecutive images are added to `colstats`, the BatchNorm workspace (groups x cmda_bn_ws_floats(C) floats, zero on entry;
 * cmda_bn_train_fwd with ws_has_stats = 1 finishes from it).  CMDA_ERR_UNSUPPORTED for dil < 2. */
int cmda_dwconv3x3_fwd_stats(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int dil,
    float* colstats, int imgs_per_group, int dtype, void* stream);
int cmda_dwconv3x3_gelu_bwd_prep(const void* x, const float* w, const float* bias, const void* da, void* dz, int B,
    int H, int W, int C, int dil, int dtype, void* stream);
int cmda_dwconv3x3_bwd_data(const void* dy, const float* w, void* dx, int B, int H, int W, int C, int dil, int
    accumulate, int dtype, void* stream);
int cmda_dwconv3x3_bwd_weight(const void* dz, const void* x, float* dw, float* dbias, int B, int H, int W, int C, int
    dil, int dtype, void* stream);
/* gelu_bwd_prep + bwd_weight of the MixFFN DWConv (mix_transformer.py:37-44,443-455 under autograd) in one pass over x / da:
 * dz written, dw [C][9] and dbias [C] accumulated. */
int cmda_dwconv3x3_gelu_bwd_fused(const void* x, const float* w, const float* bias, const void* da, void* dz, float* dw,
    float* dbias, int B, int H, int W, int C, int dil, int dtype, void* stream);

/* ---- Bilinear resize (align_corners=False), fused with the channel-concat write -- resize() + torch.cat at
 * decode_heads/daformer_head.py:263-275 (ops/wrappers.py:9-28).  y/dy is a channel slice [coff,coff+C) of rows of pitch ldy. */
int cmda_bilinear_fwd(const void* x, void* y, int B, int IH, int IW, int OH, int OW, int C, int ldy, int coff, int
    dtype, void* stream);
int cmda_bilinear_bwd(const void* dy, void* dx, int B, int IH, int IW, int OH, int OW, int C, int ldy, int coff, int
    dtype, void* stream);

/* ---- Train-mode BatchNorm2d (+ReLU) -- mmcv ConvModule's norm/activate in decode_heads/daformer_head.py:46-62,
 * aspp_head.py:33-43, sep_aspp_head.py:18-27 (batch statistics, running-stat update, eps 1e-5, momentum 0.1).
 * groups > 1: x / y / dy / dx hold `groups` consecutive blocks of M rows, each normalised with ITS OWN batch statistics;
 * mean / rstd are [groups][C]; running statistics receive the groups' updates one after the other in `order` (HOST int
 * array of `groups` entries, NULL = 0,1,2,..) -- the shared decoder of daformer_head.py:254-258,305-319 run once over the
 * image / events / fusion / ISR features instead of four times.  Also used with groups = samples (no running statistics)
 * as InstanceNorm2d for cyclegan/cyclegan_model.py:339-374.  ws: groups * cmda_bn_ws_floats(C) floats of scratch.
 * groups <= 8.  ws_has_stats = 1: ws already holds the column sums of x, accumulated by cmda_gemm's `colstats` epilogue of the
 * convolution that produced x (no reduction pass over x; the sums are taken without the per-channel shift of the reduction kernel);
 * ws is left ZERO again, so one zero-initialised workspace per stream serves every such layer. */
int64_t cmda_bn_ws_floats(int C);
int cmda_bn_train_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, float*
    running_mean, float* running_var, float* ws, int64_t M, int C, float eps, float momentum, int relu, int ldy, int
    coff, int groups, const int* order, int ws_has_stats, int dtype, void* stream);
/* Mixed storage types: x in x_dtype, y in y_dtype; res32 (optional fp32 [groups*M, C]) is added AFTER the normalisation (+ReLU) and
 * y2_bf16 (optional bf16 [groups*M, C]) receives a second copy of the result -- ResnetBlock of the Motion-Extractor generator in the
 * bf16 mode (cyclegan/cyclegan_model.py:377-434: out = x + conv_block(x)): the convolution output and the residual stream stay fp32,
 * the bf16 copy is the next convolution's operand. */
int cmda_bn_train_fwd2(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype, float* mean,
    float* rstd, float* running_mean, float* running_var, float* ws, int64_t M, int C, float eps, float momentum, int relu, int ldy,
    int coff, int groups, const int* order, const float* res32, void* y2_bf16, int ws_has_stats, void* stream);
int cmda_bn_apply(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, void* y,
    int64_t M, int C, int relu, int ldy, int coff, int dtype, void* stream);
int cmda_bn_train_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const
    float* beta, void* dx, float* dgamma, float* dbeta, float* ws, int64_t M, int C, int relu, int lddy, int coff, int
    groups, int dtype, void* stream);

/* ---- Fused up-sample + cross-entropy + accuracy -- BaseDecodeHead(Fusion).losses decode_heads/decode_head.py:588-606
 * (resize -> F.cross_entropy(reduction='none', ignore_index) losses/cross_entropy_loss.py:21-26 -> x weight -> mean over
 * all pixels losses/utils.py:60-69; accuracy losses/accuracy.py:40-50).  acc[0] += sum w*nll, acc[1] += #correct.
 * Teacher: fused up-sample + softmax-max + threshold count + pseudo-weight, uda/dacs.py:674-682,701-711;
 * cmda_upsample_logits_nchw = encode_decode's final resize, segmentors/encoder_decoder.py:733-745. */
int cmda_ce_upsample_fwd(const float* logits, const int64_t* label, const float* weight, float* lse_out, float* acc,
    int B, int h, int w, int H, int W, int nc, int ignore_index, void* stream);
int cmda_ce_upsample_bwd(const float* logits, const int64_t* label, const float* weight, const float* lse, const
    float* gscale_ptr, float gscale_mul, float* dlogits, int B, int h, int w, int H, int W, int nc, int ignore_index,
    void* stream);
int cmda_pseudo_label(const float* logits, int64_t* label_out, float* prob_out, int* count, int B, int h, int w, int
    H, int W, int nc, float thr, void* stream);
int cmda_pseudo_weight(const int* count, float* weight, int B, int H, int W, int top, int bottom, void* stream);
int cmda_upsample_logits_nchw(const float* logits, float* out, int B, int h, int w, int H, int W, int nc, void*
    stream);

/* ---- Data movement / pointwise -- permute+cast (NLC<->NCHW boundary copies mix_transformer.py:406-430 and conv-weight
 * repacks), bias gradient, a*x+b*y (fusion/attention_avg_fusion.py:49), DropPath / Dropout2d scaling (timm DropPath
 * mix_transformer.py:134,145-146; nn.Dropout2d decode_head.py:565-566), strided 2-D copy (torch.cat fusion/attention_fusion.py:52). */
int cmda_permute4(const void* src, void* dst, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3, int
    flipmask, int accumulate, int src_dtype, int dst_dtype, void* stream);
/* one launch for many re-layouts (all conv / depthwise weights after an optimizer or EMA step): desc = DEVICE array of
 * cmda_permute_desc_t, blocks = DEVICE int32 [nblocks][2] {tensor index, 1024-element chunk index}; fp32 sources. */
typedef struct cmda_permute_desc_t {
  const float* src;
  void* dst;
  int32_t d[4];       /* source dims */
  int32_t p[4];       /* dst axis a = source axis p[a] */
  int32_t flipmask;   /* bit ax set: source axis ax reversed; bits 8-10 = PADDED source axis + 1 and bits 16.. = its real extent: d[] then
                         holds the padded dims of the destination layout, entries past the real extent are written as zero (drain mode: the
                         shadow is padded, the gradient is not) */
  int32_t dst_bf16;   /* 1: dst is bf16, 0: fp32; 2: DRAIN -- dst (fp32) += src and src = 0 for d = (Co,KH,KW,Ci), p = (0,3,1,2): the conv
                         weight-gradient shadows in GEMM order -> the parameter's own [Co][Ci][KH][KW] layout */
  int64_t total;      /* d[0]*d[1]*d[2]*d[3] */
} cmda_permute_desc_t;
/* hi = bf16(src), lo = bf16(src - hi) (n % 4 == 0, bf16 outputs): the operand pair of the split-bf16 mode (CMDA_F32X3) written once, so that
 * a LARGE contraction of that mode -- the decode head's 3x3 bottleneck and its gradients, the generator's convolutions
 * (decode_heads/daformer_head.py:63-79, cyclegan/cyclegan_model.py:339-374) -- runs as three launches of the bf16 LDS-DMA kernels
 * (a_lo b_hi + a_hi b_lo + a_hi b_hi, fp32 output accumulated with beta = 1) instead of the register-staged split kernel. */
int cmda_split_bf16(const float* src, void* hi, void* lo, int64_t n, void* stream);
/* dst (activation dtype) = src; src (fp32, n % 4 == 0) = 0: drains a persistent accumulation workspace and leaves it zeroed */
int cmda_cast_clear(float* src, void* dst, int64_t n, int dst_dtype, void* stream);
int cmda_permute4_batch(const void* desc, const int* blocks, int nblocks, void* stream);
/* fp32 [rows, c] -> activation dtype [rows, cp >= c], zero-padded columns: the logit gradient of cls_seg's backward
 * (decode_head.py:563-586 under autograd) padded from 19 to 32 classes */
int cmda_cast_pad_cols(const float* src, void* dst, int64_t rows, int c, int cp, int dst_dtype, void* stream);
/* out[r][c] = bias[c] (or 0), fp32 [rows, C]: the accumulator of a split-K convolution -- the spatial-reduction convolution of
 * Attention (mix_transformer.py:73-75: B*256 output rows, K up to 4096) runs as 8 K-slices accumulating with atomics on top of it */
int cmda_rows_fill(float* out, const float* bias, int64_t rows, int C, void* stream);
/* NCHW fp32 image -> NHWC rows of `cpad` >= C channels (zeros past C) in the activation dtype: the input of OverlapPatchEmbed
 * (mix_transformer.py:169-183) with its 3 channels padded to 8, so that the 7x7 / stride-4 convolution runs on the LDS-DMA GEMM path */
int cmda_nchw_to_nhwc_pad(const float* src, void* dst, int B, int C, int64_t HW, int cpad, int dst_dtype, void* stream);
int cmda_colsum(const void* x, float* out, int64_t M, int N, int64_t ld, int dtype, void* stream);
int cmda_axpby(const void* x, const void* y, void* out, float a, float b, int64_t n, int dtype, void* stream);
int cmda_sample_scale(const void* x, const float* scale, void* out, int B, int64_t per_sample, int C, int per_channel,
    int dtype, void* stream);
int cmda_copy2d(const void* src, void* dst, int64_t rows, int cols, int64_t src_ld, int64_t dst_ld, int dtype, void*
    stream);

/* ---- Self-training state -- EMA teacher update uda/dacs.py:261-272; fused AdamW (torch.optim.AdamW semantics,
 * configs/_base_/schedules/adamw.py; optional bf16 copy of the updated weights); ClassMix of image/events/weight and of
 * labels, models/utils/dacs_transforms.py:101-131 (classes: int64 [B,max_classes] padded with -1). */
int cmda_ema_update(float* ema, const float* param, float alpha, int64_t n, void* ema_bf16 /* optional bf16 copy of the result */,
    void* stream);
int cmda_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
    float beta2, float eps, float weight_decay, int step, void* stream);
int cmda_class_mix(const void* src, const void* tgt, void* out, const int64_t* src_label, const int64_t* classes, int
    max_classes, int B, int HW, int Cch, int channels_last, int dtype, void* stream);
int cmda_class_mix_label(const int64_t* src, const int64_t* tgt, int64_t* out, const int64_t* src_label, const
    int64_t* classes, int max_classes, int B, int HW, void* stream);

/* ---- Extractors -- Image Content-Extractor (ISR) datasets/utils.py:87-152 as called in the step uda/dacs.py:729-744
 * (mean3/std3 are HOST pointers; lut = fp32[256] log table; dirs = int32[ndir][2] (dy,dx); mm = uint32[B*ndir*4] scratch)
 * and the event voxel grid datasets/dsec.py:26-70 + events_norm :80-121 (ws = 40 bytes of scratch, 8-byte aligned). */
int cmda_isr_gray(const float* img, uint8_t* gray, int B, int H, int W, const float* mean3, const float* std3, void*
    stream);
int cmda_isr_from_gray(const uint8_t* gray, const float* lut, const int* dirs, int ndir, uint32_t* mm, float* out, int
    B, int H, int W, float threshold, float clip, void* stream);
int cmda_events_to_voxel_grid(const float* t, const float* x, const float* y, const float* pol, float* grid, int64_t
    N, int bins, int H, int W, void* stream);
int cmda_events_norm(const float* events, float* out, void* ws, int64_t n, float clip_range, float final_range, void*
    stream);

/* ---- Strong augmentation of the mixed image -- models/utils/dacs_transforms.py:64-98 (kornia 0.5.8 ColorJitter and
 * GaussianBlur2d, restated), applied per sample by uda/dacs.py:721-724.  mean3/std3 are HOST pointers; prm is DEVICE
 * fp32 [B][8] (per sample: op order[4], f_brightness, f_contrast, f_saturation, f_hue); taps_x / taps_y are DEVICE fp32
 * normalised Gaussians (kernel size ky from H, kx from W); enable is a DEVICE int gate (NULL = on; 0 = leave the image
 * untouched) so that the random on/off decisions of dacs.py:446-456 do not change the launch sequence. */
int cmda_color_jitter(float* img, int B, int H, int W, const float* mean3, const float* std3, const float* prm,
    const int* enable, void* stream);
int cmda_gaussian_blur(float* img, float* tmp, const float* taps_x, const float* taps_y, int planes, int H, int W, int kx,
    int ky, const int* enable, void* stream);

/* ---- Loader-side preprocessing on the device -- mmseg/datasets/dsec.py:189-339 (target __getitem__:
 * crop -> flip -> PIL BILINEAR resize -> ToTensor/Normalize; 'L' luma for the real-time ISR), :341-366 (get_events_vg: rectify
 * gather, t normalisation), :314-322 (crop / flip / F.interpolate of the voxel grid); mmseg/datasets/cityscapes_ic.py:147-210
 * (source: resize -> crop -> flip); create_cityscapes_image_change.py:16-35 (time residual PNG).
 * cmda_pil_resize_u8: src uint8 [B][IH][IW][C] (HWC, C = 1 or 3); samp = DEVICE int32 [B][8] {src_x0, src_y0, flip_src, out_x0,
 *   out_y0, flip_out, 0, 0}; the resize sees the in_w x in_h window at (src_x0, src_y0), mirrored when flip_src; the OW x OH window
 *   of the resized image at (out_x0, out_y0), mirrored when flip_out, is produced.  hbounds/hkk/vbounds/vkk = Pillow's coefficient
 *   tables (DEVICE int32: bounds [out][2] = {first tap, tap count}, kk [out][ksize] fixed point with 22 fraction bits), built by the
 *   caller exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do.  tmp: B*in_h*OW*C bytes.  Outputs (each may be NULL):
 *   out_u8 [B][OH][OW][C]; out_f fp32 NCHW [B][3 if (C==1 && rep3) else C][OH][OW] = (u8/255 - fshift3[c]) / fscale3[c] (HOST
 *   pointers: torchvision ToTensor + Normalize(mean=fshift3, std=fscale3)); out_gray [B][OH][OW] = PIL 'L' luma (C == 3).
 * cmda_time_residual_u8: uint8 'L' frames now / front [B][H][W], lut = fp32[256] log(g + log_add); mm = uint32 [B][4] scratch.
 * cmda_event_prep: raw events (t int64 us, x / y int32 pixel, p uint8) -> t_norm, rectified x / y (rect_map fp32 [H][W][2] or NULL),
 *   polarity as fp32, the inputs of cmda_events_to_voxel_grid.
 * cmda_crop_flip_resize_f32: in fp32 [B][C][IH][IW], window cw x ch at samp[b].{src_x0,src_y0}, mirrored when flip_src, bilinear
 *   (align_corners=False) to OH x OW, every channel written `rep` times (enforce_3_channels). */
int cmda_pil_resize_u8(const uint8_t* src, int B, int IH, int IW, int C, const int* samp, int in_w, int in_h, const int* hbounds,
    const int* hkk, int hksize, const int* vbounds, const int* vkk, int vksize, int OW, int OH, uint8_t* tmp, uint8_t* out_u8,
    float* out_f, uint8_t* out_gray, const float* fscale3, const float* fshift3, int rep3, void* stream);
int cmda_luma_u8(const uint8_t* rgb, uint8_t* out, int64_t npix, void* stream);   /* PIL convert('L') of interleaved uint8 RGB */
int cmda_time_residual_u8(const uint8_t* now, const uint8_t* front, const float* lut, uint32_t* mm, uint8_t* out, int B, int H,
    int W, float threshold, float clip, void* stream);
int cmda_event_prep(const int64_t* t, const int32_t* x, const int32_t* y, const uint8_t* p, const float* rect_map, int H, int W,
    float* t_norm, float* xr, float* yr, float* pol, int64_t N, void* stream);
int cmda_crop_flip_resize_f32(const float* in, float* out, const int* samp, int B, int C, int IH, int IW, int cw, int ch, int OH,
    int OW, int rep, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CMDA_HIP_H_ */
