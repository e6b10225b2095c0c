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
  int32_t conv;         /* 0 plain, 1 im2col view */
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
} cmda_gemm_params_t;

int cmda_gemm(const cmda_gemm_params_t* p, void* stream);

/* ---- LayerNorm (mix_transformer.py:76,123,136,175,270-318) ---- */
int cmda_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                       int64_t rows, int C, float eps, int dtype, void* stream);
int cmda_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                       const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int C, int dtype,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CMDA_HIP_H_ */
