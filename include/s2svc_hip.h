/*
 * s2svc_hip.h -- C ABI of libs2svc_hip.so: the MI355X (gfx950) kernels behind the seq2seq-vc
 * hot path (VTN / AAS-VC / Transformer-TTS forward+backward, AAS alignment search, losses,
 * optimiser, STFT->log-mel).
 *
 * The reference (unilight/seq2seq-vc) has no FFI layer: its hot path is stock torch ops called
 * from seq2seq_vc/modules/ and seq2seq_vc/losses/.  Each entry point below therefore names the
 * reference call site(s) it replaces (paths relative to the reference root).  All pointers are
 * DEVICE pointers unless marked "host"; `stream` is a hipStream_t passed as void*.  Every function
 * returns 0 on success, <0 on error (s2svc_last_error() gives the message).  No function
 * synchronises the device or allocates memory: callers own all buffers.
 *
 * dtype codes: 0 = float32, 1 = bfloat16 (raw uint16 storage).  Reductions, softmax/LN/BN
 * statistics and accumulators are always fp32 (MAS: fp64).
 */
#ifndef S2SVC_HIP_H
#define S2SVC_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* s2svc_last_error(void);
int s2svc_abi_version(void);

/* ------------------------------------------------------------------------------------------ */
/* Generic tiled MFMA GEMM with implicit-convolution operand addressing.                      */
/* C[m,n] = act(alpha * sum_k A(m,k) * B(n,k) + bias[n]) + res[m,n]                           */
/* replaces: torch.nn.Linear / Conv1d / Conv2d / matmul call sites, e.g.                      */
/*   modules/transformer/attention.py:54-56,63,88,110 (QKV/out Linear, QK^T, PV)              */
/*   modules/transformer/positionwise_feed_forward.py:30-32                                   */
/*   modules/pre_postnets.py:63-66,108-185 (Prenet Linear, Postnet Conv1d)                    */
/*   modules/transformer/subsampling.py:58-70 (Conv2d 3x3 stride 2, Linear)                   */
/*   modules/alignments.py:21-26 (AlignmentModule Conv1d)                                     */
/* and their autograd backward (dgrad / wgrad are the same kernel with other operand layouts) */
/* ------------------------------------------------------------------------------------------ */
enum { S2SVC_LAYOUT_KC = 0,   /* element (r,k) at r*ld + k  (reduction index contiguous)   */
       S2SVC_LAYOUT_RC = 1 }; /* element (r,k) at k*ld + r  (row index contiguous)          */
enum { S2SVC_OP_DENSE = 0,
       S2SVC_OP_CONV1D = 1,   /* rows/reduction index m=(b,t); taps j: x[(m+j-pad)*ld + c], valid iff 0<=t+j-pad<T */
       S2SVC_OP_CONV2D_S2 = 2 /* NHWC input (B,T1,F1,C), 3x3 stride 2 no padding, m=(b,t2,f2)  */ };

typedef struct {
  const void* ptr;
  int64_t ld;
  int32_t layout;          /* S2SVC_LAYOUT_* */
  int32_t mode;            /* S2SVC_OP_*     */
  int32_t C;               /* conv: channels per tap (the implicit index is tap*C + c)        */
  int32_t T;               /* conv1d: frames per batch item                                   */
  int32_t pad;             /* conv1d: left padding (kernel-1)/2                               */
  int32_t T1, F1, T2, F2;  /* conv2d: input and output spatial dims                           */
  int64_t bs0, bs1;        /* two-level batch strides (elements)                              */
} s2svc_operand;

typedef struct {
  s2svc_operand A, B;
  void* C;
  int64_t ldc, cbs0, cbs1;
  int32_t c_dtype;         /* dtype of C / res                                                */
  const float* bias;       /* [N] fp32 or NULL                                                */
  const void* res;         /* residual added after activation, same dtype as C, or NULL       */
  int64_t ldr, rbs0, rbs1;
  int32_t M, N, K;
  int32_t nb0, nb1;        /* batch count = nb0*nb1 (>=1)                                     */
  int32_t act;             /* 0 none 1 relu 2 tanh 3 swish 4 sigmoid 5 gelu                   */
  float alpha;
  int32_t dtype;           /* dtype of A and B                                                */
  int32_t accumulate;      /* 1: C += result (C read in c_dtype)                              */
  int32_t splitk;          /* >1: partial sums go to ws[splitk][batch][M][N] fp32             */
  float* ws;
} s2svc_gemm_desc;

int s2svc_gemm(const s2svc_gemm_desc* desc /* host */, void* stream);

#ifdef __cplusplus
}
#endif
#endif
