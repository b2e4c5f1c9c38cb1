/*
 * s2svc_hip.h -- C ABI of libs2svc_hip.so: the MI355X (gfx950) kernels behind the seq2seq-vc
 * hot path (VTN / AAS-VC / Transformer-TTS forward+backward, AAS alignment search, losses,
 * optimiser, STFT->log-mel).
 *
 * The reference (unilight/seq2seq-vc) has no FFI layer: its hot path is stock torch ops called
 * from seq2seq_vc/modules/ and seq2seq_vc/losses/.  Each entry point below therefore names the
 * reference call site(s) it replaces (paths relative to the reference root, tag 2024_08_07).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless marked "host"; `stream` is a hipStream_t passed as void*;
 *   - every function returns 0 on success, <0 on error (s2svc_last_error() gives the message);
 *   - no function synchronises the device or allocates memory: callers own all buffers;
 *   - dtype codes: 0 = float32, 1 = bfloat16 (raw uint16 storage).  Activations are channel-last,
 *     contiguous (B, T, D).  Reductions / softmax / LN / BN statistics are fp32, MAS is fp64;
 *   - masks never exist as tensors: kernels take int32 per-utterance length vectors;
 *   - dropout: a mask is a pure function of (seed, element index) (SplitMix64 output function over seed + index/4, 16 bits per element); kernels read
 *     seed = *seed_base + seed_off, seed_base in device memory (may be NULL), so a captured hipGraph
 *     draws fresh masks on every replay and backward kernels regenerate masks instead of loading them.
 *   - absent rows (`int Tn, const int32_t* vlens`, ABI version 2): a captured training step allocates a (B, Tn, C) activation at a
 *     PADDED length Tn, the reference computes on the batch cropped to its longest utterance (models/vtn.py:208-214,
 *     models/aas_vc.py:523-524).  Entry points that mix along time or reduce over the batch take the B int32 lengths `vlens` in
 *     device memory (graph DATA): row r = b * Tn + t with t >= vlens[b] is ABSENT -- excluded from every sum and count, read
 *     as a convolution's zero padding, written as zero (outputs and data gradients).  vlens = NULL: every row is present.
 */
#ifndef S2SVC_HIP_H
#define S2SVC_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* s2svc_last_error(void);
int s2svc_abi_version(void);
/* Measurement aid: launches `workgroups` x `threads` of a kernel that stores one word per workgroup into `sink_1024_words`
   (4 KB of device memory).  What a DEPENDENT launch costs on this stack whatever it does (bench.py "launch_floor_us"). */
int s2svc_launch_floor(int workgroups, int threads, void* sink_1024_words, void* stream);

/* Hand-off points inside a captured graph (the data-parallel backward pass: apex DDP's per-bucket hooks, bin/vc_train.py:423-431, have
   no counterpart in a graph): _record on a CAPTURING stream adds an event-record node (hipEventRecordExternal; returns 1), on an
   ordinary stream it is hipEventRecord (returns 0); _stream_wait_event issued after the graph's launch makes `stream` wait for that
   point of that launch.  `ev` is an opaque handle from _event_create (host memory). */
int s2svc_event_create(void** out /* host */);
int s2svc_event_destroy(void* ev);
int s2svc_event_record(void* ev, void* stream);
int s2svc_stream_wait_event(void* stream, void* ev);

/* ========================================================================================== */
/* Generic tiled MFMA GEMM with implicit-convolution operand addressing.                      */
/*   C[z][m,n] = act(alpha * sum_k A_z(m,k) * B_z(n,k) + bias[n]) + res_z[m,n]                 */
/* replaces: torch.nn.Linear / Conv1d / Conv2d / matmul call sites, e.g.                      */
/*   modules/transformer/attention.py:54-56,63,88,110 (QKV/out Linear, QK^T, PV)              */
/*   modules/transformer/positionwise_feed_forward.py:30-32                                   */
/*   modules/pre_postnets.py:63-66,108-185 (Prenet Linear, Postnet Conv1d)                    */
/*   modules/transformer/subsampling.py:58-70 (Conv2d 3x3 stride 2, Linear)                   */
/*   modules/alignments.py:21-26 (AlignmentModule Conv1d), length_regulator.py:153 (matmul)   */
/*   bin/preprocess.py:63-83 (STFT as windowed-DFT GEMM with overlapping rows, mel basis)     */
/* and their autograd backward (dgrad / wgrad are the same kernel with other operand layouts) */
/* ========================================================================================== */
enum { S2SVC_LAYOUT_KC = 0,   /* element (r,k) at r*ld + k  (reduction index contiguous)   */
       S2SVC_LAYOUT_RC = 1 }; /* element (r,k) at k*ld + r  (row index contiguous)          */
enum { S2SVC_OP_DENSE = 0,
       S2SVC_OP_CONV1D = 1,   /* rows/reduction index m=(b,t); taps j: x[(m+j-pad)*ld + c], valid iff 0<=t+j-pad<T */
       S2SVC_OP_CONV2D_S2 = 2,/* NHWC input (B,T1,F1,C), 3x3 stride 2 no padding, m=(b,t2,f2)  */
       /* A only, bf16 only: DATA gradient of that convolution for ONE parity class (pt, pf) of input pixels
          (t1, f1) = (2i+pt, 2j+pf): rows m = (b, i, j) over the class grid T1 x F1 (= ceil((T_in-pt)/2) x ceil((F_in-pf)/2)),
          reduction index k = tap*C + o over the (2-pt)*(2-pf) taps of the class, tap = ta*(2-pf) + fb reads the output-
          gradient pixel (i-ta, j-fb) of the (B,T2,F2,C) tensor (zero outside); pad = 2*pt + pf.  The weight operand is
          the class matrix s2svc_tconv2d_weights() lays out; use it with the c_map of s2svc_gemm_desc. */
       S2SVC_OP_TCONV2D_S2 = 3 };

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
  int32_t zero_padded;     /* dense: the vectorised extent (k for KC, row for RC) is not a multiple of 16 bytes
                              but every row is padded with ZEROS up to one (ld covers it); lets the fast
                              kernels read whole vectors without masking                                */
  int32_t reserved_;
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
  /* optional fused row sums of A: a_rowsum[m] (+)= sum_k A(m,k)  (the bias gradient when A = dY^T in a   */
  /* wgrad GEMM); needs nb0*nb1 == 1; a_rowsum_ws: >= splitk*M floats when splitk > 1                     */
  float* a_rowsum;
  float* a_rowsum_ws;
  int32_t a_rowsum_accumulate;
  int32_t tile_hint;       /* 0 = auto, 64 or 128: output tile edge chosen by the caller's cost model */
  /* optional epilogue stage between the activation and the residual (needs nb0*nb1 == 1, ldc == N):               */
  /*   v *= dropout_scale(*seed_base + seed_off, m*N + n, drop_p)   -- the mask the standalone dropout kernels draw */
  /*   v  = emask[m*ldm + n] > 0 ? v : 0                            -- emask has the dtype of C                     */
  /* fuses `dropout(relu(x W1^T))` into the forward GEMM and `dY W2 * dropmask * relu'(h)` into the dgrad GEMM of   */
  /* positionwise_feed_forward.py:30-32, and relu' of subsampling.py:58-60 into the dgrad of the Linear after it.  */
  /*   emask_mode 1: v *= swish'(emask[m*ldm + n])  -- emask holds the PRE-activation of the forward pass (the data-gradient  */
  /*   GEMM through w_2 of a Swish feed-forward block then needs no element-wise pass either)                                  */
  const void* emask;
  int64_t ldm;
  float drop_p;
  int32_t emask_mode;
  const uint64_t* seed_base;
  uint64_t seed_off;
  /* c_map = 1 (bf16 LDS-DMA kernels only; no res / dropout / batch / split-K; emask: mode 0 only, indexed like C, i.e. by the
     MAPPED row -- relu' of the layer below fused into the transposed convolution): GEMM row m = (b, i, j) over the class grid
     cm_Tc x cm_Fc is stored at row (b*cm_T1 + 2i+cm_pt)*cm_F1 + 2j+cm_pf of C -- the four parity classes of a stride-2
     transposed convolution write one NHWC tensor (B, cm_T1, cm_F1, N) without a col2im pass */
  int32_t c_map;
  int32_t cm_T1, cm_F1, cm_Tc, cm_Fc, cm_pt, cm_pf;
  int32_t reserved3_;
  /* optional second output: alpha * A.B^T + bias BEFORE the activation (same dtype, ld and batch strides as C; no c_map) -- the
     pre-activation a Swish / GELU backward pass needs, written by the GEMM whose epilogue applies activation + dropout */
  void* c_pre;
} s2svc_gemm_desc;

int s2svc_gemm(const s2svc_gemm_desc* desc /* host */, void* stream);

/* Weight matrices of the four parity classes of the 3x3 stride-2 transposed convolution (S2SVC_OP_TCONV2D_S2), from the
   fp32 master weight w (O, C, 3, 3): out (bf16) = class (0,0) | (0,1) | (1,0) | (1,1), class (pt, pf) = [C][ntaps*O] with
   element [c][tap*O + o] = w[o, c, pt + 2*ta, pf + 2*fb], tap = ta*(2-pf) + fb; offsets 0, 4*C*O, 6*C*O, 8*C*O; 9*C*O total.
   Replaces the `col2im` half of autograd's conv2d input gradient (subsampling.py:58-63). */
int s2svc_tconv2d_weights(int O, int C, const float* w, void* out_bf16, void* stream);

/* Grouped launch of independent weight-gradient GEMMs  dW[N_out, N_in] (+)= dY^T . X  (bf16, both operands dense and
   row-contiguous -- what the backward of every Linear / 1x1 Conv1d issues; replaces the per-layer torch.nn.Linear weight
   gradients autograd computes one by one behind trainers/ar_vc.py:99 `loss.backward()`).  The host queues the
   descriptors of a few consecutive layers during backward and launches them as ONE grid: every workgroup runs the whole
   reduction of its output tile, so there is no split-K workspace and no reduction pass.
     _ok : 1 if `desc` can join a group (else launch it with s2svc_gemm);
     s2svc_gemm_grouped : `descs` is a HOST array; descriptors travel by value in the kernel arguments (10 per launch,
           n problems take ceil(n / 10) launches; hipGraph capture records them with the nodes), `tile` = 64 or 128 is the
           output tile edge, every descriptor must have splitk <= 1.
   Two descriptors of one call must not write the same C / a_rowsum (they run concurrently). */
int s2svc_gemm_grouped_ok(const s2svc_gemm_desc* desc /* host */);
int s2svc_gemm_grouped(const s2svc_gemm_desc* descs /* host */, int n, int tile, void* stream);
/* BATCHED problems of one operand-kind pair (A K-contiguous or row-contiguous, B row-contiguous; bf16, no split-K) as one
   grid: the batched products of an attention backward pass (ops/functional.py: _attn_common_bwd, _RelAttnPacked.backward;
   reference modules/transformer/attention.py:72-111, 262-305 differentiated).  0 = launched, 1 = not eligible as a group
   (nothing launched: run them with s2svc_gemm), < 0 = error. */
int s2svc_gemm_grouped_batched(const s2svc_gemm_desc* descs /* host */, int n, void* stream);
/* RAGGED weight gradients on the 8-wave kernel (csrc/gemm_8ph.hip, "W8"): C[M, N] fp32 (+)= A^T . B for dense row-contiguous bf16
   operands with M % 8 == N % 8 == 0 and ANY K (the Linear weight gradients dW = dY^T . X of the backward pass,
   /root/reference trainers/ar_vc.py:99-107: loss.backward() fills every parameter's .grad), as ONE grid of (problem, K chunk,
   256 x 128 tile) units, up to 40 problems per launch.  A reduction longer than 64 K tiles of 64 rows is cut into chunks -- a
   function of K only -- whose fp32 partial tiles go through `ws` and are added in chunk order by a second launch (deterministic).
     _ok        : 1 if the kernel takes `desc` (a function of the descriptor only).  Since round 4 that includes the exact-256
                  problems with >= 64 tiles of 128 x 128 (AAS-VC's decoder layers);
                  Round 5: B may also be the implicit im2col operand of a convolution weight gradient (S2SVC_LAYOUT_RC with
                  S2SVC_OP_CONV2D_S2 or S2SVC_OP_CONV1D, C % 128 == 0, whole images / utterances in K; Conv1d only for outputs of
                  >= 64 tiles) -- replaces autograd's conv weight gradients at subsampling.py:58-63 and alignments.py:28-60;
                  at most one convolution geometry per _grouped call;
     _ws_floats : fp32 elements of workspace the listed problems need (0 = none);
     _grouped   : launch; `ws` device memory (16-byte aligned) the caller keeps untouched until the launches have run. */
int s2svc_gemm_wgrad_ok(const s2svc_gemm_desc* desc /* host */);
int64_t s2svc_gemm_wgrad_ws_floats(const s2svc_gemm_desc* descs /* host */, int n);
int s2svc_gemm_wgrad_grouped(const s2svc_gemm_desc* descs /* host */, int n, float* ws, void* stream);
/* the same on a CAPPED grid: at most `wgs_cap` workgroups walk the units in order and leave the rest of the chip to the kernels of
   the stream the launch runs beside (forked gradient batches); same sums, same bits (wgs_cap <= 0: one workgroup per unit) */
int s2svc_gemm_wgrad_grouped_bg(const s2svc_gemm_desc* descs /* host */, int n, float* ws, void* stream, int wgs_cap);
/* A/B switch (tests, benchmarks): on = 0 / 1 (< 0: unchanged), kt_chunk = K tiles of 64 rows per chunk (<= 0: unchanged; default 64);
   returns the previous on | kt_chunk << 8. */
int s2svc_gemm_set_w8(int on, int kt_chunk);

/* Kernel-family switch for tests / A-B timing of s2svc_gemm's bf16 path: the 256-row, 8-wave, phase-interleaved kernel
   (csrc/gemm_8ph.hip: K-contiguous dense or Conv2d-3x3-s2 A operand, dense B, K % 64 == 0, >= 128 tiles) is tried first.
   mode & 15: 0 = never, 1 = default policy, 2 = policy without the half-phase skew of the two wave halves;
   (mode >> 4) & 15: 0 = tile geometry by policy, 1 = 256 x 256, 2 = 512 x 128, 3 = 256 x 128 forced (eligible problems only);
   (mode >> 8) & 15: 0 = unchanged, 1 + v sets the 256 x 96 p one-round geometry (gemm_8ph_kernel_n96): v = 0 never,
   1 by policy (default), 2 wherever N % 96 p == 0 (widest p), 4 / 5 = the same with p = 2 / 3 only.
   Returns the previous mode (mode < 0: query only).  Results do not depend on the mode beyond fp32 summation order. */
int s2svc_gemm_set_8ph(int mode);

/* ========================================================================================== */
/* LayerNorm fused with residual-add + dropout; BatchNorm1d; deterministic column reductions  */
/* replaces: modules/transformer/layer_norm.py:12-42 and the `residual + dropout(...)` lines   */
/* of encoder_layer.py:96-113, decoder_layer.py:104-127, conformer/encoder_layer.py:118-170;  */
/* torch.nn.BatchNorm1d at pre_postnets.py:124,152 and conformer/convolution.py:52,74.        */
/* ========================================================================================== */
/* s = res ? res + hscale*dropout(x) : x (written to s_out) ; y = LN(s)*gamma+beta ; mean/rstd [rows] fp32 */
int s2svc_layernorm_fwd(int dtype, int rows, int D, const void* x, const void* res, float drop_p, float hscale,
                        const uint64_t* seed_base, uint64_t seed_off, const float* gamma, const float* beta, float eps,
                        void* y, void* s_out, float* mean, float* rstd, void* stream);
/* ds = dLN(dy) + ds_extra ; dh = ds*mask*hscale (NULL to skip) */
int s2svc_layernorm_bwd(int dtype, int rows, int D, const void* dy, const void* s, const float* mean, const float* rstd,
                        const float* gamma, const void* ds_extra, float drop_p, float hscale, const uint64_t* seed_base,
                        uint64_t seed_off, void* ds, void* dh, void* stream);
/* chunked column reductions over (rows, D); modes 0..6 see csrc/norm.hip (5: sum dy, sum dy*v[r] with v in `mean`; 6: sum x, sum x^2); ws >= ws_chunks*2*D floats */
int s2svc_colreduce(int dtype, int rows, int D, int mode, const void* dy, const void* x, const float* mean,
                    const float* rstd, float scale, float* out_sum, float* out_dot, int accumulate, float* ws,
                    int ws_chunks, int Tn, const int32_t* vlens /* absent rows; scale < 0: 1 / number of present rows */, void* stream);
/* Several column reductions in two launches (stage 1, stage 2): the parameter gradients of the LayerNorm / BatchNorm /
   bias vectors of a few consecutive layers, queued by the host during backward.  Same semantics per item as
   s2svc_colreduce (ws >= ws_chunks*2*D floats each).  Two items of one call must not write the same out_sum / out_dot. */
typedef struct {
  const void* dy;
  const void* x;
  const float* mean;
  const float* rstd;
  float* out_sum;
  float* out_dot;
  float* ws;
  int32_t dtype, rows, D, mode, accumulate, ws_chunks;
  float scale;
  int32_t reserved_;
} s2svc_colreduce_item;
int s2svc_colreduce_grouped(const s2svc_colreduce_item* items /* host */, int n, void* stream);
/* LayerNorm backward (layer_norm.py:12-42 under autograd) together with the FIRST reduction stage of its parameter gradients
   (round 4): the launch also writes `chunks` partial row pairs ws[chunk][2][D] (sum of dy | sum of dy * xhat over the chunk's
   rows), which enter s2svc_colreduce_grouped as an item of mode 7 (partials ready: ws, ws_chunks = chunks, D, out_sum = d beta,
   out_dot = d gamma).  _pg_chunks: 0 = not eligible (use s2svc_layernorm_bwd and a mode-1 reduction), else the chunk count. */
int s2svc_layernorm_bwd_pg_chunks(int dtype, int rows, int D, const void* dy, const void* s, const float* gamma, const void* ds_extra,
                                  const void* ds, const void* dh);
int s2svc_layernorm_bwd_pg(int dtype, int rows, int D, const void* dy, const void* s, const float* mean, const float* rstd,
                           const float* gamma, const void* ds_extra, float drop_p, float hscale, const uint64_t* seed_base,
                           uint64_t seed_off, void* ds, void* dh, float* ws, void* stream);
int s2svc_bn_finalize(int C, int n, float eps, float momentum, const float* mean, const float* var, float* rstd,
                      float* run_mean, float* run_var, int64_t* num_batches, int var_is_ex2 /* var = E[x^2], colreduce mode 6 */,
                      int Tn, const int32_t* vlens /* absent rows: n = B * Tn -> the number of present rows */, void* stream);
int s2svc_rstd_from_var(int C, float eps, const float* var, float* rstd, void* stream);
/* The training-mode statistics of torch.nn.BatchNorm1d (pre_postnets.py:124,152, conformer/convolution.py:52,74) with the second
   reduction stage folded into the finalisation: 2 launches instead of 3 (ws >= ws_chunks*2*C floats).
   s2svc_bn_stats == s2svc_colreduce(mode 6, scale 1/rows) + s2svc_bn_finalize(var_is_ex2 = 1). */
int s2svc_bn_stats(int dtype, int rows, int C, const void* x, float eps, float momentum, float* mean, float* rstd,
                   float* run_mean, float* run_var, int64_t* num_batches, float* ws, int ws_chunks, int Tn, const int32_t* vlens,
                   void* stream);
int s2svc_bn_apply(int dtype, int64_t rows, int C, const void* x, const float* mean, const float* rstd,
                   const float* gamma, const float* beta, int act, float drop_p, const uint64_t* seed_base,
                   uint64_t seed_off, void* y, void* pre_act, int Tn, const int32_t* vlens, void* stream);
int s2svc_bn_bwd(int dtype, int64_t rows, int C, const void* dy, const void* x, const float* mean, const float* rstd,
                 const float* gamma, const float* sum_dy, const float* sum_dy_xhat, int use_batch_stats, void* dx,
                 int Tn, const int32_t* vlens, void* stream);

/* ========================================================================================== */
/* Attention probabilities: scale + relative shift + length/causal mask + softmax + dropout   */
/* replaces: modules/transformer/attention.py:63-93 (forward_attention), :237-260 / :142-160  */
/* (rel_shift new / legacy), :278-303 ((ac+bd)/sqrt(d_k)).  rel_mode 0 none, 1 new, 2 legacy. */
/* ========================================================================================== */
/* score / probability rows are `ld` >= T2 elements apart (rows padded to a vector multiple); pad columns are written as 0; */
/* rows of the relative-position term bd / dbd (length Lp) are `ldb` >= Lp elements apart (dbd pad columns are written as 0) */
int s2svc_attn_softmax_fwd(int dtype, int B, int H, int T1, int T2, int ld, const float* scores, const float* bd, int Lp,
                           int ldb, int rel_mode, float scale, const int32_t* klen, int causal, float drop_p,
                           const uint64_t* seed_base, uint64_t seed_off, void* attn, void* pdrop, void* stream);
int s2svc_attn_softmax_bwd(int dtype, int B, int H, int T1, int T2, int ld, const void* attn, const float* dp, const void* dattn,
                           float scale, float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* dscores,
                           void* dbd, int Lp, int ldb, int rel_mode, void* stream);

/* ========================================================================================== */
/* Elementwise: activations+dropout, positional encodings, head biases, GLU, casts, gathers   */
/* replaces: F.dropout / nn.Dropout sites (pre_postnets.py:63-66 always-on prenet dropout),   */
/* layers/positional_encoding.py:57-70,94-106,226-235,293-309, attention.py:283-286,          */
/* conformer/convolution.py:68 (GLU), conformer/swish.py.                                     */
/* ========================================================================================== */
int s2svc_act_dropout_fwd(int dtype, int64_t n, const void* x, int act, float p, const uint64_t* seed_base,
                          uint64_t seed_off, void* y, void* stream);
int s2svc_act_dropout_bwd(int dtype, int64_t n, const void* dz, const void* saved, int act, float p,
                          const uint64_t* seed_base, uint64_t seed_off, void* dx, void* stream);
int s2svc_posenc_fwd(int dtype, int64_t B, int T, int D, const void* x, float xscale, const float* alpha, const float* pe,
                     float p, const uint64_t* seed_base, uint64_t seed_off, void* y, void* stream);
int s2svc_posenc_bwd(int dtype, int64_t B, int T, int D, const void* dy, float xscale, const float* pe, float p,
                     const uint64_t* seed_base, uint64_t seed_off, void* dx, float* dalpha, float* partials, void* stream);
int s2svc_axpby(int dtype, int64_t n, float a, const void* x, float b, const void* y, void* out, void* stream);
/* ---- scalar / index glue of a training step (csrc/glue.hip): what sits between the fused kernels, one small launch each ---- */
#define S2SVC_SCALAR_TERMS_MAX 8
typedef struct {
  const float* x[S2SVC_SCALAR_TERMS_MAX]; /* term i: n[i] fp32 values (a scalar: n = 1) */
  int32_t n[S2SVC_SCALAR_TERMS_MAX];
  float w[S2SVC_SCALAR_TERMS_MAX];
  int32_t k;                              /* number of terms */
  int32_t reserved_;
} s2svc_scalar_terms;
/* *out = sum_i w[i] * sum_j x[i][j]: the training loss from its parts (trainers/aas_vc.py:100-139), means over the batch
   (forward_sum_loss.py:70-76, alignments.py:303-309); one wavefront, fixed order. */
int s2svc_weighted_sum(const s2svc_scalar_terms* terms /* host */, float* out, void* stream);
/* its backward: x[i][j] (written!) = w[i] * *g */
int s2svc_weighted_sum_bwd(const s2svc_scalar_terms* grads /* host; x[] are the OUTPUT buffers */, const float* g, void* stream);
/* acc[i] = beta * acc[i] + w[i] * sum_j x[i][j]: the running sums of the logged losses (trainers/base.py:198-213) */
int s2svc_scalars_axpy(const s2svc_scalar_terms* terms /* host */, float beta, float* acc, void* stream);
int s2svc_fill_zero(void* p, int64_t nbytes, void* stream);
/* *seed += inc (the device-resident dropout seed base, one bump per step) */
int s2svc_seed_advance(uint64_t* seed, uint64_t inc, void* stream);
/* out (rows, ldo) = [in (rows, N) | zeros] */
int s2svc_pad_cols(int dtype, int64_t rows, int N, int ldo, const void* in, void* out, void* stream);
/* teacher-forcing input of the AR decoder (models/vtn.py:236-243): out[b, t, :] = t == 0 ? 0 : ys[b, t * r - 1, :], ys fp32 with
   batch stride ys_batch_stride elements, out (B, Tin, D) in out_dtype */
int s2svc_decoder_input(int out_dtype, int B, int Tin, int r, int D, int64_t ys_batch_stride, const float* ys, void* out, void* stream);
/* token batch with <eos> behind every sequence (models/transformer_tts.py:139-142): out (B, T + 1) = [xs[b, :lens[b]] | eos | pad ...] */
int s2svc_append_eos(int B, int T, int64_t ldx, const int64_t* xs, const int32_t* lens, int64_t eos, int64_t pad, int64_t* out, void* stream);
/* out (rows, D) dense = src (rows, D) with row stride lds */
int s2svc_copy_rows(int dtype, int64_t rows, int D, int64_t lds, const void* src, void* out, void* stream);
/* stop-token targets (models/vtn.py:253-260): out (B, T) = labels (row stride ld_labels) with a 1 at frame lens[b] - 1 */
int s2svc_stop_labels(int B, int T, int64_t ld_labels, const float* labels, const int32_t* lens, float* out, void* stream);
/* out = x0 + x1 (+ x2 (+ x3)), k = 2..4 inputs summed in that order (fp32 arithmetic): the gradients arriving at a tensor with several
   consumers, in one launch (what autograd's accumulation does with k - 1 element-wise adds). */
int s2svc_add_n(int dtype, int64_t n, int k, const void* x0, const void* x1, const void* x2, const void* x3, void* out, void* stream);
int s2svc_add_head_bias(int dtype, int64_t rows, int D, const void* q, const float* u, const float* v, void* qu, void* qv,
                        void* stream);
/* the same with q a column block of a packed Q|K|V projection (row stride ldq elements); qu, qv dense (rows, D) */
int s2svc_add_head_bias_ld(int dtype, int64_t rows, int D, const void* q, int64_t ldq, const float* u, const float* v, void* qu,
                           void* qv, void* stream);
/* out[r, :D] = a[r, :D] + b[r, :D] over row-strided views (e.g. dQ = dQu + dQv written into the packed gradient) */
int s2svc_add_rows(int dtype, int64_t rows, int D, const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo,
                   void* stream);
int s2svc_glu_fwd(int dtype, int64_t rows, int C, const void* x, void* y, void* stream);
int s2svc_glu_bwd(int dtype, int64_t rows, int C, const void* x, const void* dy, void* dx, void* stream);
int s2svc_cast(int in_dtype, int out_dtype, int64_t n, const void* x, void* y, void* stream);
int s2svc_gather3(int in_dtype, int out_dtype, int n0, int n1, int n2, int64_t s0, int64_t s1, int64_t s2, int64_t off,
                  const void* in, void* out, void* stream);
/* Several gathers of fp32 sources in one launch: the permuted compute-dtype copies of the convolution weights
   ((O, I, k) -> (O, k, I) for the implicit-GEMM forward, (I, k reversed, O) for the data gradient, ...), refreshed once
   after the optimiser step instead of one launch per layer and pass inside the step.  `jobs` is a HOST array. */
typedef struct s2svc_gather3_job {
  const void* in;      /* fp32 */
  void* out;           /* contiguous n0 x n1 x n2, out_dtype */
  int64_t s0, s1, s2, off;
  int32_t n0, n1, n2;
  int32_t out_dtype;
} s2svc_gather3_job;
int s2svc_gather3_grouped(const s2svc_gather3_job* jobs /* host */, int n, void* stream);
/* dst[o][b][a] (+)= src[o][a][b], fp32, o < n: a convolution weight gradient (C_out, taps, C_in) as the GEMM leaves it into the
   parameter's (C_out, C_in, taps) layout (replaces the transposes autograd does inside conv backward: subsampling.py:58-63,
   pre_postnets.py:108-165, alignments.py:28-60 call sites).  A * (B + 1) * 4 bytes must fit 64 KB. */
int s2svc_permute_inner(int n, int A, int B, const float* src, float* dst, int accumulate, void* stream);
int s2svc_rowscale(int dtype, int64_t rows, int D, const void* x, const float* s, void* out, void* stream);

/* ========================================================================================== */
/* Convolution helpers                                                                         */
/* replaces: autograd of Conv2d stride 2 (subsampling.py:58-63); F.interpolate (aas_vc.py:    */
/* 340-349); depthwise Conv1d (conformer/convolution.py:42-51,70; vits/flow.py:137-146).      */
/* ========================================================================================== */
/* first front-end layer: Conv2d(1->O, 3x3, s2) + ReLU on the (B,T,F) mel batch, output NHWC; fused dW + dbias */
int s2svc_conv_in1_fwd(int dtype, int B, int Tn, int Fn, int O, const void* x, const float* w, const float* bias, void* y,
                       void* stream);
/* y != NULL: dy is the gradient of the ReLU output and is masked by (y > 0) on the fly */
int s2svc_conv_in1_wgrad(int dtype, int B, int Tn, int Fn, int O, const void* x, const void* dy, const void* y, float* dw,
                         float* db, int accumulate, float* partial, int max_chunks, void* stream);
int s2svc_col2im_s2(int dtype, int B, int T1, int F1, int C, int T2, int F2, const void* dcols, void* dx, void* stream);
/* ext_in / ext_out (device, one int32 each, or NULL): the lengths the reference's cropped tensors have when Tin / Tout are padded
   lengths of a captured step -- they set the resampling ratio; output frames >= *ext_out are zero, input frames >= *ext_in unread. */
int s2svc_interp_nearest(int dtype, int B, int Tin, int Tout, int C, const void* x, void* y, const int32_t* ext_in,
                         const int32_t* ext_out, void* stream);
int s2svc_interp_nearest_bwd(int dtype, int B, int Tin, int Tout, int C, const void* dy, void* dx, const int32_t* ext_in,
                             const int32_t* ext_out, void* stream);
int s2svc_dwconv(int dtype, int B, int Tn, int C, int ks, int dil, const void* x, const float* w, const float* bias,
                 void* y, int flip, void* stream);
/* y = dwconv(x) + add (add: a tensor of y's shape or NULL): the data gradient of a depthwise convolution whose input also feeds a
   residual connection (flow.py:148-190) takes the residual's gradient along instead of a separate add. */
int s2svc_dwconv_add(int dtype, int B, int Tn, int C, int ks, int dil, const void* x, const float* w, const float* bias,
                     const void* add, void* y, int flip, void* stream);
int s2svc_dwconv_wgrad(int dtype, int B, int Tn, int C, int ks, int dil, const void* x, const void* dy, float* dw,
                       int accumulate, float* ws, int ws_chunks, void* stream);

/* Core of the Conformer convolution module in training mode, bf16 (csrc/convmod.hip):
   GLU -> depthwise conv -> BatchNorm1d batch statistics -> Swish between the two pointwise convolutions.
   replaces: modules/conformer/convolution.py:68-75 (glu, depthwise_conv, norm, activation) and their autograd backward.
     y2 (B,Tn,2C) bf16 = output of pointwise_conv1;  w (C,1,ks) fp32, bias (C) or NULL;  z (B,Tn,C) bf16 = depthwise_conv(glu(y2));
     mean / rstd (C): batch statistics of z over all B*Tn frames (padded frames included, like the reference); run_mean / run_var /
     num_batches: torch.nn.BatchNorm1d's running buffers (may be NULL);  ws >= B * ceil(Tn / 64) * 2 * C floats.
   s2svc_convmod_supported: C % 64 == 0 and ks in {7, 15, 31}; everything else takes the separate kernels. */
int s2svc_convmod_supported(int C, int ks);
int s2svc_convmod_fwd(int B, int Tn, int C, int ks, const void* y2, const float* w, const float* bias, void* z, float eps,
                      float momentum, float* mean, float* rstd, float* run_mean, float* run_var, int64_t* num_batches, float* ws,
                      const int32_t* vlens, void* stream);
/* out = swish((z - mean) * rstd * gamma + beta), bf16, C % 64 == 0 */
int s2svc_bn_swish_apply(int64_t rows, int C, const void* z, const float* mean, const float* rstd, const float* gamma,
                         const float* beta, void* out, int Tn, const int32_t* vlens, void* stream);
/* da (B,Tn,C) bf16 = gradient of the Swish output -> dy2 (B,Tn,2C) bf16 = gradient of y2;  sdy / sdyx (C) = gradients of the
   BatchNorm bias / weight (also ADDED to dbeta_acc / dgamma_acc when given);  ws_w receives the per-tile partial sums of the
   depthwise weight and bias gradients, [B * ceil(Tn / 64)][C][ks + 1] (s2svc_convmod_wgrad_final sums them);
   ws_stats >= ceil(B * Tn / 64) * 2 * C floats. */
int s2svc_convmod_bwd(int B, int Tn, int C, int ks, const void* da, const void* z, const void* y2, const float* w, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, void* dy2, float* sdy, float* sdyx,
                      float* dgamma_acc, float* dbeta_acc, float* ws_stats, float* ws_w, const int32_t* vlens, void* stream);
int s2svc_convmod_wgrad_final(int C, int ks, int chunks, const float* ws_w, float* dw, float* db, int accumulate, void* stream);

/* BatchNorm1d (training mode) + activation + dropout on channel-last bf16 rows, C % 8 == 0, 16-byte accesses (csrc/convmod.hip).
   replaces: modules/pre_postnets.py:108-165 (BatchNorm1d -> Tanh -> Dropout of the Postnet layers) and its autograd backward.
     s2svc_bn_stats_vec: mean / rstd over (rows, C) + torch's running statistics; ws >= ceil(rows / 64) * 2 * C floats.
     s2svc_bn_act_apply_vec: y = dropout(act((x - mean) * rstd * gamma + beta)); pre_act (or NULL) = the value before act.
     s2svc_bn_act_bwd_vec: dz = d y; saved = y (relu / tanh / sigmoid) or pre_act (swish / gelu), NULL for act none and no dropout;
       -> dx, sdy = d beta, sdyx = d gamma (also ADDED to dbeta_acc / dgamma_acc when given).  The activation / dropout derivative is
       recomputed in both passes (no intermediate tensor).  Dropout masks: element index = row * C + channel, as s2svc_bn_apply. */
int s2svc_bn_stats_vec(int rows, int C, const void* x, float eps, float momentum, float* mean, float* rstd, float* run_mean,
                       float* run_var, int64_t* num_batches, float* ws, int Tn, const int32_t* vlens, void* stream);
int s2svc_bn_act_apply_vec(int rows, int C, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           int act, float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* y, void* pre_act, int Tn,
                           const int32_t* vlens, void* stream);
int s2svc_bn_act_bwd_vec(int rows, int C, const void* dz, const void* saved, const void* x, const float* mean, const float* rstd,
                         const float* gamma, int act, float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* dx, float* sdy,
                         float* sdyx, float* dgamma_acc, float* dbeta_acc, float* ws, int Tn, const int32_t* vlens, void* stream);

/* ========================================================================================== */
/* AAS alignment: pairwise -L2 + masked log-softmax, monotonic alignment search, Gaussian     */
/* upsampling weights.                                                                         */
/* replaces: modules/alignments.py:51-59 (AlignmentModule tail), :63-93 + :281-310 (numba MAS, */
/* per-utterance host loop, bincount, binarisation loss), length_regulator.py:111-154.        */
/* ========================================================================================== */
int s2svc_pairwise_l2_logsoftmax(int dtype, int B, int Tf, int Tx, int A, const void* feats, const void* text,
                                 const int32_t* text_lens, float* logp, float* dist, void* stream);
int s2svc_pairwise_l2_bwd_g(int dtype, int B, int Tf, int Tx, const float* logp, const float* dist, const float* dlogp,
                            const int32_t* text_lens, void* G, float* rowsum, void* stream);
int64_t s2svc_mas_ws_bytes(int B, int Tf, int Tx);
/* path (B,Tf) int32 (-1 beyond feat_len), ds (B,Tx) fp32 durations, binmean (B) = mean_t log_p[t, path[t]] */
int s2svc_mas(int B, int Tf, int Tx, const float* log_p_attn, const int32_t* text_lens, const int32_t* feat_lens,
              int32_t* path, float* ds, float* binmean, void* ws, void* stream);
int s2svc_mas_binloss_bwd(int B, int Tf, int Tx, const int32_t* path, const int32_t* feat_lens, const float* gout,
                          float* dlogp, void* stream);
int s2svc_gauss_upsample_probs(int dtype, int B, int Tf, int Tx, const float* ds, const int32_t* text_lens,
                               const int32_t* feat_lens, float delta, void* P, void* stream);

/* Length regulator of FastSpeech-style models -- replaces modules/length_regulator.py:46-97 (per-utterance
   torch.repeat_interleave + pad_list): frame i of utterance b is repeated ds[b, i] times.
     _index: start (B,Tx) = exclusive prefix sums of ds, idx (B,Tout) = source frame of every output frame (-1 = padding),
             total (B) = sum of ds (may be NULL);   _fwd: y (B,Tout,D) = x[b, idx] or pad_value;
     _bwd:   dx (B,Tx,D) = sum of dy over each frame's run (fixed order, no atomics). */
int s2svc_length_regulate_index(int B, int Tx, int Tout, const int32_t* ds, int32_t* start, int32_t* idx, int32_t* total,
                                void* stream);
int s2svc_length_regulate_fwd(int dtype, int B, int Tx, int Tout, int D, const void* x, const int32_t* idx, float pad_value,
                              void* y, void* stream);
int s2svc_length_regulate_bwd(int dtype, int B, int Tx, int Tout, int D, const void* dy, const int32_t* start, const int32_t* ds,
                              void* dx, void* stream);
/* Durations from attention maps -- replaces utils/duration_calculator.py:13-65: att (NH, Tf, Tx) fp32 (NH = layers * heads,
   or 1); picks the head with the largest mean row maximum, durations[j] (int64) = number of frames whose arg-max is j;
   focus_rate / head (device scalars, may be NULL) = that head's score / index. */
int s2svc_attn_durations(int NH, int Tf, int Tx, const float* att, int64_t* durations, float* focus_rate, int32_t* head,
                         void* stream);

/* ========================================================================================== */
/* Losses                                                                                      */
/* replaces: losses/seq2seq_loss.py:30-59, losses/l1_loss.py:22-49, guided_attention_loss.py: */
/* 142-165, forward_sum_loss.py:26-116 (F.ctc_loss loop + scipy beta-binomial prior).          */
/* ========================================================================================== */
int s2svc_seq_loss_fwd(int dtype, int B, int Tm, int D, const void* after, const void* before, const void* logits,
                       const float* ys, const float* labels, const int32_t* olens, float pos_weight, float* partial,
                       float* out, void* stream);
int s2svc_seq_loss_bwd(int dtype, int B, int Tm, int D, const void* after, const void* before, const void* logits,
                       const float* ys, const float* labels, const int32_t* olens, float pos_weight, const float* stats,
                       const float* g_l1, const float* g_bce, void* d_after, void* d_before, void* d_logits, void* stream);
int s2svc_guided_attn_loss_fwd(int dtype, int B, int H, int To, int Ti, const void* att, const int32_t* ilens,
                               const int32_t* olens, float sigma, float alpha, float* partial, float* out, void* stream);
int s2svc_guided_attn_loss_bwd(int dtype, int B, int H, int To, int Ti, const int32_t* ilens, const int32_t* olens,
                               float sigma, float alpha, const float* stats, const float* gout, void* datt, void* stream);
/* masked log-domain MSE of the deterministic duration predictor (losses/duration_predictor_loss.py:38-57);   */
/* stats[0] = number of valid tokens (kept for the backward)                                                 */
int s2svc_duration_loss_fwd(int B, int T, const float* d_outs, const float* ds, const int32_t* lens, float offset, int mean,
                            float* stats, float* out, void* stream);
int s2svc_duration_loss_bwd(int B, int T, const float* d_outs, const float* ds, const int32_t* lens, float offset, int mean,
                            const float* stats, const float* g, float* dd, void* stream);
int64_t s2svc_forward_sum_ws_bytes(int B, int Tf, int Tx);
int s2svc_forward_sum(int B, int Tf, int Tx, const float* log_p_attn, const float* prior, const int32_t* text_lens,
                      const int32_t* feat_lens, float log_blank, void* ws, float* loss_b, float* grad, void* stream);
int s2svc_betabinom_prior(int B, int Tf, int Tx, const int32_t* text_lens, const int32_t* feat_lens, float* prior,
                          void* stream);

/* ========================================================================================== */
/* Fused attention for short sequences (bf16, T1, T2 <= 64, d_k in {32,64,96,128}): one launch */
/* forward (scores, mask, softmax, dropout, P.V; writes the attention map) and one backward.   */
/* replaces: modules/transformer/attention.py:63-111 for VTN's shapes (T = 63/64, d_k = 96).   */
/* q/k/v/out/grads: element [b*bs + t*ld + h*dk + d] (bf16; slices of packed projections are   */
/* fine); attn / dattn: (B, H, T1, ld) bf16, pad columns >= T2 written as 0.                   */
/* ========================================================================================== */
int s2svc_attn_fused_supported(int dtype, int T1, int T2, int dk);
int s2svc_attn_fused_fwd(int B, int H, int T1, int T2, int dk, const void* q, int64_t ldq, int64_t qbs, const void* k, int64_t ldk,
                         int64_t kbs, const void* v, int64_t ldv, int64_t vbs, const int32_t* klen, int causal, float scale,
                         float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* attn, int ld, void* out, int64_t ldo,
                         int64_t obs, void* stream);
int s2svc_attn_fused_bwd(int B, int H, int T1, int T2, int dk, const void* q, int64_t ldq, int64_t qbs, const void* k, int64_t ldk,
                         int64_t kbs, const void* v, int64_t ldv, int64_t vbs, const void* dout, int64_t ldo, int64_t obs,
                         const void* attn, const void* dattn, int ld, float scale, float drop_p, const uint64_t* seed_base,
                         uint64_t seed_off, void* dq, int64_t lddq, int64_t dqbs, void* dk_out, int64_t lddk, int64_t dkbs, void* dv,
                         int64_t lddv, int64_t dvbs, void* stream);

/* ========================================================================================== */
/* Relative-position self-attention without the (B, H, T, 2T-1) position term in memory        */
/* (csrc/relattn.hip): bf16, T <= 256, d_k % 32 == 0, the "new" rel_shift (pos has 2T-1 rows). */
/* replaces: modules/transformer/attention.py:237-260 (rel_shift), :283-303 (q + pos_bias_u/v, */
/* matrix_ac, matrix_bd, the shift, the scaling) and :63-93 (mask, softmax, mask) forward; the */
/* backward of the same up to the gradient of the scaled scores and of matrix_bd.              */
/*   fwd: attn / pdrop (B, H, T, ld) bf16 (pdrop = the dropped copy, NULL when drop_p == 0),   */
/*        qu = q + pos_bias_u, qv = q + pos_bias_v (B, T, H*dk) for the backward GEMMs.        */
/* ========================================================================================== */
int s2svc_relattn_supported(int dtype, int T, int dk, int rel_mode);
int s2svc_relattn_fwd(int B, int H, int T, int dk, const void* q, int64_t ldq, int64_t qbs, const void* k, int64_t ldk, int64_t kbs,
                      const void* pos, int64_t ldp, int L, const float* u, const float* v, const int32_t* klen, float scale,
                      float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* attn, void* pdrop, int ld, void* qu, void* qv,
                      void* stream);

/* ========================================================================================== */
/* Attention map of plain multi-head attention, medium sequences (csrc/attn_map.hip): bf16,   */
/* T2 <= 512 keys, d_k % 32 == 0.  replaces: modules/transformer/attention.py:63-93 (scores / */
/* sqrt(d_k), masked_fill(min), softmax, masked_fill(0), dropout) as one launch; the context   */
/* P.V stays a GEMM.  q (B, T1, .) / k (B, T2, .) views (row strides ldq / ldk, batch strides  */
/* qbs / kbs, head h at columns h * dk); klen (B) int32 or NULL; causal: keys j <= i only;     */
/* attn / pdrop (B, H, T1, ld) bf16, ld = T2 rounded up to 8, pad columns zero (pdrop = the    */
/* dropped copy, NULL when drop_p == 0); masks as s2svc_attn_softmax_fwd draws them.           */
/* ========================================================================================== */
int s2svc_attn_map_supported(int dtype, int T1, int T2, int dk);
/* 1 when the launches below can also carry the product of their finished tile with a (T2, d_k) matrix of the head (d_k in {64, 96,   */
/* 128}): forward the context ctx = (dropped map) . v (attention.py:95-111 up to the output projection), backward dq = dS . k.        */
int s2svc_attn_map_product_supported(int dk);
/* v != NULL: ctx (B, T1, .) view (row stride ldc, batch stride cbs) receives the context in the same launch. */
int s2svc_attn_map_fwd(int B, int H, int T1, int T2, int dk, const void* q, int64_t ldq, int64_t qbs, const void* k, int64_t ldk,
                       int64_t kbs, const int32_t* klen, int causal, float scale, float drop_p, const uint64_t* seed_base,
                       uint64_t seed_off, void* attn, void* pdrop, int ld, const void* v, int64_t ldv, int64_t vbs, void* ctx,
                       int64_t ldc, int64_t cbs, void* stream);
/* backward of the same up to the gradient of the scaled scores: ds = attn * (dP * mask + dattn - rowsum(attn * (dP * mask + dattn)))  */
/* * scale with dP = dctx . v^T computed on chip (replaces one batched GEMM + s2svc_attn_softmax_bwd); dctx (B, T1, .) / v (B, T2, .)  */
/* views, attn the stored map, dattn the gradient that reached the map itself (or NULL), ds (B, H, T1, ld) bf16, pad columns zero.     */
/* dbd != NULL: relative-position self-attention (attention.py:237-260 "new" rel_shift, T1 == T2): dbd (B, H, T1, ldb) bf16 receives   */
/* the gradient of matrix_bd BEFORE the shift (dbd[b,h,i,T1-1-i+j] = ds[b,h,i,j], zero elsewhere; ldb >= 2 T1 - 1, a multiple of 8).   */
/* k != NULL: dq (B, T1, .) view (row stride lddq, batch stride dqbs) receives ds . k in the same launch.                              */
int s2svc_attn_map_bwd(int B, int H, int T1, int T2, int dk, const void* dctx, int64_t ldo, int64_t obs, const void* v, int64_t ldv,
                       int64_t vbs, const void* attn, const void* dattn, float scale, float drop_p, const uint64_t* seed_base,
                       uint64_t seed_off, void* ds, int ld, void* dbd, int ldb, const void* k, int64_t ldk, int64_t kbs, void* dq,
                       int64_t lddq, int64_t dqbs, void* stream);

/* ========================================================================================== */
/* Token embedding of Transformer-TTS (models/transformer_tts.py:63-77, Embedding(idim, adim, */
/* padding_idx=0)): y[i,:] = W[idx[i],:] ; dW[v,:] = sum_{idx[i]==v} dy[i,:], dW[padding]=0     */
/* ========================================================================================== */
int s2svc_embedding_fwd(int dtype, int64_t n, int D, int V, const int64_t* idx, const float* w, void* y, void* stream);
int s2svc_embedding_bwd(int dtype, int64_t n, int D, int V, const int64_t* idx, const void* dy, int64_t padding_idx,
                        float* dw, int accumulate, void* stream);

/* ========================================================================================== */
/* Stochastic duration predictor (VITS flows), channel-last rows r=(b,t), mask = t < lens[b]   */
/* replaces: modules/duration_predictor.py:211-304, modules/vits/flow.py:18-310,               */
/* modules/vits/transform.py:17-216.  Spline / glue stages are fp32.                           */
/* ========================================================================================== */
int s2svc_mask_rows(int dtype, int B, int T, int C, const void* x, const int32_t* lens, void* y, void* stream);
/* Conv1d(1->C,k=1) + conditioning, masked: y[r,c] = mask*(a[r]*w[c] + bias[c] + g[r,c])   (flow.py:290-292) */
int s2svc_expand_fwd(int dtype, int B, int T, int C, const float* a, const float* w, const float* bias, const void* g,
                     const int32_t* lens, void* y, void* stream);
/* dg = mask*dy ; da[r] = sum_c dg[r,c]*w[c]   (dw, dbias: s2svc_colreduce mode 5 over dg with v = a) */
int s2svc_expand_bwd(int dtype, int B, int T, int C, const void* dy, const float* w, const int32_t* lens, void* dg, float* da,
                     void* stream);
/* y = mask*(res + dropout(act(LayerNorm(x))))   (one DDS half-layer, flow.py:148-190); res / lens optional, D <= 1024 */
int s2svc_ln_act_fwd(int dtype, int rows, int D, int T, const void* x, const float* gamma, const float* beta, float eps, int act,
                     const void* res, const int32_t* lens, float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* y,
                     float* mean, float* rstd, void* stream);
/* the first half of a DDS layer in one launch (fp32): u = depthwise Conv1d(x; dw_w (D, ks), dw_b, dilation dil, zero padding inside
   the T frames of an utterance) -- flow.py:137-146 -- written out for the backward pass, y = act(LayerNorm(u)); D <= 512, ks odd */
int s2svc_dw_ln_act_fwd(int B, int T, int D, int ks, int dil, const float* x, const float* dw_w, const float* dw_b,
                        const float* gamma, const float* beta, float eps, int act, float* u, float* y, float* mean, float* rstd,
                        void* stream);
/* du = gradient at the LayerNorm output (feeds dgamma/dbeta via colreduce mode 1), dx = LN input grad, dres = mask*dy */
int s2svc_ln_act_bwd(int dtype, int rows, int D, int T, const void* dy, const void* x, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, int act, const int32_t* lens, float drop_p,
                     const uint64_t* seed_base, uint64_t seed_off, void* du, void* dx, void* dres, void* stream);

/* rational-quadratic spline coupling with linear tails; h (rows, 3*bins-1); lad accumulates when asked (transform.py:96-216) */
int s2svc_rq_spline_fwd(int B, int T, int bins, const float* x, const float* h, float hscale, float bound, const int32_t* lens,
                        int inverse, float* out, float* lad, int lad_accumulate, void* stream);
/* g_lad: (B) gradient shared by all rows of an utterance (the log-dets are summed over t) */
int s2svc_rq_spline_bwd(int B, int T, int bins, const float* x, const float* h, float hscale, float bound, const int32_t* lens,
                        const float* g_out, const float* g_lad, float* dx, float* dh, void* stream);
/* glue: noise -> first affine flow | dequantise + log flow + affine | per-utterance NLL (duration_predictor.py:239-280) */
int s2svc_sdp_head_fwd(int B, int T, const float* noise, const int32_t* lens, const float* m, const float* logs, float* z0,
                       float* z1, void* stream);
int s2svc_sdp_head_bwd(int B, int T, const float* noise, const int32_t* lens, const float* logs, const float* dz0,
                       const float* dz1, float* part, void* stream);
int s2svc_sdp_mid_fwd(int B, int T, const float* zu, const float* z1, const float* w, const int32_t* lens, const float* m,
                      const float* logs, float* y0, float* y1, float* lz, void* stream);
int s2svc_sdp_mid_bwd(int B, int T, const float* zu, const float* z1, const float* w, const int32_t* lens, const float* logs,
                      const float* dy0, const float* dy1, const float* dlz, float* dzu, float* dz1, float* part, void* stream);
int s2svc_sdp_tail_fwd(int B, int T, const float* noise, const int32_t* lens, const float* zu, const float* lz,
                       const float* lad_q, const float* lad_p, const float* af, const float* bf, const float* logs_q,
                       const float* logs_p, float* out, int normalize /* 1: / sum_b min(lens[b], T), aas_vc.py:403 */, void* stream);
int s2svc_sdp_tail_bwd(int B, int T, const float* g, const int32_t* lens, const float* zu, const float* af, const float* bf,
                       float* d_af, float* d_bf, float* d_lz, float* d_zu, float* neg_g, float* part, int normalize, void* stream);
/* inference read-out: dur = ceil(exp((a - m0)*exp(-logs0)))*mask   (duration_predictor.py:300-304) */
int s2svc_sdp_inverse_out(int B, int T, const float* a, const int32_t* lens, const float* m, const float* logs, float* dur,
                          void* stream);

/* ========================================================================================== */
/* Autoregressive decode step (static K/V cache, device-resident step index, device stop test) */
/* replaces: models/vtn.py:344-389 / models/transformer_tts.py:268-321 (generation loop),      */
/* modules/transformer/decoder.py:239-273 (forward_one_step), decoder_layer.py:85-132 (cache). */
/* `pos` is one device int32 (0-based position of the frame being generated); every kernel of */
/* a step reads it, s2svc_decode_advance bumps it: one captured hipGraph serves all steps.    */
/* ========================================================================================== */
/* y[b,:] = x[b,:]*xscale + alpha*pe[pos,:]   (embedding.py:73-88 / :115-125 for one position) */
int s2svc_decode_posenc(int dtype, int B, int D, const void* x, float xscale, const float* alpha, const float* pe,
                        const int32_t* pos, void* y, void* stream);
/* one query row per (utterance, head) against a K/V cache (B, Tk, .) with time stride ldt and batch stride cbs.   */
/* knew/vnew != NULL (self-attention): the row is first appended at position *pos, keys 0..*pos are attended;      */
/* else (source attention) keys 0..klen[b]-1.  att != NULL: probabilities (zeros past the valid keys) are stored   */
/* at att[b*att_bs + h*att_hs + *pos*att_ps + j], j < Tk  (vtn.py:364-375 collects src_attn.attn[0,:,-1]).          */
int s2svc_decode_attn(int dtype, int B, int H, int dk, const void* q, int64_t ldq, void* kcache, void* vcache,
                      int64_t ldt, int64_t cbs, const void* knew, const void* vnew, int64_t ldn, const int32_t* pos,
                      const int32_t* klen, int Tk, float scale, void* ctx, int64_t ldo, float* att, int64_t att_bs,
                      int64_t att_hs, int64_t att_ps, void* stream);
/* outs[b, pos*r+i, :] = feat[b, i, :]; probs = sigmoid(logit); prev[b,:] = last frame; stop test vtn.py:378-381 */
int s2svc_decode_emit(int dtype, int B, int r, int odim, const void* feat, const void* logit, float threshold,
                      const int32_t* minlen, const int32_t* maxlen, const int32_t* pos, float* outs, int64_t outs_bs,
                      float* probs, int64_t probs_bs, void* prev, int32_t* stop_at, void* stream);
int s2svc_decode_advance(int32_t* pos, uint64_t* seed_base, uint64_t seed_stride, void* stream);
/* LayerNorm -> Linear of a decode step, one launch: C[M <= 64, N] = act(LN(A)[M, K] . B[N, K]^T + bias) [dropout] (+ res);
   `desc` carries A (the pre-LayerNorm rows), B, C and the epilogue fields of s2svc_gemm (dense K-contiguous operands,
   unbatched, unsplit); gamma / beta / eps: the LayerNorm (both NULL: plain skinny linear); y_out != NULL: LN(A) is also
   written there (row stride ldy) -- the residual input of a post-LN layer (decoder_layer.py:104-127).
   Replaces one LayerNorm launch + one Linear launch per projection of decoder.py:239-273. */
/* 1 if s2svc_decode_ln_linear takes an (M x K) input of this dtype (else: s2svc_layernorm_fwd + s2svc_gemm) */
int s2svc_decode_ln_linear_supported(int dtype, int M, int K);
int s2svc_decode_ln_linear(const s2svc_gemm_desc* desc /* host */, const float* gamma, const float* beta, float eps, void* y_out,
                           int64_t ldy, void* stream);
/* Round 6 (csrc/decode_fused.hip): s2svc_decode_emit on ONE packed feat_out | prob_out projection (feat / logit rows with stride ldf)
   + s2svc_decode_advance by the last workgroup to finish (`ticket`: one zero-initialised device uint32, zero again afterwards)
   + pe_next != NULL: pe_next[0 .. D) = alpha * pe[(*pos + 1) * D + .] in `dtype` -- the positional-encoding row of the NEXT position,
   which the input Linear of the next step adds as a residual with row stride 0 (embedding.py:115-125; no s2svc_decode_posenc launch). */
int s2svc_decode_emit_advance(int dtype, int B, int r, int odim, const void* feat, const void* logit, int64_t ldf, float threshold,
                              const int32_t* minlen, const int32_t* maxlen, int32_t* pos, float* outs, int64_t outs_bs, float* probs,
                              int64_t probs_bs, void* prev, int32_t* stop_at, uint64_t* seed_base, uint64_t seed_stride,
                              uint32_t* ticket, const float* pe, const float* alpha, int D, int pe_rows, void* pe_next, void* stream);

/* ========================================================================================== */
/* Optimiser: grad-norm -> clip -> WarmupLR -> Adam (+ bf16 shadow) over one flat buffer       */
/* replaces: trainers/ar_vc.py:99-107 (clip_grad_norm_, Adam.step, scheduler.step),            */
/* schedulers/warmup_lr.py:54-61.  state: 4 device floats {step, lr, grad_norm, clip_coef}.    */
/* ========================================================================================== */
int s2svc_adam_step(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* bf16_shadow,
                    float beta1, float beta2, float eps, float max_norm, float base_lr, float warmup_steps,
                    double* partial, float* state, void* stream);

/* dst[c, r] = src[r, c] (bf16) for every 64x64 tile listed in `tiles` (device, 4 x int64 per tile: src offset of  */
/* the matrix, dst offset, rows<<32|cols, tile index): keeps W^T beside the bf16 weight shadow so that the data-  */
/* gradient GEMMs read K-contiguous operands.                                                                   */
int s2svc_transpose_tiles(int64_t ntiles, const int64_t* tiles, const void* src, void* dst, void* stream);

/* ========================================================================================== */
/* STFT -> log-mel glue (the two contractions use s2svc_gemm)                                  */
/* replaces: bin/preprocess.py:63-92 (librosa.stft reflect padding, abs, max(eps,.), log10)    */
/* and optionally bin/normalize.py:172-193 ((x-mean)/scale fused into the log kernel).         */
/* ========================================================================================== */
int s2svc_reflect_pad(int64_t n, int pad, const float* x, float* y, void* stream);
int s2svc_magnitude(int64_t frames, int nb, const float* z, float* spc, void* stream);
int s2svc_log_clamp(int64_t n, int D, const float* x, float eps, float inv_log_base, const float* mean,
                    const float* inv_scale, float* y, void* stream);

/* Batched form: B utterances per launch, wav batch -> normalised zero-padded (B, Tmax, n_mels) log-mel batch -- what
   bin/preprocess.py:200-310 (one utterance at a time), bin/normalize.py:172-193 and the collater's padding
   (collaters/ar_vc.py:42-45) produce in three passes over HDF5 files.
     s2svc_reflect_pad_batch: x (B, Nmax) zero-padded waveforms, nlen (B) sample counts -> y (B, ld), ld >= Nmax + 2*pad:
         row b = reflect-padded utterance b followed by zeros (frames past the utterance then transform zeros);
     [one batched s2svc_gemm: A = y with row stride `hop` and batch stride ld, B = windowed DFT basis -> z (B, Tmax, 2*nb)]
     s2svc_mel_log_batch: z -> out (B, Tmax, nmel): magnitude, mel projection over each filter's bins [lo, hi) only,
         max(eps, .), log, optional (x - mean) * inv_scale; frames t >= frames[b] are written as zeros.
     s2svc_ragged_to_padded: ragged feature rows (concatenated utterances, row offsets (B + 1) int64) -> (B, Tmax, D)
         zero-padded batch (+ optional normalisation, + optional stop labels (B, Tmax): 1 from the last valid frame on). */
int s2svc_reflect_pad_batch(int B, int64_t Nmax, int pad, int64_t ld, const float* x, const int32_t* nlen, float* y, void* stream);
/* The same batch in ONE launch with the FFT in LDS (csrc/stft_fft.hip; n_fft in {512, 1024, 2048}): one wavefront per frame --
   windowed load with the reflect padding as index arithmetic, N/2-point complex radix-4 Stockham FFT of the even/odd packed
   frame, unpack, magnitude, sparse mel projection, clamp, log, normalisation, zero rows for the padding frames.
     tables (16-byte aligned, fp32, zero-filled to whole 16-byte vectors): w_half [n_fft/2] complex exp(-2 pi i m / (n_fft/2)) |
     w_full [n_fft/2 + 1] complex exp(-2 pi i k / n_fft) | win [n_fft] | melw [melw_n rounded up to even];
     melw: the non-zero weights of filter m are melw[mel_off[m] .. mel_off[m] + mel_hi[m] - mel_lo[m]) for bins [mel_lo, mel_hi);
     mel_maxw = the widest filter (bins); the list (melw_n values) ends with a zero tail of at least mel_maxw + 1 values. */
int s2svc_stft_logmel_fft_supported(int n_fft, int nmel, int melw_n);
int s2svc_stft_logmel_fft(int B, int64_t Nmax, int Tmax, int n_fft, int hop, int nmel, const float* x, const int32_t* nlen,
                          const int32_t* frames, const float* tables, const int32_t* mel_lo, const int32_t* mel_hi, const int32_t* mel_off, int melw_n, int mel_maxw,
                          float eps, float inv_log_base, const float* mean, const float* inv_scale, float* out, void* stream);

/* n_fft = 1024 and triangular mel filters (every bin has at most two non-zero weights, in neighbouring filters; n_mels <= 128): the
   512-point FFT as three passes of in-register 8-point DFTs, per-lane tables in registers, mel projection by segments.
     tables: the packed table of s2svc_stft_logmel_fft (w_half | w_full | win are read);
     seg_lo / seg_len [nmel]: the contiguous bin range whose HIGHEST filter is m;  wud [513][2]: weight of that filter and of the
     one below it at bin k;  mel[m] = sum_{k in seg m} wud[k][0] |X[k]| + sum_{k in seg m+1} wud[k][1] |X[k]|. */
int s2svc_stft_logmel_fft8(int B, int64_t Nmax, int Tmax, int hop, int nmel, const float* x, const int32_t* nlen, const int32_t* frames,
                           const float* tables, const int32_t* seg_lo, const int32_t* seg_len, const float* wud, float eps,
                           float inv_log_base, const float* mean, const float* inv_scale, float* out, void* stream);

int s2svc_mel_log_batch(int B, int Tmax, int nb, int nmel, const float* z, const int32_t* frames, const float* melb,
                        const int32_t* lo, const int32_t* hi, float eps, float inv_log_base, const float* mean,
                        const float* inv_scale, float* out, void* stream);
int s2svc_ragged_to_padded(int B, int Tmax, int D, const float* ragged, const int64_t* offsets, const float* mean,
                           const float* inv_scale, float* out, float* labels, void* stream);

#ifdef __cplusplus
}
#endif
#endif
