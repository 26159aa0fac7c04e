/* mart_hip.h -- C ABI of libmart_hip.so, the gfx950 (MI355X) kernels behind the MarT / MKGformer hot path.
 *
 * The reference (zjunlp/MKG_Analogy, /root/reference) is pure Python on torch ops: it has no FFI.  The drop-in
 * boundary is its Python operator API (MarT/models/model.py:7 MKGformerKGC, MarT/models/modeling_unimo.py:848
 * UnimoForMaskedLM.forward) and trainer surface (MarT/lit_models/transformer.py:18 TransformerLitModel).  This
 * header is what the Python host side binds with ctypes; every entry point names the reference lines whose
 * eager torch ops it replaces.
 *
 * Conventions
 *   - plain pointers + sizes; the CALLER owns every buffer (incl. workspaces); kernels never allocate or free
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no internal synchronisation
 *   - return 0 on success, <0 on error (-1 bad argument, -2 launch failure); mart_last_error() has the text
 *   - stateless apart from a per-device "dynamic-LDS attribute set" flag per kernel; one host thread per GPU process is the
 *     supported model (two threads racing on a kernel's first launch both set the attribute: harmless); several devices per
 *     process are fine (the flag is kept per device)
 *   - bf16 tensors are raw uint16 storage; "ld*" are leading dimensions in ELEMENTS
 */
#ifndef MART_HIP_H
#define MART_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MART_ACT_NONE 0
#define MART_ACT_GELU 1   /* erf GELU   : transformers ACT2FN["gelu"],       modeling_unimo.py:454,967 */
#define MART_ACT_QGELU 2  /* quick GELU : transformers ACT2FN["quick_gelu"], modeling_unimo.py:279     */
#define MART_ACT_STORED 3 /* mul_act only: mulz already holds act'(z) (written by a forward call with preact_grad) */

const char* mart_last_error(void);
int mart_abi_version(void);
/* 0 if the current HIP device is gfx950, <0 otherwise (the product path refuses to run elsewhere) */
int mart_check_device(void);

/* ---------------------------------------------------------------- dense contractions
 * C = epi( alpha * (A[M,K] B[N,K]^T + A2[M,K2] B2[N,K2]^T) ), bf16 operands, fp32 accumulate.
 * epi: +bias(+bias2) -> store preact -> act -> *act'(mulz) -> +res_f32 -> +res_bf16 -> store C (bf16|f32) [+C2 bf16]
 * preact_grad: the preact buffer receives act'(z) instead of z -- the forward pass has sigmoid / erf of z in registers
 * anyway, and the backward product (mulz = that buffer, mul_act = MART_ACT_STORED) becomes a plain multiply: the two
 * transcendentals per element of the derivative were a third of the data-gradient GEMM's epilogue time.
 * Replaces nn.Linear / F.linear / conv-as-GEMM / bmm call sites: modeling_unimo.py:123-124 (patch embedding),
 * :223-225,270 (CLIP q/k/v/out_proj), :284-286 (CLIP MLP), :327-333 (BERT q/k/v), :388,459-463,475 (BERT dense,
 * intermediate.dense + fusion_dense as one K=2H contraction, output.dense), :405,411 (fusion bmm),
 * :958,973 (MLM head: only the gathered [MASK] rows x the gathered entity rows of the tied embedding). */
typedef struct {
  const void* A; const void* B; const void* A2; const void* B2;   /* bf16; A2/B2 share lda/ldb, rows and strides */
  int lda, ldb;
  int M, N, K, K2;
  const int32_t* a_rows; const int32_t* b_rows;                   /* optional row gathers (int32 indices) */
  int batch; long long stride_a, stride_b, stride_c, stride_aux;   /* elements; aux = res/mulz */
  const float* bias; const float* bias2; int bias_by_brow;        /* bias index = b_rows[n] when set */
  int act; void* preact;                                          /* preact: bf16 [M,N] ld = ldc */
  const void* mulz; int mul_act;                                  /* bf16 [M,N] ld = ldres */
  const float* res_f32; const void* res_bf16; int ldres;          /* 0 -> ldc */
  float alpha;
  void* C; int ldc; int c_f32;
  void* C2; int ldc2;                                             /* optional bf16 copy; 0 -> ldc */
  int tile_cfg;                                                   /* 0 auto (256x256 tiles from 128 tiles up and M > 128, else 128x128), 128, 256; 2561 = 256-tile with the
                                                                     general epilogue (test hook).  Anything else is rejected by the product library; the timing / A-B
                                                                     experiment codes of tools/nt_harness exist only in -DMART_EXPERIMENTS builds of gemm_nt.hip */
  int preact_grad;                                                /* preact receives act'(z) rather than z (needs preact and act) */
  int b_blocked;                                                  /* B (and B2) stored tile-blocked [N/256][K/64][256][64] (mart_block_table): every LDS-DMA stage of the weight
                                                                     operand is one contiguous 32 KB run instead of 256 strided 128-B rows; needs N % 256 == 0, no b_rows */
  int a_src_rows, b_src_rows;                                     /* with a_rows / b_rows: number of rows of the table the indices point into (the kernel forms 32-bit
                                                                     element offsets row * ld, so src_rows * ld must stay below 2^32; checked when stated, 0 = the caller
                                                                     vouches for it).  Without a gather the same check uses M * lda / N * ldb. */
  int in_f16, c_f16;                                              /* in_f16: A, B (A2, B2) hold IEEE fp16 instead of bf16 (v_mfma_f32_16x16x32_f16 / 32x32x16_f16, same rate, 11-bit
                                                                     significands): the forward linear layers of the text stream.  c_f16 (needs in_f16, 16-bit C): C is
                                                                     written as fp16 and C2, when given, is its bf16 copy (what the backward pass reads).  preact stays bf16. */
  int c_split3;                                                   /* C is bf16 [M, >= 3N]: the f32 result (after bias / activation) as the two-term split [hi | lo | hi]
                                                                     (mart_split_bf16x3 role 0) -- the A operand of the next GEMM of the fp32-accurate path, written by
                                                                     the epilogue instead of an f32 C plus a split pass.  256 x 256 tiles, N % 256 == 0, no other outputs;
                                                                     NOT available with b_blocked or operands beyond 2^31 elements (status < 0, as of ABI 9) */
  /* LayerNorm folded into the product that consumes it (nn.LayerNorm -> nn.Linear: modeling_unimo.py:509 -> :223-225 and :518 -> :284-286), for forward
   * passes that keep nothing for a backward pass:  LN(x) W^T + b = rstd (x (gamma o W)^T - mean s) + b'  (mart_ln_fold_prep makes gamma o W, s, b').
   * row_stats (producer: f32 C + res_f32 + C2, N % 64 == 0): float [M][N / 64][2], per row and 64-column slice the (sum, sum of squares) of the f32 values
   * written; mart_ln_stats_finalize turns them into mean / rstd.  ln_mean / ln_rstd [M], ln_colsum [N] (consumer: 16-bit C, plain or activation-only
   * epilogue): A is the producer's C2, B the folded weight, bias the folded bias; the epilogue applies (acc - mean[m] s[n]) rstd[m] + bias[n]. */
  float* row_stats;
  const float* ln_mean; const float* ln_rstd; const float* ln_colsum;
} mart_gemm_nt_desc;
int mart_gemm_nt(const mart_gemm_nt_desc* d, void* stream);

/* out[NX,NY] (f32, atomically accumulated) += alpha * X[M,NX]^T Y[M,NY]; optional colsum[NX] += sum_m X[m,:].
 * Weight / bias gradients of every nn.Linear above (autograd of the same reference lines), the tied
 * embedding gradient rows (out_rows scatter), and the fusion op's d(vis). */
typedef struct {
  const void* X; const void* Y; int ldx, ldy;                     /* bf16 */
  int M, NX, NY;
  float* out; int ldo; const int32_t* out_rows;                  /* out_rows: scatter of the NX result rows; the indices must be UNIQUE when a workspace
                                                                     is given (the ordered reduction adds each row with a plain read-modify-write); the
                                                                     atomic path tolerates repeats */
  float* colsum; int colsum_by_row;                               /* colsum index = out_rows[nx] when set */
  int batch; long long stride_x, stride_y, stride_o;
  int splits;                                                     /* 0 auto */
  float alpha;
  void* workspace; long long workspace_bytes;                     /* optional, caller-owned (mart_gemm_tn_workspace_bytes): when given (and batch <= 1) the
                                                                     split reduction is DETERMINISTIC -- partial tiles go to the workspace and an ordered second
                                                                     kernel adds them into out / colsum (an M % 64 tail is added by one workgroup per tile);
                                                                     without it: f32 atomics */
} mart_gemm_tn_desc;
int mart_gemm_tn(const mart_gemm_tn_desc* d, void* stream);
long long mart_gemm_tn_workspace_bytes(int M, int NX, int NY, int splits);

/* ---------------------------------------------------------------- layer norm family
 * s = x_f32 (+ dropout(y_bf16 | y_f32, p)) ; out = LN(s) * gamma + beta.  nn.LayerNorm call sites
 * modeling_unimo.py:509,518,711 (CLIP pre-LN, eps 1e-5), :390,477 (BERT post-LN with dropout+residual,
 * eps 1e-12), :975 (head transform LN). */
typedef struct {
  const float* x_f32; const void* y_bf16;                         /* either may be NULL, not both */
  float p_drop; uint64_t seed;                                    /* dropout on y_bf16 only */
  const float* gamma; const float* beta; float eps;
  int M, H;
  float* s_out;                                                   /* optional: the pre-LN sum (saved for bwd) */
  float* out_f32; void* out_bf16;                                 /* either may be NULL */
  float* mean; float* rstd;                                       /* [M] */
  const float* y_f32;                                             /* optional: the dropped-out branch in f32 instead of y_bf16 (the text layers' dense outputs) */
  void* out_f16;                                                  /* optional: the normalised output as IEEE fp16 (A operand of the text stream's fp16 forward products) */
  const int32_t* x_rows;                                          /* optional row gather of the residual input: row m reads x_f32[x_rows[m]] (last text layer, computed
                                                                     only on the rows the loss reads); y_*, s_out, outputs, mean / rstd and the dropout index stay compact */
  void* out_split3;                                               /* optional bf16 [M, 3H]: the output as the two-term split [hi | lo | hi] (mart_split_bf16x3 role 0),
                                                                     the A operand of the fp32-accurate path's GEMMs -- written here instead of by a separate pass */
} mart_ln_fwd_desc;
int mart_ln_fwd(const mart_ln_fwd_desc* d, void* stream);

/* ds = LNbwd(dy_f32 + dy_bf16) + add_f32 ; dgamma/dbeta atomically accumulated.
 * dx_bf16 = bf16(ds) * dropmask/(1-p) when p_drop>0 (gradient of the dropped-out branch y_bf16). */
typedef struct {
  const float* dy_f32; const void* dy_bf16;
  const float* s; const float* mean; const float* rstd; const float* gamma;
  const float* add_f32;
  int M, H;
  float* ds_f32; void* ds_bf16;
  float p_drop; uint64_t seed;
  float* dgamma; float* dbeta;
  int bf16_total;                                                  /* 1: ds_bf16 = bf16(ds + add_f32) (no dropout mask) */
  float* ws; long long ws_bytes;                                   /* optional caller-owned workspace (768 * 2 * H floats suffice): per-workgroup dgamma / dbeta
                                                                      partials, added in workgroup order by a second kernel (deterministic); NULL: f32 atomics */
  const float* add2_f32;                                           /* optional second residual operand (needs add_f32): ds = LNbwd(dy) + add_f32 + add2_f32 -- the
                                                                      d(visual) of the fusion op of the text layer below, delivered through a side buffer */
  int defer_reduce;                                                /* 1 (needs ws): only the partials are written; the caller adds them with mart_ln_dgb_reduce
                                                                      (same order, same result) -- e.g. on the weight-gradient stream, off the data-gradient chain */
  const void* add_bf16;                                            /* instead of add_f32: the residual operand as bf16 -- the gradient w.r.t. the residual stream carried in
                                                                      bf16 between layers (with bf16_total and no ds_f32: 10 bytes per element instead of 16) */
} mart_ln_bwd_desc;
int mart_ln_bwd(const mart_ln_bwd_desc* d, void* stream);
/* number of per-workgroup partial rows mart_ln_bwd writes into ws for M rows ([partials][2][H] floats: dgamma, dbeta) */
int mart_ln_bwd_partials(int M);
/* dgamma[c] += sum_g ws[g][0][c], dbeta[c] += sum_g ws[g][1][c] in a fixed order: the second half of mart_ln_bwd's deterministic path */
int mart_ln_dgb_reduce(const float* ws, int partials, int H, float* dgamma, float* dbeta, void* stream);
/* LayerNorm fold (mart_gemm_nt_desc.row_stats / ln_*; nn.LayerNorm -> nn.Linear, modeling_unimo.py:509 -> :223-225, :518 -> :284-286):
 * mart_ln_fold_prep: from the f32 weight W [N, K], its bias (may be NULL) and the LayerNorm's gamma / beta [K]:  Wf = bf16(gamma o W),
 * s[n] = sum_k Wf[n][k], bf[n] = bias[n] + sum_k beta[k] W[n][k].  mart_ln_stats_finalize: the [M][H / 64][2] partial sums a row_stats epilogue
 * wrote -> mean [M], rstd [M] = 1 / sqrt(var + eps). */
int mart_ln_fold_prep(const float* W, const float* bias, const float* gamma, const float* beta, void* Wf, float* s, float* bf, int N, int K, void* stream);
int mart_ln_stats_finalize(const float* partials, int M, int H, float eps, float* mean, float* rstd, void* stream);

/* ---------------------------------------------------------------- embeddings
 * pixels f32 [B,2,3,S,S] -> bf16 patch matrix [B*2*P, 3*p*p] (k = c*p*p + ky*p + kx): the im2col-free
 * operand of the bias-free patch conv, modeling_unimo.py:110-112,123-124. */
int mart_patchify(const float* pixels, void* out_bf16, int B, int S, int p, void* stream);
/* Device-side batch assembly (DataCollatorForSeq2Seq.__call__, MarT/data/data_module.py:126-142,161): the per-entity
 * image table [N_img,3,S,S] stays resident in HBM; index[b*2+slot] picks the row, < 0 = the all-zero image of a missing
 * slot.  Produces the same patch matrix as mart_patchify on the stacked pixel_values, without materialising them. */
int mart_patchify_gather(const float* table, const int32_t* index, void* out_bf16, int B, int S, int p, void* stream);
/* the stacked tensor itself, for callers that want pixel_values [B,2,3,S,S] */
int mart_gather_images(const float* table, const int32_t* index, float* out, int B, int S, void* stream);
/* s[b,t,:] = [cls | patch(b,0,:) | patch(b,1,:)] + pos[0,1..P,1..P]   (modeling_unimo.py:127-130) */
/* tail_shift: position row of the second image's patch t is t - tail_shift (MKGformer 0: pos[1..P]; FLAVA 1: pos[0..P-1],
 * the reference's FlavaImageEmbeddings quirk, modeling_flava.py:338) */
int mart_vision_assemble(const void* patch_bf16, const float* cls, const float* pos, float* s, int B, int P, int H, int tail_shift, void* stream);
/* backward of the assemble: dpatch (bf16) and atomically accumulated dcls, dpos */
int mart_vision_assemble_bwd(const float* ds, void* dpatch_bf16, float* dcls, float* dpos, int B, int P, int H, int tail_shift, void* stream);
/* same result, deterministic: slice partials through a caller-owned workspace ((P+1) * 16 * 2 * H floats), reduced in slice order */
int mart_vision_assemble_bwd_det(const float* ds, void* dpatch_bf16, float* dcls, float* dpos, int B, int P, int H, int tail_shift,
                                 float* ws, long long ws_bytes, void* stream);
/* BertEmbeddings.forward (modeling_unimo.py:152-186): word[ids]+type[tt]+pos[:L] -> LN(eps) -> dropout */
typedef struct {
  const int64_t* ids; const int64_t* tt;
  const float* word; const float* pos; const float* type;
  const float* gamma; const float* beta; float eps;
  float p_drop; uint64_t seed;
  int B, L, H;
  float* s_out; float* mean; float* rstd;
  float* out_f32; void* out_bf16;
  void* out_f16;                                                   /* optional fp16 copy of the output (engine.text_f16) */
} mart_text_embed_desc;
int mart_text_embed_fwd(const mart_text_embed_desc* d, void* stream);
/* dy*dropmask/(1-p) -> dyd (f32) : first half of the text-embedding backward (LN bwd follows) */
int mart_dropout_bwd_f32(const float* dy_f32, const void* dy_bf16, float* out, long long n, float p, uint64_t seed, void* stream);
/* scatter ds [B*L,H] into dword[ids], dpos[0..L), dtype[tt] (atomic) */
int mart_text_embed_scatter(const float* ds, const int64_t* ids, const int64_t* tt, float* dword, float* dpos, float* dtype,
                            int B, int L, int H, void* stream);
/* same result, deterministic (no float atomics).  order[B*L]: the flat token positions sorted by id (stable sort of ids; index
 * preparation on the host side); ws: (ceil(B*L/64) * 2 + L * 16 * 2) * H floats; meta: ceil(B*L/64) int32.  Token types 0 / 1. */
int mart_text_embed_scatter_det(const float* ds, const int64_t* ids, const int64_t* tt, const int64_t* order, float* dword, float* dpos, float* dtype,
                                int B, int L, int H, float* ws, long long ws_bytes, int32_t* meta, void* stream);

/* ---------------------------------------------------------------- attention (head_dim 64)
 * Flash-style multi-head attention, K/V tiles staged in LDS, wave-level online softmax.
 *  vision: CLIPAttention.forward, modeling_unimo.py:212-272 -- no mask, keys = [text prefix (K,V) | own]
 *  text  : BertSelfAttention.forward, :317-377 -- scores/8, adaptive analogy reweight (:342-349),
 *          additive padding mask (:355, :55-56), softmax, attention-probs dropout (:362)
 * q/k/v are column blocks of the fused projection output: element (row, h*64+d) at ptr[row*ld + h*64 + d],
 * row = b*S + i.  Prefix keys j < Lp come from pk/pv (rows b*Lp + j). */
typedef struct {
  const void* q; const void* k; const void* v; int ldq, ldk, ldv;  /* bf16 */
  const void* pk; const void* pv; int ldp; int Lp;                  /* optional text prefix */
  int B, nh, Sq, Sk;                                                /* Sk = own keys (excl. prefix) */
  float scale;
  const int64_t* attn_mask;                                         /* [B,Sk] 1 = keep, NULL = none */
  const int64_t* sep; int sep_stride;                               /* s = sep[b*sep_stride]; NULL = no reweight */
  const float* w0; const float* w1;                                 /* device scalars, clamped in-kernel */
  float p_drop; uint64_t seed;
  void* ctx; int ldctx;                                             /* bf16 out [B*Sq, nh*64] */
  float* lse;                                                       /* [B,nh,Sq] softmax statistic, LOG2 domain: log2(sum_j 2^(s_ij*log2e)) */
  int rw_skip_row0;                                                 /* FLAVA reweight variant: query row 0 ([CLS]) is not scaled (modeling_flava.py:494) */
  void* ctx_f16;                                                    /* optional: the context also as fp16 (rounded once from f32; ld = ldctx): A operand of the fp16 output projection */
} mart_attn_fwd_desc;
int mart_attn_fwd(const mart_attn_fwd_desc* d, void* stream);

typedef struct {
  mart_attn_fwd_desc f;                                             /* same inputs as forward (ctx = forward output) */
  const void* dctx; int lddctx;                                     /* bf16 */
  float* delta;                                                     /* workspace [B,nh,Sq] */
  void* dq; void* dk; void* dv; int lddq, lddk, lddv;               /* bf16 out */
  void* dpk; void* dpv; int lddp;                                   /* bf16 out for prefix keys */
  int accum_dkv;                                                    /* dk/dv += (text K/V already hold the prefix grads) */
  float* dw;                                                        /* [2] accumulated d(w0), d(w1) */
  float* dw_ws;                                                     /* optional workspace, 2 * 4 * B * nh * ceil(Sq/128) floats: per-wave partials of dw, summed by a
                                                                       second tiny kernel instead of thousands of atomics on two addresses */
} mart_attn_bwd_desc;
int mart_attn_bwd(const mart_attn_bwd_desc* d, void* stream);

/* ---------------------------------------------------------------- BertFusion as one kernel per direction (modeling_unimo.py:400-414)
 * fusion_scores = hidden @ visual^T (no scale, :405) ; fusion_probs = softmax(fusion_scores, -1) (:410) ;
 * fusion_output = fusion_probs @ visual (:411).  q = hidden [B*Lq, H], v = visual [B*Nv, H] (bf16, row-major).  The scores stay on
 * chip; probs [B*Lq, ldp] (bf16, columns Nv..ldp zero) is written for the backward pass.  Shapes: mart_fusion_supported(Lq, Nv, H)
 * (H = 768, Lq a multiple of 32, Nv <= 512 within the LDS budget); callers fall back to mart_gemm_nt + mart_softmax_* otherwise. */
typedef struct {
  const void* q; int ldq; const void* v; int ldv;
  void* out; int ldo;                                               /* fusion_output bf16 [B*Lq, H] */
  void* probs; int ldp;                                             /* bf16 [B*Lq, ldp], ldp a multiple of 8 in [Nv, 64*ceil(Nv/64)] */
  int B, Lq, Nv, H;
  void* out_f16;                                                    /* optional fp16 copy of fusion_output (ld = ldo) */
} mart_fusion_fwd_desc;
int mart_fusion_supported(int Lq, int Nv, int H);
int mart_fusion_fwd(const mart_fusion_fwd_desc* d, void* stream);
/* autograd of the three lines above: dq = d(hidden) (bf16, written), dv_f32 += d(visual) in place (the vision-stream gradient; each
 * element is updated by one wave per launch and the 64-query blocks are sequential launches: no atomics, fixed order),
 * dv_bf16 = optional bf16 copy of the updated rows.  dv_f32 == NULL (round 6): the vision-stream gradient is carried in bf16 and dv_bf16 itself is
 * read, updated and written (f32 arithmetic, one rounding). */
typedef struct {
  const void* q; int ldq; const void* v; int ldv;
  const void* dout; int lddo;                                       /* d(fusion_output) bf16 [B*Lq, H] */
  const void* probs; int ldp;
  void* dq; int lddq;
  float* dv_f32; int lddv; void* dv_bf16; int lddvb;
  int B, Lq, Nv, H;
} mart_fusion_bwd_desc;
int mart_fusion_bwd(const mart_fusion_bwd_desc* d, void* stream);

/* ---------------------------------------------------------------- row softmax (fusion op, modeling_unimo.py:410)
 * probs (bf16, ldp >= C, columns C..ldp zero-filled) = softmax(scores f32 [R,C]) */
int mart_softmax_fwd(const float* scores, int lds_, void* probs_bf16, int ldp, int R, int C, void* stream);
/* dscores (bf16, zero padded) = probs * (dprobs - sum(dprobs*probs)) */
int mart_softmax_bwd(const void* probs_bf16, int ldp, const float* dprobs, int ldd, void* dscores_bf16, int ldo, int R, int C, void* stream);
/* batched 2-D transpose with zero padding: out[b, c, r] = in[b, r, c], r < R else 0; out ld = Rp */
int mart_transpose_bf16(const void* in, int ldi, long long stride_i, void* out, int Rp, long long stride_o, int R, int C, int batch, void* stream);

/* ---------------------------------------------------------------- scoring head / loss / ranking
 * LabelSmoothSoftmaxCEV1.forward (lit_models/utils.py:42-66): per-row loss and lse over C classes.
 * ignore_index (utils.py:37,49-52,58): rows with that label have loss 0, are not counted in n_valid and get a zero gradient row.
 * Any other label outside [0, C) -- where the reference's scatter_ raises -- is never used as an index: the row's loss (and gradient
 * row) is NaN and status[0] (device int32, may be NULL; the caller zeroes it once and polls it when it likes) is set to 1.
 * reduction: MART_REDUCE_NONE (loss_rows only), MART_REDUCE_MEAN (loss_out[0] = sum / n_valid, utils.py:59-60), MART_REDUCE_SUM (:61-62);
 * loss_out is float[2] on the device: {loss, n_valid}. */
#define MART_REDUCE_NONE 0
#define MART_REDUCE_MEAN 1
#define MART_REDUCE_SUM 2
int mart_lsce_fwd(const float* logits, int ld, const int64_t* label, long long ignore_index, float eps, float* loss_rows, float* lse,
                  float* loss_out, int reduction, int* status, int R, int C, void* stream);
/* dlogits = g * (softmax * sum(target) - target), g = gscale[gscale_per_row ? row : 0] * rowscale / (n_valid ? n_valid[0] : 1); ignored rows 0;
 * bf16 (zero padded to ldo) and/or f32 */
int mart_lsce_bwd(const float* logits, int ld, const int64_t* label, long long ignore_index, const float* lse, float eps, const float* gscale,
                  int gscale_per_row, float rowscale, const float* n_valid, void* dlogits_bf16, int ldo, float* dlogits_f32, int R, int C,
                  void* stream);
/* rank = 1 + #(logit > logit[label])  == argsort(argsort(-logits))[label]+1 without ties (lit_models/transformer.py:162-164);
 * a label outside [0, C) gives rank 0 */
int mart_rank(const float* logits, int ld, const int64_t* label, int64_t* rank, int R, int C, void* stream);
/* relaxation loss rows (lit_models/transformer.py:103-108): relu(cos(q,a)) + 1 - cos(r0,r1) on rows of trans.
 * rows == NULL: trans is the dense [B,L,H] tensor.  rows != NULL (row-subset pass): trans is the COMPACT [B*nr, H] tensor and rows[b*nr + j] the flat id
 * b*L + position of its slot j; a position the pass was not promised makes that example's loss NaN (nothing is read for it). */
int mart_simloss_fwd(const float* trans, const int64_t* rel_idx, const int64_t* q_idx, const int64_t* a_idx, const int32_t* rows, int nr, float* loss_rows,
                     int B, int L, int H, void* stream);
/* dtrans (atomic +=; same layout as trans) for the four gathered rows; g = gscale[0]*rowscale */
int mart_simloss_bwd(const float* trans, const int64_t* rel_idx, const int64_t* q_idx, const int64_t* a_idx, const int32_t* rows, int nr, const float* gscale,
                     float rowscale, float* dtrans, int B, int L, int H, void* stream);

/* ---------------------------------------------------------------- row-subset passes (forward(needed_rows=...), engine.forward(rows=...))
 * rows [B, nr] int32: flat ids b*L + position of the rows example b was promised, grouped by example.
 * mart_needed_rows: the rows of a training / evaluation step built on the device in one launch -- column 0 the first position of `token` ([MASK];
 * absent: position 0 and status bit 1, as mart_find_token), then rel_idx[:,0], rel_idx[:,1], q_idx, a_idx when given (nr = 5, else 1); negative
 * positions wrap, then clamp.  mask_row_out (may be NULL) [B] = column 0. */
int mart_needed_rows(const int64_t* ids, int B, int L, int64_t token, const int64_t* rel_idx, const int64_t* q_idx, const int64_t* a_idx,
                     int32_t* rows_out, int32_t* mask_row_out, int32_t* status, void* stream);
/* out[i] = compact index (b*nr + slot, first slot that names it) of flat row id flat[i]; a row outside the promise: slot 0 of its example and status bit 2 (value 4) */
int mart_rows_lookup(const int32_t* flat, int n, const int32_t* rows, int nr, int L, int32_t* out, int32_t* status, void* stream);
/* out[i] = sum_s parts[s*n + i], s = 0 .. S-1 in order: the reduction of a split-K product run as a batched mart_gemm_nt (n % 4 == 0) */
int mart_sum_splits_f32(const float* parts, int S, long long n, float* out, void* stream);
/* dense [B*L, H] f32 view of a compact [B*nr, H] tensor: promised rows copied (first slot wins), `fill` elsewhere */
int mart_rows_dense(const float* src, const int32_t* rows, int nr, int B, int L, int H, float* dst, float fill, void* stream);

/* ---------------------------------------------------------------- small utilities
 * [MASK] position per row: (input_ids == mask_id).nonzero() without the host sync of lit_models/transformer.py:94
 * pos_out[b] = first position (-1: absent), row_out[b] (may be NULL) = b*L + max(pos, 0); status (device int32, may be NULL): bit 1 (value 2) is
 * OR-ed in when some example does not contain the token (the reference's fancy index raises a shape error there) */
int mart_find_token(const int64_t* ids, int B, int L, int64_t token, int32_t* pos_out, int32_t* row_out, int32_t* status, void* stream);
int mart_cast_f32_bf16(const float* src, void* dst, long long n, void* stream);
int mart_cast_bf16_f32(const void* src, float* dst, long long n, void* stream);
int mart_cast_f32_f16(const float* src, void* dst_f16, long long n, void* stream);   /* fp16 shadow of the text-stream weights */
int mart_cast_bf16_f16(const void* src_bf16, void* dst_f16, long long n, void* stream);   /* exact for |x| in [2^-14, 65504] */
/* dst[r, 0:C] = bf16(src[r, 0:C]), dst[r, C:ldd] = 0 */
int mart_cast_pad_f32_bf16(const float* src, int lds_, void* dst, int ldd, int R, int C, void* stream);
/* dst[r,:] = src[rows[r],:] for 2-byte (bf16) elements; H multiple of 8 */
int mart_gather_rows_bf16(const void* src, int ld, const int32_t* rows, void* dst, int R, int H, void* stream);
/* out = dy * act'(z), all bf16 (backward of the head transform activation, modeling_unimo.py:974) */
int mart_act_bwd(const void* dy_bf16, const void* z_bf16, int act, void* out_bf16, long long n, void* stream);
int mart_gather_rows_f32(const float* src, int ld, const int32_t* rows, float* dst, int R, int H, void* stream);
int mart_scatter_add_rows_f32(const float* src, const int32_t* rows, float* dst, int ld, int R, int H, void* stream);
/* ---- row subsets of the last text layer (lit_models/transformer.py:94-95,103-107 read <= 5 rows per example of trans_hidden_states; nothing reads
 * text layer 11's other rows: modeling_unimo.py:616 exports K/V of layers idx-1 <= 10 only).  rows[R] are flat row ids, `group` consecutive entries
 * belong to one example.
 * mart_gather_rows_first_f32: dst[r] = src[rows[r]], or 0 when an EARLIER slot of the same group names the same row (the gradient of a row that was
 * requested twice is taken once).
 * mart_scatter_rows: slots of a group applied in order by one workgroup (deterministic); src f32, dst f32 or bf16 (dst_bf16 != 0: sums are formed
 * in f32 and rounded once).  accumulate = 1: dst[rows[r]] += src[r] (gradient rows; a row named twice receives both); accumulate = 0:
 * dst[rows[r]] = src[r] and the FIRST slot naming a row wins (forward values). */
int mart_gather_rows_first_f32(const float* src, int ld, const int32_t* rows, int group, float* dst, int R, int H, void* stream);
int mart_scatter_rows(const float* src, const int32_t* rows, int group, void* dst, int ld, int dst_bf16, int accumulate, int R, int H, void* stream);
/* out = a (+ b) with mixed dtypes; used to merge gradient contributions */
int mart_add_f32_bf16(const float* a, const void* b_bf16, float* out_f32, void* out_bf16, long long n, void* stream);
/* debug/test hook: the dropout keep-mask the kernels derive from (seed, index) */
int mart_dropout_mask(uint8_t* out, long long n, float p, uint64_t seed, void* stream);

/* ---------------------------------------------------------------- optimizer (lit_models/transformer.py:224-241)
 * Multi-tensor AdamW over the flat parameter buffer; also refreshes the bf16 shadow the GEMMs read.
 * chunks: int32 triples (start, length, flags), start/length in elements; flags bit 0 = weight decay applies, bit 1 = the chunk also
 * refreshes the fp16 shadow (forward weights of the text stream, engine.text_f16). */
typedef struct {
  float* master; float* grad /* read; written (zeros) only with zero_grad */; float* m; float* v; void* shadow_bf16;
  const int32_t* chunks; int n_chunks;
  float lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale;
  void* shadow_f16;                                                 /* optional fp16 shadow, same layout as master (written for chunks with flag bit 1) */
  int zero_grad;                                                    /* 1: the gradient elements of the updated chunks are written back as 0 (the step's zero-fill folded into the update) */
} mart_adamw_desc;
int mart_adamw(const mart_adamw_desc* d, void* stream);
/* batched bf16 transposes described by int64 quadruples (src_off, dst_off, rows, cols) into the W^T shadow */
int mart_transpose_table(const void* src_bf16, void* dst_bf16, const int64_t* table, int n, void* stream);
/* row-major [rows, cols] -> tile-blocked [rows/256][cols/64][256][64] per table entry (src_off, dst_off, rows, cols); rows % 256 == 0, cols % 64 == 0 */
int mart_block_table(const void* src_bf16, void* dst_bf16, const int64_t* table, int n, void* stream);

/* ---------------------------------------------------------------- fp32-accurate evaluation path (csrc/precise.hip)
 * The reference is fp32 end to end.  Precise mode keeps activations in fp32 and runs every dense contraction through
 * mart_gemm_nt on two-term bf16 splits laid out K-concatenated (K' = 3K):  role 0 (activations) [hi|lo|hi],
 * role 1 (weights) [hi|hi|lo]  =>  hi*hi + lo*hi + hi*lo accumulated in fp32. */
int mart_split_bf16x3(const float* src, long long ld, void* dst_bf16 /* [rows, 3K] (terms 2) or [rows, 6K] (terms 3) */, int rows, int K, int role, int terms, void* stream);
/* terms = 3 (verification mode): x = h + m + l, six products hh + hm + mh + mm + hl + lh (exact to ~2^-24): role 0 [h|h|m|m|h|l], role 1 [h|m|h|m|l|h] */
/* ... of gathered source rows: dst row r = split(src[gather[r]]).  The scoring head of the TRAINING path runs on these too (mask rows of
 * trans_hidden x the scored rows of the tied embedding, modeling_unimo.py:958): 2 x 768 x (B + ids) extra MFMA flops per step buy back the
 * third of the bf16 logit error that the head alone contributed (tools/error_budget.py). */
int mart_split_bf16x3_rows(const float* src, long long ld, const int32_t* gather, void* dst_bf16 /* [rows, 3K | 6K] */, int rows, int K, int role, int terms, void* stream);
/* f32 twins of mart_patchify / mart_patchify_gather (index NULL: pixels are [B,2,3,S,S]; else table rows, -1 = zero image) and mart_vision_assemble */
int mart_patchify_f32(const float* pixels_or_table, const int32_t* index, float* out, int B, int S, int p, void* stream);
int mart_vision_assemble_f32(const float* patch, const float* cls, const float* pos, float* s, int B, int P, int H, int tail_shift, void* stream);
/* fp32 attention, plain FMA arithmetic: CLIPAttention (modeling_unimo.py:212-272, prefix keys), BertSelfAttention
 * (:317-377, scale, adaptive reweight, additive mask) with D = 64, and BertFusion (:400-414) with nh = 1, D = 768, scale = 1 */
typedef struct {
  const float* q; const float* k; const float* v; long long ldq, ldk, ldv;   /* rows [B*S, ld]; head h at column h*D */
  const float* pk; const float* pv; long long ldp; int Lp;
  int B, nh, D, Sq, Sk;
  float scale;
  const int64_t* attn_mask; const int64_t* sep; int sep_stride; const float* w0; const float* w1; int rw_skip_row0;
  float* ctx; long long ldctx;
  int fast;                                                        /* 1 (evaluation passes): head-dim-64 calls may run on two-term bf16 operand splits
                                                                      (three products, 2^-16 relative: the arithmetic of the path's GEMMs) instead of exact f32 */
  void* ctx_split3; long long ldctx3;                              /* optional, fast path only: the context ALSO as [hi | lo | hi] bf16 rows of 3 * nh * D columns
                                                                      (row stride ldctx3 elements): the A operand of the output projection, no separate split pass */
} mart_attn_f32_desc;
int mart_attn_fwd_f32(const mart_attn_f32_desc* d, void* stream);

/* ---- fp32-accurate backward (verification mode: model.set_precision("fp32") in a training step; engine_precise.PreciseUnimoTrain).
 * Autograd of the same reference lines as the forward entry points above.  dq is written; dk / dv / dpk / dpv are ACCUMULATED (atomics: zero
 * them first; the prefix gradients land in the text layer's k / v gradient blocks); dw[2] accumulates d(adaptive_weight.0 / .1) with
 * torch.clamp's sub-gradient (modeling_unimo.py:342-349). */
typedef struct {
  mart_attn_f32_desc f;                                            /* the forward call (ctx unused) */
  const float* dctx; long long lddctx;
  float* dq; float* dk; float* dv; long long lddq, lddk, lddv;
  float* dpk; float* dpv; long long lddp;
  float* dw;
} mart_attn_bwd_f32_desc;
int mart_attn_bwd_f32(const mart_attn_bwd_f32_desc* d, void* stream);
/* row-stacked two-term splits for weight gradients through mart_gemm_tn (contraction over 3M rows): role 0 (X) [hi;lo;hi], role 1 (Y) [hi;hi;lo] */
int mart_split_bf16x3_stack(const float* src, long long ld, void* dst_bf16 /* [3M | 6M, K] */, int M, int K, int role, int terms, void* stream);
int mart_act_f32(const float* z, float* a, int act, long long n, void* stream);                       /* a = act(z), erf GELU / quick-GELU in f32 */
int mart_act_bwd_f32(const float* dy, const float* z, int act, float* dz, long long n, void* stream);  /* dz = dy * act'(z) */
int mart_colsum_f32(const float* src, long long ld, float* out /* [C], accumulated */, int R, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif
