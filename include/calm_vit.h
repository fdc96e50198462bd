/*
 * calm_vit.h — C-ABI of libcalmvit_hip.so, the MI355X (gfx950) kernel library behind the
 * CALM-ViT cross-axial latent-masking attention path.
 *
 * The reference (focegueda1998/CALM-ViT-DTE) is pure Python: its "FFI" for this path is the
 * set of ATen ops that CALM-ViT/Vi_Tools_CNN_less_V2.py and CALM-ViT/CALM_ViT_V2.py call.
 * Each entry point below replaces one such call site (cited as file:line into
 * /root/reference/CALM-ViT/) and is bound from Python with ctypes
 * (calm-vit-dte_amd/_lib.py; see INTEGRATION.md for the reference-side stub).
 *
 * Conventions
 *  - plain device pointers + sizes; element strides unless stated; no torch types.
 *  - every call only ENQUEUES work on `stream` (hipStream_t passed as void*): no allocation,
 *    no host synchronisation, re-entrant, hipGraph-capturable.
 *  - return value: 0 = ok, <0 = invalid argument / unsupported shape (CALM_E_*),
 *    >0 = hipError_t of the failed launch.  No C++ exceptions cross the boundary.
 *  - NaN/Inf propagate (GradScaler's inf check relies on it, distributed_trainer_cls.py:88,93).
 *  - tensors are fp32 unless an entry point takes a storage-type argument (CALM_ST_*; ABI v4: the bf16 pipeline).
 */
#ifndef CALM_VIT_H
#define CALM_VIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CALM_ABI_VERSION 7

#define CALM_E_INVAL   (-1)   /* null pointer / negative size                 */
#define CALM_E_LAYOUT  (-2)   /* stride pattern the kernel cannot address     */
#define CALM_E_UNSUPP  (-3)   /* combination of options not implemented       */

/* calm_gemm_args.dtype — tensors are fp32 in memory in every mode; the mode selects the matrix pipe:
 *   CALM_F32    exact fp32 MFMA (v_mfma_f32_32x32x2_f32), bit-for-bit an fmaf chain
 *   CALM_BF16   operands rounded to bf16 in LDS, fp32 accumulate (what autocast(bfloat16) computes)
 *   CALM_BF16X3 operands split hi+lo bf16, 3 MFMA passes, fp32 accumulate: ~2^-17 relative product error */
enum { CALM_F32 = 0, CALM_BF16 = 1, CALM_BF16X3 = 2 };
enum { CALM_ST_F32 = 0, CALM_ST_BF16 = 1, CALM_ST_FP8_E4M3 = 2, CALM_ST_FP8_E5M2 = 3 };   /* storage type of a tensor in HBM
                                                          (fp8: OCP e4m3fn / e5m2 as gfx950 implements them) */
enum { CALM_ACT_NONE = 0, CALM_ACT_GELU = 1, CALM_ACT_GELU_BWD = 2 };

int         calm_abi_version(void);
const char* calm_build_info(void);     /* "gfx950 ..." */

/* ---------------------------------------------------------------------------------------
 * Strided, batched GEMM with fused epilogue on the matrix cores (v_mfma_f32_32x32x2_f32).
 *
 *   acc(m,n)  = sum_k A(m,k) * B(n,k)                 A(m,k) = A[m*a_rs + k*a_cs + b0*a_b0 + b1*a_b1]
 *   z         = acc * alpha / (inv_scale ? *inv_scale : 1) + (bias ? bias[n] : 0)
 *   C_pre(m,n)= z                                      (optional second output, C's layout)
 *   y         = act == GELU      ? gelu_erf(z)
 *             : act == GELU_BWD  ? z * gelu_erf'(aux(m,n))   (aux: C's layout)
 *             : z
 *   C(m,n)    = y * (col_scale ? col_scale[n] : 1) + (residual ? residual(m,n) : 0) [+ C(m,n) if accumulate]
 *
 * One of (a_rs, a_cs) must be 1, likewise (b_rs, b_cs); C is n-contiguous.
 * Operand row / column strides must be below 2^21 elements (CALM_E_UNSUPP otherwise): a tile is addressed with
 * 32-bit byte offsets from a per-tile base.
 * batch = batch0*batch1 independent problems (b0,b1 strides per operand; 0 = broadcast).
 * reduce_batch: all batches are summed into ONE C (c_b0/c_b1 ignored) — the weight gradient of
 * the sequence-axis linears.  split_k: 0 = library picks, 1 = off, >1 = that many K-slices; slices
 * and reduce_batch combine through fp32 atomics and then allow alpha/inv_scale only.
 *
 * Grouped form (n_group = batch0 in 1..CALM_GEMM_MAX_GROUP; batch1 must be 1): the b0 entries are separate
 * allocations — entry g uses A_group[g], B_group[g], C_group[g] (a table whose first entry is NULL falls back to
 * base + g*stride) and its own spectral-norm scale inv_scale_group[g] (a NULL table entry = 1).
 *   without reduce_batch: n_group independent products in ONE launch (the q/k/v projections of a block read one
 *     activation: Vi_Tools_CNN_less_V2.py:265-267), each divided by its sigma in the epilogue;
 *   with reduce_batch:    C = sum_g A_g B_g^T / sigma_g in one pass over the concatenated reduction (their
 *     input gradient); runs unsplit — deterministic, full epilogue — unless split_k > 1.
 * Grouped launches take no C_pre / aux / residual.
 *
 * Replaces: nn.Linear / matmul / bmm call sites Vi_Tools_CNN_less_V2.py:226-231, 251-267, 276-277,
 * 288-290 (raw QK^T + linear_mask), 293-298 (QK^T, PV of SDPA), 300, 305-308, 312;
 * CALM_ViT_V2.py:76 (head), 1x1 convs Vi_Tools:380,384 — and all their autograd backward GEMMs.
 * ------------------------------------------------------------------------------------- */
typedef struct calm_gemm_args {
    const void* A; const void* B; void* C;
    int32_t M, N, K;
    int32_t batch0, batch1;
    int64_t a_rs, a_cs, a_b0, a_b1;
    int64_t b_rs, b_cs, b_b0, b_b1;
    int64_t c_rs, c_b0, c_b1;
    float        alpha;
    const float* inv_scale;      /* device scalar (spectral-norm sigma) or NULL */
    const float* bias;           /* [N] or NULL */
    const float* col_scale;      /* [N] or NULL (LayerScale ls_att / ls_mlp) */
    const void*  residual; int64_t r_rs, r_b0, r_b1;
    void*        C_pre;          /* optional pre-activation output */
    const void*  aux;            /* GELU_BWD: saved pre-activation */
    int32_t act;
    int32_t accumulate;
    int32_t reduce_batch;
    int32_t split_k;
    int32_t dtype;
    int32_t n_group;             /* 0 = plain strided batches */
    const void*  A_group[4];
    const void*  B_group[4];
    void*        C_group[4];
    const float* inv_scale_group[4];
    void*        workspace;      /* optional (ABI v3): device scratch for split launches, see calm_gemm_workspace_bytes */
    int64_t      workspace_bytes;
    /* ABI v4 — storage type of each tensor in HBM (CALM_ST_F32 / CALM_ST_BF16; strides stay in ELEMENTS of the tensor).
     * bf16 tensors are accepted by the bf16 matrix pipe only (dtype == CALM_BF16): the bf16 pipeline keeps the
     * activations that only GEMMs consume (LayerNorm outputs, MLP hidden states, attention outputs, their
     * gradients) and a per-step copy of the weights as bf16 — rounding an operand when it is stored instead of
     * every time a tile of it is staged: identical products, half the operand bytes.  A bf16 operand needs its
     * contiguous extent (K if k-contiguous, else its M / N), its other stride and its batch strides to be
     * multiples of 8 elements and a 16-byte aligned base (CALM_E_LAYOUT otherwise).  C_pre has C's type.
     * Split / batch-reduced outputs (fp32 atomics or workspace) must be fp32. */
    int32_t      a_type, b_type, c_type, aux_type, r_type;
    int32_t      reserved_;
    /* fp8 operands (BASELINE configs[4], "bf16 + fp8 MFMA GEMMs"): a_type in {FP8_E4M3, FP8_E5M2}, b_type FP8_E4M3, both
     * k-contiguous with K and row strides multiples of 16, dtype == CALM_BF16, no groups / k-split; a_dq / b_dq are the
     * DEVICE dequantisation factors written by calm_quantize_fp8 (amax / FP8_MAX): the epilogue multiplies the fp32
     * accumulator by a_dq * b_dq before alpha, sigma, bias ...  Products run on v_mfma_f32_32x32x16_{fp8,bf8}_fp8. */
    const float* a_dq; const float* b_dq;
} calm_gemm_args;
#define CALM_GEMM_MAX_GROUP 4

int calm_gemm(const calm_gemm_args* args, void* stream);
/* Bytes of `workspace` with which this launch would combine its k-slices through per-slice partial tiles and one
 * reduction pass instead of fp32 atomics (0: the launch is not split, or atomics are as fast — large outputs).
 * Passing less (or NULL) is always valid: the launch then uses atomics.  The caller owns the buffer; it is only used
 * until the launch's kernels have run on `stream`. */
int64_t calm_gemm_workspace_bytes(const calm_gemm_args* args);

/* ABI v6 — kernel-family switches of calm_gemm (process-wide tuning knobs, not part of the reference's interface: every
 * setting computes the same products to rounding).  Returns the previous value, CALM_E_INVAL for an unknown option.
 *   CALM_GEMM_OPT_PIPE    1 (default; env CALM_GEMM_PIPE):  bf16-tensor launches on the pipelined persistent family
 *   CALM_GEMM_OPT_PIPE32  0 (default; env CALM_GEMM_PIPE32): fp32-tensor launches on the same pipeline
 *                         (gemm_f32p_kernel) — 1: every eligible launch, 2: k-contiguous operand pairs only.  Off by
 *                         default: the fp32 matrix pipe is clock-limited under sustained load and both families end
 *                         at the same rate. */
/*   CALM_GEMM_OPT_DETERMINISTIC (ABI v7) 0 (default; env CALM_GEMM_DETERMINISTIC): 1 = EVERY k-split / batch-reduced
 *                         launch asks for a workspace (calm_gemm_workspace_bytes) and combines its slices through
 *                         per-slice partial tiles and one fixed-order reduction pass — no fp32 atomics, the weight
 *                         gradients repeat bit for bit (a launch that is handed no workspace still uses atomics).
 *                         Off by default: below ~48 slices per output the atomics are faster (the partial tiles
 *                         are written and read back once more).  The forward and input-gradient products never split:
 *                         they are reproducible in either mode. */
#define CALM_GEMM_OPT_PIPE   0
#define CALM_GEMM_OPT_PIPE32 1
#define CALM_GEMM_OPT_DETERMINISTIC 2
int calm_gemm_set_option(int32_t option, int32_t value);

/* ABI v7 — what calm_gemm would launch for `args` (nothing is enqueued): kernel family, tile, work decomposition.
 * Diagnostic surface: tests use it to map a wrong output element back to (tile, persistent workgroup, XCD, wave,
 * 16-row strip) — see tests/locate.py — and the bench to label its per-shape table.
 *   family  0 fp32 128-row tiles (gemm_f32_kernel)        3 bf16 pipelined persistent (gemm_bf16p_kernel)
 *           1 bf16-operand 128-row tiles                   4 fp32 pipelined persistent (gemm_f32p_kernel)
 *           2 bf16-operand 256x128 tiles                   5 fp8
 *   items = tiles_m * tiles_n * (batch entries or k-slices); the persistent families launch min(items, 256) workgroups
 *   that walk items in the XCD-aware order of gemm_bf16p.h::decode, the others one workgroup per (tile, y). */
typedef struct calm_gemm_plan {
    int32_t family;
    int32_t tile_m, tile_n, tile_k;
    int32_t tiles_m, tiles_n;
    int32_t k_slices;          /* k-slices per output (1: not split) */
    int32_t items;
    int32_t grid;              /* workgroups of the main launch */
    int32_t epi_unit;          /* pipelined families: columns per lane of the row-layout epilogue (4 / 8); else 0 */
    int32_t uses_workspace;    /* slices combined through args->workspace (1) or fp32 atomics (0) */
    int32_t threads;           /* threads per workgroup */
} calm_gemm_plan;
int calm_gemm_describe(const calm_gemm_args* args, calm_gemm_plan* plan);

/* ---------------------------------------------------------------------------------------
 * ABI v7 — fixed-order cross-workgroup reductions.  Every entry point whose result sums over workgroups (LayerNorm dw,
 * RoPE d_inv_freq, bias column sums, the latent KL sum, the CNN tail's weight gradients) takes `partials`: device
 * scratch of at least calm_reduce_scratch_floats(op, rows, cols) floats (16-byte aligned).  Workgroup g writes one
 * row of partial sums; a second launch of the same call adds the rows in workgroup order.  No atomics remain outside
 * calm_gemm's optional k-split (CALM_GEMM_OPT_DETERMINISTIC) and the standalone calm_dwconv3x3_bwd helper: forward and
 * backward of the path repeat bit for bit.  The scratch is only used until the call's kernels have run on `stream`.
 *   op                        rows                 cols
 *   CALM_RED_LAYERNORM_BWD    rows                 D
 *   CALM_RED_ROPE_BWD         B*S*H                dr
 *   CALM_RED_LATENT_FWD       rows                 mvh
 *   CALM_RED_COLSUM           rows                 cols
 *   CALM_RED_CNN_BWD          B                    S
 * ------------------------------------------------------------------------------------- */
enum { CALM_RED_LAYERNORM_BWD = 0, CALM_RED_ROPE_BWD = 1, CALM_RED_LATENT_FWD = 2, CALM_RED_COLSUM = 3, CALM_RED_CNN_BWD = 4 };
int64_t calm_reduce_scratch_floats(int32_t op, int64_t rows, int32_t cols);

/* ---------------------------------------------------------------------------------------
 * LayerNorm(D, eps, bias=False) over the last axis (Vi_Tools:131-132,197,494; fwd 211-215,311,523).
 * x,y: [rows, D] contiguous.  mean,rstd: [rows] saved for backward.
 * bwd: dx = rstd*(w*dy - mean_D(w*dy) - xhat*mean_D(w*dy*xhat)) [+ dx_add]; dw[D] += sum_rows dy*xhat
 * (added onto the caller's dw in a fixed order through `partials`, see calm_reduce_scratch_floats).  dx_add (nullable, x's layout): gradient arriving
 * through the skip connection that bypasses the norm (x feeds LN and the block's residual add, Vi_Tools:209-211,
 * 309-315), summed into dx here instead of by a separate elementwise pass.
 * y_type / dy_type (CALM_ST_*, ABI v4): the normalised output and the gradient arriving for it may be bf16 tensors
 * (bf16 pipeline: the output only feeds GEMMs; statistics, x and dx stay fp32).
 * ------------------------------------------------------------------------------------- */
int calm_layernorm_fwd(const float* x, const float* w, void* y, float* mean, float* rstd,
                       int64_t rows, int32_t D, float eps, int32_t y_type, void* stream);
int calm_layernorm_bwd(const void* dy, const float* x, const float* w, const float* mean,
                       const float* rstd, float* dx, float* dw, const float* dx_add, int64_t rows, int32_t D,
                       int32_t dy_type, float* partials /* ABI v7 */, void* stream);

/* ---------------------------------------------------------------------------------------
 * Learned-frequency NeoX RoPE + head assembly (Vi_Tools:80-95 applied at 275-285).
 * out[b,s,h, 0:dc]      = content[b,s,h,:]               (dc may be 0: plain blocks)
 * out[b,s,h, dc:dc+dr]  = rope(xr[b,s,h,:], t=s, inv_freq[dr/2])
 * content: [B,S,H,dc], xr: [B,S,H,dr], out: [B,S,H,dc+dr], all contiguous.
 * table: [2, S, dr/2] workspace; fwd fills it with cos|sin(t * inv_freq) (rebuilt every forward
 * because inv_freq is a learned parameter, Vi_Tools:70-71,86-91) and bwd reads it back.
 * bwd: d_content, d_xr from d_out; d_inv_freq[dr/2] += ... (fixed order through `partials`; dr/2 <= 256 and tensors
 * below 2^31 elements, CALM_E_UNSUPP otherwise).
 * Each tensor may be fp32 or bf16 (type arguments): in the bf16 pipeline the projections write bf16 and the
 * attention reads bf16 q / k; the rotation itself, the table and d_inv_freq are fp32.
 * ------------------------------------------------------------------------------------- */
int calm_rope_fwd(const void* content, const void* xr, const float* inv_freq, float* table, void* out,
                  int32_t B, int32_t S, int32_t H, int32_t dc, int32_t dr,
                  int32_t content_type, int32_t xr_type, int32_t out_type /* CALM_ST_*, ABI v4 */, void* stream);
int calm_rope_bwd(const void* d_out, const void* xr, const float* table,
                  void* d_content, void* d_xr, float* d_inv_freq,
                  int32_t B, int32_t S, int32_t H, int32_t dc, int32_t dr,
                  int32_t dout_type, int32_t xr_type, int32_t dcontent_type, int32_t dxr_type,
                  float* partials /* ABI v7 */, void* stream);

/* ---------------------------------------------------------------------------------------
 * Row softmax of the masked logits and its backward (the softmax inside
 * F.scaled_dot_product_attention, Vi_Tools:293-298).  In place.  rows x cols contiguous.
 * bwd: dp <- p * (dp - sum_j p*dp).
 * calm_sum_heads: dm[b,i,j] = sum_h dl[b,h,i,j]  (gradient of the head-broadcast mask, :291).
 * ------------------------------------------------------------------------------------- */
int calm_softmax_fwd(float* x, int64_t rows, int32_t cols, void* stream);
int calm_softmax_bwd(const float* p, float* dp, int64_t rows, int32_t cols, void* stream);
/* the two steps above in one pass for P, dP: [B,H,Sq,cols]: dP <- softmax backward, dm[b,i,j] = sum_h dP[b,h,i,j] */
int calm_softmax_bwd_heads(const float* p, float* dp, float* dm, int32_t B, int32_t H, int32_t Sq, int32_t cols,
                           void* stream);
int calm_sum_heads(const float* dl, float* dm, int32_t B, int32_t H, int64_t per_head, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused cross-axial latent-mask attention, forward (Vi_Tools:288-299):
 *   R = Q_all K_all^T;  M = W2 gelu(W1 R^T + b1) + b2 (along the key axis, weights divided by their
 *   sigmas);  out_h = softmax(Q_h K_h^T / sqrt(hd) + M) V_h  — the mask is produced and applied in-kernel.
 * q:[B,Sq,H*hd]  k,v:[B,Skv,H*hd]  w1:[2Skv,Skv] b1:[2Skv]  w2:[Skv,2Skv] b2:[Skv]  out:[B,Sq,H*hd].
 * Saved for backward: R:[B,Sq,Skv], hp/hg:[B,Sq,2Skv] (mask-MLP hidden before/after GELU), the mask
 * Mk:[B,Sq,Skv] and, if P != NULL, the probabilities P:[B,H,Sq,Skv].
 * calm_attention_fwd_supported() tells whether the shape has a fused instantiation
 * (Sq,Skv multiples of 16 with Skv/16 in {2,3,5,8,11,14}, hd%4==0, hd<=128); otherwise the caller
 * composes the same result from calm_gemm + calm_softmax_fwd.
 * ------------------------------------------------------------------------------------- */
int calm_attention_fwd_supported(int32_t Sq, int32_t Skv, int32_t H, int32_t hd);
int calm_attention_fwd(const float* q, const float* k, const float* v, const float* w1, const float* b1,
                       const float* s1, const float* w2, const float* b2, const float* s2, float* out, float* R,
                       float* hp, float* hg, float* Mk, float* P, int32_t B, int32_t Sq, int32_t Skv, int32_t H, int32_t hd,
                       void* stream);

/* Backward of the attention core for the same shapes (two launches: query side, key/value side):
 *   dP = dO_h V_h^T; dS = P o (dP - rowsum(P o dP)); dM = sum_h dS; dQ_h = dS K_h / sqrt(hd);
 *   dV_h = P^T dO_h; dK_h = dS^T Q_h / sqrt(hd).
 * dS:[B,H,Sq,Skv] is scratch written and re-read by the call; dq,dk,dv,dM are written (not accumulated).
 * The mask-MLP backward (through dM) and the dR = dM-path terms are calm_gemm calls of the caller. */
int calm_attention_bwd_preferred(int32_t Sq, int32_t Skv, int32_t H, int32_t hd);   /* supported AND measured faster */
int calm_attention_bwd(const float* q, const float* k, const float* v, const float* dout, const float* P, float* dS,
                       float* dq, float* dk, float* dv, float* dM, int32_t B, int32_t Sq, int32_t Skv, int32_t H,
                       int32_t hd, void* stream);

/* ---------------------------------------------------------------------------------------
 * The same attention on the bf16 matrix pipe (ABI v4; the bf16 pipeline — what autocast(bfloat16) makes of
 * Vi_Tools:288-299): q, k, v, out and the saved R / hp / hg / Mk are bf16 tensors; w1 [2S,S] / w2 [S,2S] are the
 * step's bf16 weight copies; biases, sigmas and the row log-sum-exp lse [B,H,S] are fp32.  Accumulation and softmax
 * are fp32 (v_mfma_f32_16x16x32_bf16).  Every shape with S % 8 == 0, S <= 384, hd % 4 == 0, hd <= 128 is
 * supported (keys padded to 32 in-kernel with mask -inf, head dims to 32 with zero columns).  The probabilities are
 * NOT stored: the backward recomputes them from q, k, the mask and lse.  The mask is saved twice, Mk [B,query,key]
 * and MkT [B,key,query] (the query-side / key-side backward passes keep queries / keys on the MFMA lanes).
 *   calm_attention16_bwd: dq, dk, dv (bf16, written), dM [B,S,S] (bf16, written; the caller's mask-MLP backward
 *   continues from it); delta [B,H,S] fp32 scratch (rowsum(dO o O) = rowsum(P o dP), written by the query-side
 *   pass, read by the key-side pass); out = the forward's output.
 * Two kernel generations sit behind these entry points, chosen by shape (same arithmetic, same rounding points):
 * S <= 224 with hd <= 64 (every stage of Base-224) runs the LDS-DMA pipelined kernels of round 3
 * (csrc/attention_bf16_fwd2.h, csrc/attention_bf16_bwd2.h), everything else the register-staged ones.  No entry point
 * uses atomics: results repeat bit for bit.  With B % 8 == 0 the workgroups of one image are dealt to one XCD (L2 reuse
 * of its K / V); any B is valid.
 * ------------------------------------------------------------------------------------- */
int calm_attention16_supported(int32_t S, int32_t H, int32_t hd);
int calm_attention16_fwd(const void* q, const void* k, const void* v, const void* w1, const float* b1, const float* s1,
                         const void* w2, const float* b2, const float* s2, void* out, void* R, void* hp, void* hg,
                         void* Mk, void* MkT, float* lse, int32_t B, int32_t S, int32_t H, int32_t hd, void* stream);
int calm_attention16_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const void* Mk,
                         const void* MkT, const float* lse, float* delta, void* dq, void* dk, void* dv, void* dM,
                         int32_t B, int32_t S, int32_t H, int32_t hd, void* stream);

/* ---------------------------------------------------------------------------------------
 * Latent bottleneck sampling (Vi_Tools:232-242) + KL partial sum (Vi_Tools:24-25).
 * mv: [rows, 2*mvh] (mean | raw).  std = softplus(raw)+1e-6;  z = mean + noise*std (noise NULL
 * in eval: z = mean).  kl_sum (device scalar) += sum(1 + 2 log std - mean^2 - std^2) — block partials through
 * `partials`, added in block order (ABI v7: reproducible; rounds 1-3 used fp32 atomics).
 * bwd: dmv from dz and the scalar d(kl_sum) (device pointer).
 * ------------------------------------------------------------------------------------- */
int calm_latent_fwd(const float* mv, const float* noise, float* z, float* std_out, float* kl_sum,
                    int64_t rows, int32_t mvh, float* partials /* ABI v7 */, void* stream);
int calm_latent_bwd(const float* dz, const float* d_kl_sum, const float* mv, const float* noise,
                    const float* std_in, float* dmv, int64_t rows, int32_t mvh, void* stream);

/* ---------------------------------------------------------------------------------------
 * Spectral norm (hook-based torch.nn.utils.spectral_norm wrapped around every Linear/Conv:
 * Vi_Tools:137-205,380-384; CALM_ViT_V2.py:50-52,62-66), batched over a table of layers.
 * training!=0: v <- normalize(W^T u); u <- normalize(W v); sigma = u.(W v)   (in place on u,v)
 * training==0: sigma = u.(W v) with the stored u,v.
 * Every reduction has a fixed order (no atomics): u, v, sigma are bit-reproducible, hence identical on every
 * data-parallel rank that holds the same weights (what DDP's per-forward buffer broadcast guarantees in the reference).
 * Three launches cover ALL layers of the model (the reference issues 882 mv + 883 div per forward).
 *
 * Usage: fill a host array of calm_sn_layer (device pointers inside), call calm_sn_plan() to get
 * the size of / fill a host "plan" blob, copy the blob to device memory once (pointers must stay
 * valid), then call calm_sn_power_iter(plan_dev, ...) every step with `scratch` of
 * plan_info.scratch_floats floats.
 * ------------------------------------------------------------------------------------- */
typedef struct calm_sn_layer {
    const float* w;      /* [rows, cols] row-major (conv weights flattened over dim 0) */
    float* u;            /* [rows] */
    float* v;            /* [cols] */
    float* sigma;        /* [1] */
    int32_t rows, cols;
} calm_sn_layer;

typedef struct calm_sn_plan_info {
    int64_t blob_bytes;      /* size of the plan blob */
    int64_t scratch_floats;  /* size of the per-call scratch */
    int32_t n_layers, n_work;
    int32_t n_work_a, reserved;   /* ABI v4: work items of the column-owning W^T u pass */
} calm_sn_plan_info;

/* blob_host == NULL: only fills *info.  Otherwise writes blob_bytes bytes to blob_host. */
int calm_sn_plan(const calm_sn_layer* layers, int32_t n, void* blob_host, calm_sn_plan_info* info);
int calm_sn_power_iter(const void* plan_dev, const calm_sn_plan_info* info, int32_t training,
                       float eps, float* scratch, void* stream);

/* ---------------------------------------------------------------------------------------
 * bf16 copies of many fp32 tensors in ONE launch (ABI v4): the bf16 pipeline refreshes the bf16 copy of every weight
 * once per forward (W_orig itself, not W_orig/sigma: the GEMM epilogues keep dividing by sigma in fp32), next to the
 * power iteration.  entries_dev: table in device memory; work items are chunks of calm_cast_chunk_elems() consecutive
 * elements of one entry: chunk_entry_dev[k] = entry of chunk k, entry.chunk0 = its first chunk.  Round to nearest even
 * (what autocast's weight cast does inside `with autocast(device_type="cuda", dtype=torch.bfloat16)`,
 * distributed_trainer_cls.py:84-85: torch `.to(bfloat16)` of every Linear weight per forward).
 * ------------------------------------------------------------------------------------- */
typedef struct calm_cast_entry {
    const float* src;
    void*        dst;          /* bf16[numel] */
    int64_t      numel;
    int32_t      chunk0, reserved;
} calm_cast_entry;
int32_t calm_cast_chunk_elems(void);
int calm_cast_bf16(const calm_cast_entry* entries_dev, const int32_t* chunk_entry_dev, int32_t n_chunks, void* stream);
/* Per-tensor fp8 quantisation (ABI v4): q[i] = fp8(x[i] * FP8_MAX / amax(|x|)) with FP8_MAX = 448 (e4m3) / 57344 (e5m2),
 * state[0] <- amax, state[1] <- amax / FP8_MAX (the dequantisation factor calm_gemm takes as a_dq / b_dq); x fp32 or
 * bf16, n % 4 == 0.  Two launches (amax, then convert: just-in-time scaling, no history).
 * calm_transpose_u8: out[c][r] = in[r][c] on bytes — the transposed fp8 weight copy of the input-gradient product. */
int calm_quantize_fp8(const void* x, int32_t x_type, int64_t n, void* q, int32_t q_type, float* state, void* stream);
int calm_transpose_u8(const void* in, void* out, int32_t rows, int32_t cols, void* stream);
/* one tensor: dst[i] = bf16(src[i]) — the fp32 residual-stream gradient that two backward GEMMs are about to read */
int calm_cast_bf16_one(const float* src, void* dst, int64_t n, void* stream);
/* ABI v7 — dst[i] = (float)src[i] for a bf16 tensor: the fp32 copies of the rare bf16 launch whose strides rule out
 * 16-byte staging (10-class head, 36-token stage of the fixture models) — rounds 1-3 used torch casts there */
int calm_cast_f32_one(const void* src, float* dst, int64_t n, void* stream);

/* Weight gradient through W = W_orig / sigma (and an optional LayerScale on the output):
 *   d_ls[c]  = sum_k G[c,k] * W_orig[c,k] / sigma            (only if ls != NULL; written)
 *   G0       = (ls ? ls[c] : 1) * G
 *   dW_orig  = (G0 - <G0, W_orig/sigma> u v^T) / sigma       (written)
 * G: [rows, cols] = dY^T X.  scratch: >= rows+2 floats. */
int calm_sn_weight_bwd(const float* G, const float* w_orig, const float* u, const float* v,
                       const float* sigma, const float* ls, float* d_w_orig, float* d_ls,
                       int32_t rows, int32_t cols, float* scratch, void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimizer-side step over ALL parameters in three launches (distributed_trainer_cls.py:88-96,158):
 * GradScaler.unscale_ + inf/NaN check, clip_grad_norm_(max_norm), AdamW — torch.optim.AdamW's update exactly:
 *   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * with g = grad * min(1, max_norm/(||grad||+1e-6)) / grad_scale; the whole update is skipped when a gradient is
 * inf/NaN (what scaler.step() does).  A tensor with sn_sigma != NULL carries the gradient w.r.t. its NORMALISED
 * weight (backward deferred the correction): dW_orig = (G - <G, W_orig/sigma> u v^T)/sigma is applied on the fly
 * and is what enters the norm.
 *
 * tensors_dev: table in device memory.  Work items are chunks of calm_optim_chunk_elems() consecutive elements of one
 * tensor: chunk_tensor_dev[k] = tensor index of chunk k, tensor.chunk0 = its first chunk.
 * scratch: 6*n_tensors + 4 + 3*n_chunks floats.  stats_out[2] (device): total gradient norm, found_inf.
 * Every reduction of the call has a fixed order (per-chunk partials, summed per tensor in chunk order): given equal
 * gradients — what the all-reduce leaves on every rank — all ranks apply bit-identical updates.
 * step_dev (nullable, device int32, ABI v4): the bias-correction step count kept on the device; it advances by one
 * per call EXCEPT when the update is skipped for inf/NaN gradients (torch's fused AdamW under a GradScaler), and then
 * replaces hparams->step.
 * ------------------------------------------------------------------------------------- */
typedef struct calm_optim_tensor {
    float*       param;
    const float* grad;
    float*       exp_avg;
    float*       exp_avg_sq;
    const float* sn_u;        /* [rows]  NULL for plain tensors */
    const float* sn_v;        /* [cols] */
    const float* sn_sigma;    /* [1] */
    int64_t numel;
    int32_t rows, cols;       /* rows*cols == numel for deferred spectral-norm weights */
    int32_t chunk0, reserved;
} calm_optim_tensor;

typedef struct calm_optim_hparams {
    float lr, beta1, beta2, eps, weight_decay, max_norm;   /* max_norm <= 0: no clipping */
    int32_t step;                                           /* 1-based */
} calm_optim_hparams;

int32_t calm_optim_chunk_elems(void);
int calm_optim_step(const calm_optim_tensor* tensors_dev, int32_t n_tensors, const int32_t* chunk_tensor_dev,
                    int32_t n_chunks, float* scratch, const calm_optim_hparams* hparams,
                    const float* grad_scale /* device scalar or NULL */, float* stats_out, int32_t* step_dev,
                    const float* lr_dev /* ABI v7: device scalar overriding hparams->lr, or NULL — lets a captured
                                           (hipGraph) step follow an LR schedule without re-capture */,
                    void* stream);

/* ---------------------------------------------------------------------------------------
 * Device-side batch collate feeding the path: uint8 images [B,3,H,W] -> normalised fp32 batch with
 * per-sample horizontal flip (flip[b] != 0; NULL = none) and the batch-level CutMix / MixUp of
 * distributed_trainer_cls.py:58-61 (torchvision.transforms.v2 semantics: partner = the batch rolled by one):
 *   mode 0 none | 1 MixUp: lam*x + (1-lam)*x_rolled | 2 CutMix: box = {y1,y2,x1,x2} (host ints) taken from x_rolled
 * mean/std: host float[3] (Normalize after ToDtype(scale=True), cls:137-139).  The random draws (lam ~ Beta, box,
 * flips) and the soft labels lam*onehot(y) + (1-lam)*onehot(y_rolled) are the caller's.
 * ------------------------------------------------------------------------------------- */
int calm_collate_mix(const uint8_t* img_u8, const uint8_t* flip, float* out, int32_t B, int32_t H, int32_t W,
                     int32_t mode, float lam, const int32_t* box, const float* mean, const float* std, void* stream);
/* ABI v5 — the same pass with the per-sample RandomCrop((H, W)) of distributed_trainer_cls.py:130 folded in and, optionally,
 * the output written directly as the row tokens the first Block consumes (Vi_Tools_CNN_less_V2.py:389-391):
 *   img_u8 [B,3,Hs,Ws] (the loader's Resize((256,256)) output, cls:129), crop_yx: DEVICE int32 [B,2] top-left corners
 *   (y0 in [0, Hs-H], x0 in [0, Ws-W]; NULL needs Hs == H, Ws == W); flips, mix and box act on the cropped window (the
 *   order of the reference's transform list: crop, ..., flip, ..., Normalize, then the batch-level CutMix / MixUp);
 *   out_tokens == 0: out [B,3,H,W];  != 0: out [B,H,3W] with out[b,i,3j+c] = image[b,c,i,j] (needs H == W for the model). */
int calm_collate_crop_mix(const uint8_t* img_u8, int32_t Hs, int32_t Ws, const int32_t* crop_yx, const uint8_t* flip,
                          float* out, int32_t B, int32_t H, int32_t W, int32_t out_tokens, int32_t mode, float lam,
                          const int32_t* box, const float* mean, const float* std, void* stream);

/* ---------------------------------------------------------------------------------------
 * Tokenisation (bit-exact index work).
 * image_to_rows : rows[b,i,3j+c] = img[b,c,i,j]           (Vi_Tools:389-391); rows_to_image inverse.
 * grid_transpose: out[b,j,3i+c]  = in[b,i,3j+c]           (Vi_Tools:394-395,397-398; self-inverse)
 * ------------------------------------------------------------------------------------- */
int calm_image_to_rows(const float* img, float* rows, int32_t B, int32_t S, void* stream);
int calm_rows_to_image(const float* rows, float* img, int32_t B, int32_t S, void* stream);
int calm_grid_transpose(const float* in, float* out, int32_t B, int32_t S, void* stream);

/* ---------------------------------------------------------------------------------------
 * Depthwise 3x3 conv (padding 1, zeros) + bias on a channels-last [B,S,S,C] grid
 * (middle layer of Block.proj, Vi_Tools:382; CALM_ViT_V2.py:64).  w: [C,3,3], bias: [C].
 * act: CALM_ACT_NONE/GELU; y_pre (optional) receives the pre-activation.
 * bwd: dx from dz (gradient wrt the pre-activation); dw[C,9], db[C] accumulated (caller zeroes; fp32 atomics —
 * a standalone helper, the path runs the convolutions fused in calm_cnn_residual_*).
 * ------------------------------------------------------------------------------------- */
int calm_dwconv3x3_fwd(const float* x, const float* w, const float* inv_scale, const float* bias,
                       float* y, float* y_pre, int32_t act, int32_t B, int32_t S, int32_t C, void* stream);
int calm_dwconv3x3_bwd(const float* dz, const float* x, const float* w, const float* inv_scale,
                       float* dx, float* dw, float* db, int32_t B, int32_t S, int32_t C, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused CNN residual of Block.proj / ViT.proj (Vi_Tools:378-385,400-403; CALM_ViT_V2.py:60-67,80-83):
 *   out = x + conv1x1_{hidden->3}(gelu(dwconv3x3(gelu(conv1x1_{3->hidden}(x)))))   on [B,S,S,3] tokens.
 * w0:[hidden,3] w2:[hidden,9] w4:[3,hidden] are the *_orig weights, s0/s2/s4 their device sigmas.
 * The hidden maps stay in LDS per 16x16 tile; backward recomputes them.  bwd writes dx and ADDS (fixed order through
 * `partials`) the gradients wrt the effective weights W/sigma (g0,g2,g4) and biases onto the caller's tensors.
 * hidden must be 32.  residual (ABI v7): 1 = the form above (Block.forward / ViT.forward, Vi_Tools:400-403); 0 = the bare
 * `proj(x)` a caller of the reference may invoke on its own (out = conv(...), dx without the skip term).
 * ------------------------------------------------------------------------------------- */
int calm_cnn_residual_fwd(const float* x, const float* w0, const float* s0, const float* b0, const float* w2,
                          const float* s2, const float* b2, const float* w4, const float* s4, const float* b4,
                          float* out, int32_t B, int32_t S, int32_t hidden, int32_t residual, void* stream);
int calm_cnn_residual_bwd(const float* dy, const float* x, const float* w0, const float* s0, const float* b0,
                          const float* w2, const float* s2, const float* b2, const float* w4, const float* s4,
                          const float* b4, float* dx, float* g0, float* gb0, float* g2, float* gb2, float* g4,
                          float* gb4, int32_t B, int32_t S, int32_t hidden, int32_t residual, float* partials,
                          void* stream);

/* ---------------------------------------------------------------------------------------
 * Small streaming helpers.
 * add          : out = a + b                         (residual / U-net skips, Vi_Tools:309,315,403,513-522)
 * gelu_fwd     : y = gelu_erf(x)                     (ABI v7: the GELU module called on its own; fused everywhere else)
 * gelu_bwd     : dz = dy * gelu'(z)                  (where it is not fused into a GEMM epilogue)
 * colsum       : out[n] += sum_m x[m,n]              (bias gradients; fixed order through `partials`)
 * row_scale    : out[r,c] = x[r,c] * s[r]            (ls-scaled weight for the dgrad of out_proj/mlp.3; fp32 or bf16 out)
 * mean_seq     : y[b,d] = mean_s x[b,s,d]            (AdaptiveAvgPool1d, CALM_ViT_V2.py:74-75) and bwd
 * ------------------------------------------------------------------------------------- */
int calm_add(const float* a, const float* b, float* out, int64_t n, void* stream);
int calm_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int calm_gelu_bwd(const float* dy, const float* z, float* dz, int64_t n, void* stream);
int calm_colsum(const void* x, float* out, int64_t rows, int32_t cols, int32_t x_type /* CALM_ST_* */,
                float* partials /* ABI v7 */, void* stream);
int calm_row_scale(const float* x, const float* s, void* out, int32_t rows, int32_t cols, int32_t out_type /* CALM_ST_* */,
                   void* stream);
int calm_mean_seq_fwd(const float* x, float* y, int32_t B, int32_t S, int32_t D, void* stream);
int calm_mean_seq_bwd(const float* dy, float* dx, int32_t B, int32_t S, int32_t D, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CALM_VIT_H */
