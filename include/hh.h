/* libhh -- C ABI of the MI355X-native (gfx950) hot path of helping_hand_for_egocentric_videos.
 *
 * The reference has no FFI layer: its "operator API" for this path is PyTorch nn.Module.forward
 * (SURVEY.md section 8b).  Each entry point below names the reference op group it replaces
 * (file:line into /root/reference).  The Python host (helping_hand_for_egocentric_videos_amd/ops.py)
 * binds these with ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory owned by the caller.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never allocates,
 *     never synchronises, keeps no pointer after returning.
 *   - return 0 on success, negative hh_status otherwise; hh_last_error_string() describes the
 *     last failure of the calling thread.
 *   - bf16 = 16-bit brain float (uint16 storage); "rows" are token rows, row-major, last dim contiguous.
 */
#ifndef HH_H
#define HH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* hh_stream_t;

enum hh_status { HH_OK = 0, HH_ERR_SHAPE = -1, HH_ERR_DTYPE = -2, HH_ERR_UNSUPPORTED = -3, HH_ERR_LAUNCH = -4,
                 HH_ERR_ALIGN = -5 };
enum hh_dtype { HH_F32 = 0, HH_BF16 = 1 };
enum hh_act { HH_ACT_NONE = 0, HH_ACT_QUICKGELU = 1, HH_ACT_RELU = 2 };
/* layout of a packed q|k|v buffer of R = B*N token rows and `heads` heads of 64: element (row, which in {q,k,v}, head, d) sits at
 *   HH_QKV_TOKEN_MAJOR: row * 3*heads*64 + which * heads*64 + head * 64 + d        (what nn.Linear writes, LaviLa.py:249)
 *   HH_QKV_HEAD_MAJOR : ((which * heads + head) * R + row) * 64 + d                (3*heads planes of [R, 64]: a head's rows are
 *                       contiguous 128-byte lines; written by hh_gemm_bf16 with c_block_stride = R * 64) */
enum hh_qkv_layout { HH_QKV_TOKEN_MAJOR = 0, HH_QKV_HEAD_MAJOR = 1 };
/* OR-ed into the qkv_layout argument of hh_space_attn_fwd / hh_time_attn_fwd (round 5): walk the (clip, ...) problems LAST TO FIRST.  Same
 * results; a kernel that starts on the rows its predecessor wrote last finds them in the 256 MB Infinity Cache (hh_gemm_epilogue.walk_reverse
 * is the GEMM side).  Honoured by the joint-block space kernels (n = 256, n = 576) and the MFMA time kernels; the generic 16-query space kernel ignores it. */
#define HH_QKV_WALK_REVERSE 2

int hh_version(void);
/* sizeof(hh_gemm_epilogue) / sizeof(hh_qgemm_opts) as the library was compiled (name = the struct's name; -1: unknown): the binding's own
 * declaration must agree (helping_hand_for_egocentric_videos_amd/_lib.py checks it at load time) */
int hh_abi_sizeof(const char* name);
/* Performance knobs for A/B measurements (never change results).  Together with the per-stream CU budget below this is the
 * library's ONLY process-global mutable state; no entry point reads the environment.
 *   "gemm256"       0 = 128x128 kernel only, 1 = 256x256 one tile per block, 2 = + wave-row stagger, 3 = persistent 8-wave kernel,
 *                   4 = 4-wave kernel (128x128 per wave), one tile per block, 5 (default) = persistent 4-wave kernel where K >= 384 and
 *                   K % 128 == 0, else 3.  All produce bit-identical results.
 *                   256x256: one workgroup per CU walks its tiles in one continuous k-tile stream, four barriers per k-tile
 *   "gemm_tail"     1 (default) = row tails of <= 64 rows (M - 256*floor(M/256), K % 512 == 0) inside the persistent kernel where it
 *                   runs (pieces of <= 32 rows x 32 columns per workgroup), else as 2; 2 = tails of <= 64 rows always on the separate
 *                   split-K-in-workgroup kernel; 0 = on the 128x128 kernel.  Same results in all three.
 *   "gemm256_dynamic" 1 (default) = the persistent 4-wave kernel takes every tile after a workgroup's first from per-XCD atomic counters
 *                   (per launch stream, zeroed again by the launch's last workgroup), 0 = static stride
 *   "gemm256_min_tiles" fewest 256x256 tiles for which the 256x256 kernels are used (default 192; the persistent 4-wave kernel takes
 *                   its shapes from 128 tiles on); smaller GEMMs run on the 128x128 kernel
 *   "gemm256_group" m-tiles per XCD-local group of the tile walk (0 = per-shape default)
 *   "gemm256_skew"  -1 auto / 0 off / 1 on: start-time skew of the one-tile-per-block kernel (spreads the epilogue HBM bursts)
 *   "gemm256_pskew" 0..64: start skew quantum of the persistent kernel (default 0)
 *   "gemm_ln_w4"    1 (default) = LayerNorm-fold epilogues (ln_stats / z_out below) on the persistent 4-wave kernel where it takes the shape,
 *                   0 = on the 128x128 kernel (A/B, debugging);  "gemm_ln_pskew" 0..256 / "gemm_ln_phases" 0..32: start skew of the fold's
 *                   producer launches in s_sleep(8) steps x phases (default 0: measured +-0 in the step)
 *   "gemm_tile224"  0 (default) / 1: 224-row tiles on the persistent 4-wave kernel (bf16 bias-only and fold-producer epilogues) where they
 *                   turn a partial last round into whole rounds (hh_gemm256_tile_rows); bit-identical results
 *   "space_joint"   1 (default) = space attention on the joint-block kernel where n / 16 divides by 4 waves x {4, 3, 2} blocks,
 *                   0 = always the 16-query-block kernel;  "space_waves" waves per workgroup of the joint kernel: 0 (default) = automatic (12 waves x
 *                   3 blocks when K / V fill the LDS, i.e. one workgroup per CU, and n / 16 divides by 36 -- config 4's n = 576 --, else 4), 4 / 12 =
 *                   force where the shape divides;  "space_prog" 1 (default) = K / V staged progressively (compute starts on the first key
 *                   segment while the rest is in flight) where a specialised kernel exists: n = 576; 0 = off; 2 = also n = 256 (experiment: within
 *                   +-0.1 % at step level).  (Round 4's persistent cross-problem-prefetch variant, value 3, was slower and is removed: DESIGN.md 4.2);
 *                   "space_debug" 0 / 1 / 2: full kernel / memory traffic only / no staging
 *   "space_mfma32"  1 (default) = space attention with n <= 256, n % 64 == 0 on the 32x32x16-MFMA kernel whose exponentials are software-pipelined
 *                   under the matrix core inside each wave (round 6), one problem per workgroup; 2 = n = 256 on the persistent wave-specialised
 *                   form (4 compute + 4 loader waves per CU: fastest in the step, but it holds every CU for its whole duration and stretches
 *                   the decoder stream) and n = 576 on the third-step pipelined 16x16x32 kernel (equal in time to the default progressive
 *                   one); 0 = the 16x16x32 kernels of rounds 2-5
 *   "mattn_no_ticket" 0 (default) / 1: tests only -- hh_mattn_fwd / _bwd act as if no ticket row were free (more than 32 launch streams seen):
 *                   they then run ONE key slice per (clip, head group) instead of failing; same results up to fp32 re-association
 *   "gemm256_debug_ts", "gemm256_debug_nostore": diagnostics (timeline recording; skip the epilogue stores) */
int hh_set_tuning(const char* name, int value);
const char* hh_last_error_string(void);
/* Debug only: after hh_set_tuning("gemm256_debug_ts", 1), every persistent 256x256 GEMM launch records, for the first 8 tiles of
 * each workgroup, the 100 MHz timestamps {tile start, k-tile 0 landed, main loop done, next prologue issued, stores issued};
 * this copies them to host memory: out[blocks][8][7] (entries 5, 6: s_memtime at stamps 1, 2 -> shader clock during the
 * main loop). */
int hh_debug_gemm_timeline(unsigned long long* out, int blocks);

/* Per-stream CU budget for the software-pipelined step (the reference's step, run/train.py:103-203, runs the frozen towers
 * and the decoder back to back on one stream; here they overlap on two streams).  One-workgroup-per-CU persistent kernels
 * (the 256x256 GEMM) launched on `stream` use n_cus workgroups instead of one per CU of the device, which leaves the other CUs
 * to kernels of concurrent streams.  n_cus % 8 == 0 (one XCD-balanced slice); 0 restores the default (all CUs).  Never
 * changes results. */
int hh_stream_set_cu_budget(hh_stream_t stream, int n_cus);
int hh_stream_get_cu_budget(hh_stream_t stream, int* out);

/* ---- built-in per-kernel timing (bench.py `roofline`).  hh_prof_enable(stride > 0): from now on every stride-th launch of each
 * instrumented kernel class is bracketed by two events recorded on its launch stream with ONLY that kernel between them (stride 0
 * = off, the default; every call clears the records).  hh_prof_read sums them for one class: launches timed / seen, elapsed ms,
 * and the launches' algorithmic work (flops for the GEMM classes, bytes for the others; formulas in DESIGN.md section 5).  It
 * synchronises on the recorded events. */
enum hh_prof_class {
    HH_PROF_GEMM256 = 0,      /* persistent 256x256 GEMM kernels (gemm256w4p_kernel, gemm256d_kernel): 2*M*N*K of the full-tile rows */
    HH_PROF_GEMM_OTHER = 1,   /* row tails, 128x128 kernel, one-tile-per-block 256x256 kernel: 2*M*N*K of their rows */
    HH_PROF_SPACE_ATTN = 2,   /* space_attnj_kernel (joint blocks; n = 256) / space_attnp_kernel (progressive staging; n = 576) / space_attn16_kernel:
                                 8*B*N*D bytes (q,k,v read + o written, bf16) */
    HH_PROF_TIME_ATTN = 3,    /* time_attn_mfma*_kernel: 8*B*N*D bytes */
    HH_PROF_ADD_LN = 4,       /* fused residual add + LayerNorm: bytes read + written */
    HH_PROF_GEMM_TN = 5,      /* weight-gradient GEMM: 2*M*N*K */
    HH_PROF_XATTN_FWD = 6,    /* decoder cross-attention forward: K,V bytes read */
    HH_PROF_XATTN_BWD = 7,    /* decoder cross-attention backward: K,V read + dK,dV written */
    HH_PROF_CLASSES = 8
};
/* number of launching entry-point calls this process has made so far (every hh_* call that enqueued at least one kernel; monotonic).
 * bench.py divides the host time of issuing one step by the difference over that step: host microseconds per library call. */
int64_t hh_call_count(void);
int hh_prof_enable(int stride);
int hh_prof_read(int klass, int64_t* launches_timed, int64_t* launches_seen, double* total_ms, double* total_work);
/* roles (round 5): the host names the part of the step it is launching -- 0 = decoder / losses / optimizer (default), 1 = vision tower,
 * 2 = text tower -- and every timed launch carries the role current at its launch; the stride applies per (class, role).  The three
 * parts run on three concurrent streams in the pipelined step: only the sums of ONE role are comparable with a step's wall time.
 * hh_prof_read_role(klass, role, ...) sums one role's records (role -1 = all roles = hh_prof_read). */
int hh_prof_set_role(int role);
int hh_prof_read_role(int klass, int role, int64_t* launches_timed, int64_t* launches_seen, double* total_ms, double* total_work);
/* the kernel (template instantiation, spelled as rocprofv3 --kernel-trace prints it) that the LAST launch of the class dispatched since
 * hh_prof_enable(stride > 0); "" if none.  A static string owned by the library. */
const char* hh_prof_kernel_name(int klass);

/* ---- caller-owned workspaces.  No entry point allocates; these return the size in BYTES of the scratch buffer an entry point
 * takes (negative = bad arguments):
 *   gemm_splitk      : C of hh_gemm_bf16 with epi->splitk = S  (fp32 [S, M, N], split_stride = M*N)
 *   gemm_tn          : `partials` of hh_gemm_tn_bf16           (fp32 [splits, M, N])
 *   gemm_zstats      : epi->z_partials of hh_gemm_bf16             (fp32 [M, N / 128, 2] per-slice row sums)
 *   xattn_bwd        : `dq` of hh_xattn_bwd                    (fp32 [dq_splits, B, Q, heads*64])
 *   attn_cls_partial : `cls_partial` of hh_space_attn_fwd (time_mode 0) / hh_time_attn_fwd (1) and input of hh_cls_combine
 *                      (fp32 [B, heads, G, 68]) */
int64_t hh_workspace_bytes_gemm_splitk(int64_t M, int N, int splitk);
int64_t hh_workspace_bytes_gemm_zstats(int64_t M, int N);      /* `z_partials` of hh_gemm_bf16's producer-side LayerNorm fold */
int64_t hh_workspace_bytes_gemm_tn(int M, int N, int splits);
int64_t hh_workspace_bytes_xattn_bwd(int B, int Q, int heads, int dq_splits);
int64_t hh_workspace_bytes_xattn_fwd(int B, int Q, int heads, int splits);
/*   mattn_fwd        : `workspace` of hh_mattn_fwd, slices > 1 (fp32 [slices, B*Q, 8, 512 + 4] partial pooled rows + statistics)
 *   mattn_bwd        : `workspace` of hh_mattn_bwd, slices > 1  (fp32 [slices, B*Q, 8 * 512]) */
int64_t hh_workspace_bytes_mattn_fwd(int B, int Q, int slices);
int64_t hh_workspace_bytes_mattn_bwd(int B, int Q, int slices);
int64_t hh_workspace_bytes_attn_cls_partial(int B, int T, int n, int heads, int time_mode);

/* ---- LayerNorm over the last dim (model/LaviLa.py:439,456 eps 1e-6 / 1e-5; tfm_decoder.py:57,375-377)
 * y[r,:] = (x[r,:]-mean)/sqrt(var+eps)*gamma+beta ; x dtype / y dtype in {HH_F32, HH_BF16}; gamma/beta fp32.
 * cols % 8 == 0 and cols <= 2048.
 * If mean_out/rstd_out are non-NULL the per-row statistics (fp32) are stored for the backward. */
int hh_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                     float* mean_out, float* rstd_out, int64_t rows, int cols, float eps, hh_stream_t stream);

/* Fused residual add + LayerNorm: x (fp32 [rows, cols]) = (x + delta) + delta2 (bf16; delta2 may be NULL) -- written back iff
 * write_x -- and y = LN(x) (bf16 or fp32).  Used for x + attn / x + mlp of SpaceTimeBlock (model/LaviLa.py:364,384,388); two
 * deltas let the x += space-branch update ride on the next block's x += mlp-branch pass. */
int hh_add_layernorm_fwd(float* x, const void* delta, const void* delta2, int write_x, const float* gamma, const float* beta, void* y,
                         int y_dtype, int64_t rows, int cols, float eps, hh_stream_t stream);

/* LayerNorm backward: dx fp32 [rows, cols]; dgamma / dbeta fp32 [cols] are ACCUMULATED with atomics (zero them first).
 * x is the forward input (fp32 or bf16), mean / rstd the statistics saved by hh_layernorm_fwd.  cols <= 1024. */
int hh_layernorm_bwd(const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd,
                     const float* dy, float* dx, float* dgamma, float* dbeta, int64_t rows, int cols,
                     hh_stream_t stream);

/* ---- bf16 MFMA GEMM with fused epilogue (replaces nn.Linear: LaviLa.py:249,281,186-189; tfm_decoder.py:156,
 * nn.MultiheadAttention in-proj :438-441)
 *   C[m,n] = epilogue( sum_k A[m,k] * W[n,k] )        A [M,K] bf16 (lda), W [N,K] bf16 (ldw)
 *   epilogue: (+ bias[n]) -> (* colscale for n < colscale_cols) -> act -> (+ resid[m,n] fp32, ldr) -> store
 *   out dtype bf16 or fp32, ldc in elements.  Output row remap for token-major scatter:
 *   out_row = m + (m / remap_group) * remap_skip + remap_offset   (remap_group 0 = identity).
 *   Requirements: K % 64 == 0, N % 128 == 0, 16-byte aligned pointers/strides.  M arbitrary. */
typedef struct hh_gemm_epilogue {
    const float* bias;        /* [N] or NULL */
    const float* resid;       /* fp32 [M(out rows), ldr] or NULL; may alias C when c_dtype == HH_F32 */
    int64_t ldr;
    float colscale;           /* applied to columns < colscale_cols (q *= d^-0.5, LaviLa.py:252) */
    int colscale_cols;
    int act;                  /* hh_act */
    int c_dtype;              /* hh_dtype */
    int64_t remap_group, remap_skip, remap_offset;
    int splitk;               /* <=1: off.  S>1: split s handles k-tiles [s*ceil(nk/S), ...) and writes its partial
                                 (same epilogue, caller passes no bias/resid) at C + s*split_stride; caller sums */
    int64_t split_stride;     /* elements between partial slabs */
    int64_t c_block_stride;   /* 0: C is row-major [M, ldc].  S > 0: C is column-blocked -- N / 64 planes of [M, 64], plane j at
                                 C + j * S (S >= 64 * M): element (m, n) at (n / 64) * S + m * 64 + n % 64; ldc is ignored.  With
                                 N = 3 * heads * 64 this is the HH_QKV_HEAD_MAJOR layout of the attention kernels */
    /* ---- LayerNorm folded into the GEMMs on either side of it (model/LaviLa.py:372-388: norm1 / norm2 sit between the attention
     * output projection and the next qkv / fc1 Linear).  Algebra:  LN(z) W^T + b = rstd * (z (gamma o W)^T) - rstd*mean * colsum(gamma o W) + (beta W^T + b).
     * PRODUCER side (z_out != NULL; the projection GEMM): with v = acc + bias (fp32, before the bf16 rounding of C),
     *     z[m,n] = z_resid[m,n] + v[m,n]   ->  z_out (bf16);   z_stats[m] = (rstd, -rstd * mean) of row m of z over all N columns;
     *   C is still written (the branch output the residual stream adds later) unless skip_c; with z_update the fp32 z also replaces z_resid.  No act / colscale / resid / remap / split-K
     *   / column-blocked C with it.  z_partials: workspace of hh_workspace_bytes_gemm_zstats(M, N) bytes.
     * CONSUMER side (ln_stats != NULL; the qkv / fc1 GEMM): A holds the rows z, W holds bf16(gamma o W), bias holds beta W^T + b, and
     *     acc <- ln_stats[m][0] * acc + ln_stats[m][1] * ln_colsum[n] + bias[n]   before colscale / act (bias must be non-NULL).
     *   No resid / remap / split-K with it. */
    const float* ln_stats;    /* fp32 [M, 2] = (rstd, -rstd * mean) per row of A, or NULL */
    const float* ln_colsum;   /* fp32 [N]: sum_k float(W[n, k]) of the bf16 operand actually multiplied */
    const void* z_resid;      /* fp32 [M, z_ldr] (bf16 with z_resid_dtype = HH_BF16) */
    int64_t z_ldr;
    void* z_out;              /* bf16 [M, z_ldc], or NULL (producer side off) */
    int64_t z_ldc;
    float* z_stats;           /* fp32 [M, 2] out */
    float* z_partials;        /* workspace */
    float z_eps;              /* LayerNorm eps of the statistics */
    int skip_c;               /* != 0 with z_out: C is not written (a branch nobody adds to the residual stream: the time branch, LaviLa.py:372-384) */
    int z_update;             /* != 0 with z_out: z (fp32, before its bf16 rounding) is also written back to z_resid IN PLACE -- the residual
                                 stream update x <- x + branch (LaviLa.py:384,388) happens in the producing GEMM's epilogue */
    int z_resid_dtype;        /* HH_F32 (0, default) or HH_BF16 (round 5): z_resid holds bf16 rows -- the producer that only feeds a LayerNorm
                                 (the time branch: z1 = x + t goes to norm1 alone, LaviLa.py:372) reads the 2-byte z = bf16(x) its own block's
                                 norm3 consumed instead of the 4-byte fp32 stream; no z_update with it */
    void* z_resid_lo;         /* != NULL (round 5, with z_out == z_resid, z_resid_dtype = HH_BF16, z_update != 0, z_ldr == z_ldc): the residual stream is
                                 kept as a PAIR of bf16 rows x = hi + lo (z_resid = hi, which IS the LayerNorm input z of the next GEMM; z_resid_lo
                                 = lo): the epilogue reads both (4 B), adds acc + bias in fp32, and writes hi' = bf16(x'), lo' = bf16(x' - hi')
                                 back IN PLACE (4 B) -- 8 bytes per element instead of the 10 of fp32 x + separate z, ~17 significant bits */
    int walk_reverse;         /* != 0 (round 5): the persistent kernel walks its m-tiles last to first (same tiles, same arithmetic, same results):
                                 the tower alternates the direction from kernel to kernel so that each one starts on the rows its predecessor
                                 wrote last -- the part of its input still in the Infinity Cache (model/LaviLa.py: SpaceTimeBlock.fused) */
} hh_gemm_epilogue;

int hh_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                 int64_t M, int N, int K, const hh_gemm_epilogue* epi, hh_stream_t stream);

/* Row statistics for the consumer side of the LayerNorm fold when z does not come out of hh_gemm_bf16 (tests, other producers):
 * stats[m] = (rstd, -rstd * mean) of row m of z (bf16 [rows, ldz], cols <= 2048, cols % 8 == 0), two-pass in fp32. */
int hh_ln_rowstats(const void* z, int64_t ldz, float* stats, int64_t rows, int cols, float eps, hh_stream_t stream);

/* ---- casts / transposes (host-side plumbing for weights and wgrad operands) */
int hh_cast_f32_to_bf16(const float* x, void* y, int64_t n, hh_stream_t stream);
int hh_cast_bf16_to_f32(const void* x, float* y, int64_t n, hh_stream_t stream);
/* y[c, r] = x[r, c]  (bf16, x [rows, cols] with ldx) -> y [cols, rows] with ldy; in_dtype F32 converts on the fly */
int hh_transpose_to_bf16(const void* x, int x_dtype, int64_t ldx, void* y, int64_t ldy, int64_t rows, int64_t cols,
                         hh_stream_t stream);

/* ---- TimeSformer patch embedding front end (model/LaviLa.py:218-223,540-559)
 * im2col: video [B*T,3,H,W] fp32 -> patches bf16 [B*T*n, Kpad] (k = c*P*P + i*P + j, zero padded to Kpad) */
int hh_patch_im2col(const float* video, void* patches, int64_t frames, int H, int W, int P, int Kpad,
                    hh_stream_t stream);
/* same, straight from decoded uint8 frames with ToTensor + Normalize fused in ((u/255 - mean[c]) / std[c]; mean3/std3 are
 * HOST pointers to 3 floats; data_loader/transforms.py:38-75, run/train.py:442-445); channels_last: [F,H,W,3] else [F,3,H,W] */
int hh_patch_im2col_u8(const uint8_t* video, void* patches, int64_t frames, int H, int W, int P, int Kpad,
                       int channels_last, const float* mean3, const float* std3, hh_stream_t stream);
/* x[b,0,:]   = LN(cls + pos[0]) ; x[b,1+f*n+p,:] = LN(tok[(b*T+f)*n+p,:] + pos[1+p] + temporal[f])  (eps, ln_pre)
 * tok fp32 [B*T*n, D]; x fp32 [B, 1+T*n, D].  z_out / z_stats (optional, both or neither): z = bf16(x) [rows, D] and its row statistics
 * (rstd, -rstd * mean) fp32 [rows, 2] with eps z_eps -- the operands of the first block's folded LayerNorm (hh_gemm_epilogue.ln_stats). */
int hh_embed_ln_pre(const float* tok, const float* cls, const float* pos, const float* temporal,
                    const float* gamma, const float* beta, float* x, int B, int T, int n, int D, float eps,
                    void* z_out, float* z_stats, float z_eps,
                    void* z_lo /* NULL, or bf16 [rows, D]: x - z_out rounded to bf16 (the pair stream x = z_out + z_lo; x itself may then be NULL) */,
                    hh_stream_t stream);

/* ---- weight-gradient GEMM in its natural layout (backward of the nn.Linear layers of tfm_decoder.py:156,438-441):
 * partials[s, m, n] (fp32, [splits, M, N]) = sum over the s-th token slice of At[k, m] * Bt[k, n]; At bf16 [K, M] (row stride lda),
 * Bt bf16 [K, N] (row stride ldb), i.e. dY and X exactly as they sit in memory (token-major) -- no transposed copies.
 * M % 128 == 0, N % 128 == 0, any K; the caller sums the `splits` partials.  colsum_partials (optional, fp32 [splits, M]):
 * sum over the s-th token slice of At[k, m] -- the bias gradient of the nn.Linear whose dY is At -- as a by-product. */
int hh_gemm_tn_bf16(const void* At, int64_t lda, const void* Bt, int64_t ldb, float* partials, float* colsum_partials, int M, int N,
                    int64_t K, int splits, hh_stream_t stream);
/* out[i] (fp32, n values) = sum over s of partials[s * n + i], planes added in order (deterministic): the sum of hh_gemm_tn_bf16's split-K
 * planes / column-sum partials (the reference's autograd does this inside its one cuBLAS call per nn.Linear weight gradient).  n % 4 == 0. */
int hh_sum_partials(const float* partials, float* out, int splits, int64_t n, hh_stream_t stream);
/* batched two-pair form:  C_z [M, N] fp32 (dense, batch stride stride_c) = At_z^T Bt_z + At2_z^T Bt2_z  for z = 0 .. batch - 1; operand z
 * of a pair sits at + z * stride_a / + z * stride_b elements; both pairs share lda / ldb / K; At2 = Bt2 = NULL: one pair.  No split-K (K
 * is short here: the rows of hh_mattn_bwd's Pd^T / dS^T), K % 64 == 0 with two pairs.  The memory-side gradient of the decoder's
 * cross-attention over all six layers (tfm_decoder.py:433-441 backward) in one launch. */
int hh_gemm_tn_bf16_batched2(const void* At, const void* Bt, const void* At2, const void* Bt2, int64_t lda, int64_t ldb, int64_t stride_a,
                             int64_t stride_b, float* C, int64_t stride_c, int M, int N, int64_t K, int batch, hh_stream_t stream);

/* Debug / tests: number of query blocks the space-attention kernels have redone on their running-maximum path since the last reset (a 16-query
 * block of the 16x16x32 kernels: a score more than 2^127 above its reference maximum; a 32-query block of the 32x32x16 kernels: a row sum outside
 * [2^-100, 2^100]).  Synchronises the device; reset != 0 zeroes the counters.  -1 on error. */
int64_t hh_debug_space_redo_count(int reset);

/* ---- divided space-time attention cores (model/LaviLa.py:246-283, attn() :194-198)
 * qkv bf16 [B, N=1+T*n, 3*D] (q|k|v, head-major inside D) or its head-major planes (qkv_layout: enum hh_qkv_layout), out bf16
 * [B, N, D] token-major in both cases; head dim 64.  The q columns are PRE-SCALED by the
 * QKV GEMM epilogue: by d^-1/2 for hh_time_attn_fwd (LaviLa.py:252), by d^-1/2 * log2(e) for hh_space_attn_fwd -- its scores are
 * base-2 logits, so that a probability costs one v_exp_f32 (hh_cls_attn_fwd: q_log2 = 1 for such a buffer, 0 otherwise).
 * space: per (b, head, frame): n queries x (CLS + n) keys.  time: per (b, head, patch): T queries x (CLS + T) keys.
 * cls: the CLS query attends all N keys (row 0 of out).  Rows 1.. are written by space/time, row 0 by cls.
 * cls_partial (optional, fp32 [B, heads, G, 68], G = T for space, ceil(n / (128/T)) for time): when non-NULL the kernel
 * also emits the CLS query's partial softmax statistics over its own key group (record = m, l, 0, 0, o[64]);
 * hh_cls_combine merges the G records into out row 0, which replaces the separate hh_cls_attn_fwd pass. */
int hh_space_attn_fwd(const void* qkv, int qkv_layout, void* out, float* cls_partial, int B, int T, int n, int heads,
                      hh_stream_t stream);
int hh_time_attn_fwd(const void* qkv, int qkv_layout, void* out, float* cls_partial, int B, int T, int n, int heads,
                     hh_stream_t stream);
int hh_cls_attn_fwd(const void* qkv, int qkv_layout, void* out, int B, int N, int heads, int q_log2, hh_stream_t stream);
int hh_cls_combine(const float* partial, int G, void* out, int B, int N, int heads, hh_stream_t stream);

/* ---- causal self-attention of the CLIP text tower (model/openai_model.py:182-232; mask model/LaviLa.py:636-642)
 * qkv bf16 [S, L, 3*W] (q|k|v, head-major inside W = heads*64, q pre-scaled by 64^-0.5), out bf16 [S, L, W]; L <= 80. */
int hh_text_attn_fwd(const void* qkv, void* out, int S, int L, int heads, hh_stream_t stream);

/* ---- decoder cross-attention core (nn.MultiheadAttention inside tfm_decoder.py:438-441; 13 x 4096, 8 heads)
 * q fp32 [B, Q, C] (already scaled by d^-0.5), k/v bf16 [B, M, ldkv] (head-major columns, C = heads*64 used),
 * out fp32 [B, Q, C], lse fp32 [B, heads, Q] (log-sum-exp for the backward).  Q <= 16. */
/* dropout_p > 0 applies inverted dropout to the attention probabilities (nn.MultiheadAttention(dropout=0.1),
 * tfm_decoder.py:365) with a counter-based mask keyed by (seed, clip, head, query, key); the backward regenerates it. */
int hh_xattn_fwd(const float* q, const void* k, const void* v, int64_t ldkv, float* out, float* lse,
                 int B, int Q, int M, int heads, float dropout_p, uint32_t seed, hh_stream_t stream);
/* the same forward with the keys cut into `splits` slices, one workgroup per (clip, head, slice), folded by a merge kernel: for
 * small B*heads (long clips at small batch) where hh_xattn_fwd's B*heads workgroups leave most of the CUs idle.  workspace:
 * hh_workspace_bytes_xattn_fwd(B, Q, heads, splits) bytes.  Same dropout mask as hh_xattn_fwd for the same seed. */
int hh_xattn_fwd_split(const float* q, const void* k, const void* v, int64_t ldkv, float* out, float* lse, float* workspace, int splits,
                       int B, int Q, int M, int heads, float dropout_p, uint32_t seed, hh_stream_t stream);
/* backward: the keys are cut into dq_splits slices (one workgroup each per (clip, head)); dq fp32 [dq_splits, B, Q, C] holds the
 * slices' partial dq (the caller sums them); dk/dv bf16 [B, M, lddkv] */
int hh_xattn_bwd(const float* q, const void* k, const void* v, int64_t ldkv, const float* out, const float* lse,
                 const float* dout, float* dq, int dq_splits, void* dk, void* dv, int64_t lddkv,
                 int B, int Q, int M, int heads, float dropout_p, uint32_t seed, hh_stream_t stream);

/* ---- decoder cross-attention WITHOUT the memory-side K/V projections (round 5; tfm_decoder.py:433-441; csrc/mattn.hip).
 * q_h . ((memory + pos) Wk_h^T + bk_h) = (q_h Wk_h) . (memory + pos) + const  and  sum_i p_i (memory_i Wv_h^T + bv_h) = (sum_i p_i memory_i) Wv_h^T
 * + (sum_i p_i) bv_h:  the caller maps the query rows of every head into memory space (qt = q_h Wk_h: a head-batched hh_qgemm_f32x3),
 * these kernels attend over the UN-projected rows, and the caller applies Wv_h / bv_h to the pooled rows.  d_model 512, 8 heads (the
 * reference's decoder, tfm_decoder.py:51); Q <= 16; M % 32 == 0.
 *   qt fp32 [B*Q, 8*512]: row (clip, query), columns head*512 + k (already scaled by d^-1/2);  mp = memory + pos, mem = memory: bf16 [B, M, ld]
 *   forward:  pooled fp32 [B*Q, 8*512] = sum_i Pd[., i] mem[i, :],  Pd = dropout(softmax_i(qt . mp[i]));  lse2 fp32 [B*Q, 8] = log2 sum_i
 *             2^(score_i log2 e) (for the backward);  rsum fp32 [B*Q, 8] = sum_i Pd (1 without dropout: the weight of the value bias).
 *             The keys are cut into `slices` slices (one workgroup of 8 waves per (clip, 4 heads, slice)); with slices > 1 the partial rows
 *             go through `workspace` (hh_workspace_bytes_mattn_fwd) and the LAST workgroup of a (clip, 4 heads) to finish folds them in the
 *             same launch (agent-scope release / ticket / acquire; fixed summation order).  hh_mattn_slices(M, wanted) = the slice count used.
 *   backward: dqt fp32 [B*Q, 8*512] = sum_i dS[., i] mp[i, :] (slices folded the same way; workspace: hh_workspace_bytes_mattn_bwd) with dS = P o (dP - delta),
 *             dP = mask/(1-p) o (dpooled . mem[i] + dca . bv), delta = dca . ca per head (dca / ca fp32 [B*Q, 512]: gradient and value of
 *             the head outputs, bv fp32 [512] the layer's value bias);  pdT / dsT bf16 [B, rows_total, M]: rows [row_off, row_off + 128)
 *             (row_off + head*16 + query) receive Pd^T and dS^T, keys contiguous -- the operands of hh_gemm_tn_bf16_batched2, which makes
 *             d mem + d mp for all layers at once;  qt16 / dp16 (optional, bf16 [B, rows_total, 512]): the same rows of bf16(qt) / bf16(dpooled),
 *             the other operands of that GEMM.  Same dropout mask as the forward for the same seed.
 *   keys_valid (round 6): 0 < keys_valid <= M -- keys [keys_valid, M) are PADDING rows (finite, e.g. zeros) that no softmax sees: their
 *             probabilities, Pd^T and dS^T columns are exact zeros.  Lets a caller round any memory length up to the 32-key chunk / the 128-row
 *             tile of hh_gemm_tn_bf16_batched2 (model/tfm_decoder.py pads M = T * n to a multiple of 128).  <= 0: all M keys are valid. */
int hh_mattn_slices(int M, int slices);
int hh_mattn_fwd(const float* qt, const void* mp, const void* mem, int64_t ld, float* pooled, float* lse2, float* rsum, float* workspace,
                 int slices, int B, int Q, int M, int heads, int C, float dropout_p, uint32_t seed, int keys_valid, hh_stream_t stream);
int hh_mattn_bwd(const float* qt, const float* dpooled, const float* lse2, const float* dca, const float* ca, const float* bv,
                 const void* mp, const void* mem, int64_t ld, float* dqt, float* workspace, int slices, void* pdT, void* dsT, void* qt16, void* dp16,
                 int rows_total, int row_off, int B, int Q, int M, int heads, int C, float dropout_p, uint32_t seed, int keys_valid, hh_stream_t stream);

/* ---- query side of the decoder (model/tfm_decoder.py:430-461 forward_pre on the 13 object queries, :208-233 heads, and the
 * txt_proj / obj_proj projections of run/train.py:124-125,187-189): fp32 operands, bf16 matrix cores at fp32-grade accuracy.
 * hh_qgemm_f32x3: C = epilogue(prologue(A) . B); every operand fp32; each fp32 value is split on the fly into bf16 hi + lo and
 * a product takes three MFMAs (Ahi.Bhi + Alo.Bhi + Ahi.Blo).
 *   mode 0 (NT, forward of nn.Linear)   C[m,n] = sum_k A[m,k] B[n,k]     A [M,K] (lda), B [N,K] (ldb)
 *   mode 1 (NN, input gradient)         C[m,n] = sum_k A[m,k] B[k,n]     A [M,K],       B [K,N]
 *   mode 2 (TN, weight gradient)        C[m,n] = sum_k A[k,m] B[k,n]     A [K,M],       B [K,N]
 * C fp32 [M,N] (ldc).  Contiguous dimensions and leading dimensions must be multiples of 4; M, N, K otherwise arbitrary.
 *   prologue on A:  A *= a_scale (0 = off);  dropout mask: element at memory (row r, col c) of A kept iff
 *                   hash(a_drop_seed, r * a_drop_ld + c) >= a_drop_p * 2^32, scaled by 1/(1-a_drop_p)  [= the mask the forward
 *                   epilogue drew for element (m, n) of an output with N = a_drop_ld columns]
 *   epilogue:       + bias[n] -> * scale for n < scale_ncols (scale 0 = off, scale_ncols 0 = all) -> ReLU -> dropout(drop_p,
 *                   drop_seed, element index m * N + n) -> * (relu_mask[m,n] > 0 ? mask_scale : 0) -> + resid[m,n]
 *   colsum (mode 2 only, optional): colsum[m] = sum_k A_eff[k, m]  (bias gradient of the same nn.Linear)
 *   splitk: see the struct */
typedef struct hh_qgemm_opts {
    float a_scale;
    float a_drop_p; uint32_t a_drop_seed; int32_t a_drop_ld;
    const float* bias;
    float scale; int32_t scale_ncols;
    int32_t relu;
    float drop_p; uint32_t drop_seed;
    const float* relu_mask; int64_t ldmask; float mask_scale;
    const float* resid; int64_t ldr;
    float* colsum;
    int32_t splitk;           /* > 1: the contraction is cut into that many slices whose partial products are added ATOMICALLY into C
                                 (and colsum): the caller zeroes them first; no epilogue options.  For long contractions with few
                                 output tiles (weight gradients of the box heads: 512 x 512 outputs over 6656 rows) */
    /* round 5 -- batched form (the per-head maps of the K/V-projection-free cross-attention, hh_mattn_*): batch > 1 runs `batch`
     * independent products in one launch; product z reads A + z * stride_a, B + z * stride_b, writes C + z * stride_c, and takes
     * bias + z * stride_bias / colsum + z * stride_colsum / rowscale + z * stride_rowscale (strides in elements, multiples of 4 for the
     * matrices and the bias).  No split-K, relu_mask or residual with it.
     * rowscale (optional, batched or not): NT / NN -- the bias term becomes bias[n] * rowscale[m * ld_rowscale] (the value bias of an
     * attention whose dropped probabilities no longer sum to one); TN -- colsum[m] = sum_k A[k, m] * rowscale[k * ld_rowscale]. */
    int32_t batch;
    int64_t stride_a, stride_b, stride_c, stride_bias, stride_colsum;
    const float* rowscale; int64_t ld_rowscale, stride_rowscale;
} hh_qgemm_opts;
int hh_qgemm_f32x3(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, int mode,
                   const hh_qgemm_opts* opts, hh_stream_t stream);
/* Several INDEPENDENT products of one mode in one launch (round 6): the nine weight gradients of a decoder layer's backward
 * (tfm_decoder.py:420-461: autograd issues one cuBLAS call each), the q|k and v halves of its self-attention in-projection.  Item i is
 * exactly the argument list of hh_qgemm_f32x3 (all forms: split-K, batched, every prologue / epilogue option); results are bit-identical
 * to n single calls.  1 <= n <= HH_QGEMM_GROUP_MAX, all items of the same mode; products must not read what another product of the
 * group writes. */
#define HH_QGEMM_GROUP_MAX 12
typedef struct hh_qgemm_item {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int32_t M, N, K, mode;
    hh_qgemm_opts opts;
} hh_qgemm_item;
int hh_qgemm_f32x3_group(const hh_qgemm_item* items, int n, hh_stream_t stream);
/* self-attention over the Q <= 16 queries of a clip (nn.MultiheadAttention(q = k = x + query_pos, v = x), tfm_decoder.py:433-436):
 * qkv fp32 [B*Q, 3*heads*64] (q | k | v projections, not pre-scaled; the kernel applies 64^-1/2), out fp32 [B*Q, heads*64];
 * attention dropout by the same counter-based hash (probabilities and mask are recomputed in the backward). */
int hh_qself_attn_fwd(const float* qkv, float* out, int B, int Q, int heads, float dropout_p, uint32_t seed, hh_stream_t stream);
int hh_qself_attn_bwd(const float* qkv, const float* dout, float* dqkv, int B, int Q, int heads, float dropout_p, uint32_t seed,
                      hh_stream_t stream);
/* LayerNorm of clips of `tokens_per_clip` rows whose first row is the CLS token (the tower's final norm, LaviLa.py:572): the CLS rows go to
 * y_cls [clips, cols], the other rows, clip after clip, to y_patches [clips * (tokens_per_clip - 1), cols] -- the decoder's [B, T*n, D] grid
 * (tfm_decoder.py:200-205) without a strided copy of the feature map. */
int hh_layernorm_split_cls_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y_patches, void* y_cls, int y_dtype,
                               int64_t clips, int tokens_per_clip, int cols, float eps,
                              const void* x_lo /* NULL, or (x_dtype = HH_BF16) the low halves of a bf16 pair stream: the rows are x + x_lo */, hh_stream_t stream);
/* LayerNorm with a second output y_plus_pos = LN(x) + pos[row % pos_rows] (pos fp32 [pos_rows, cols]): the operands of an attention
 * whose keys / queries carry a positional embedding and whose values do not (tfm_decoder.py:431-441: q = k = x + query_pos, v = x;
 * key = memory + pos, value = memory) come out of ONE pass.  Same dtypes / limits as hh_layernorm_fwd. */
int hh_layernorm_pos_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, void* y_plus_pos, int y_dtype,
                         const float* pos, int pos_rows, float* mean_out, float* rstd_out, int64_t rows, int cols, float eps,
                         hh_stream_t stream);
/* hh_layernorm_bwd with the gradient of the residual path added: dx = dx_add + LayerNorm-backward(dy)  (dx_add may alias dx) */
int hh_layernorm_bwd_add(const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd, const float* dy,
                         const float* dx_add, float* dx, float* dgamma, float* dbeta, int64_t rows, int cols, hh_stream_t stream);

/* ---- Hungarian matching + box losses (model/box_utils.py:43-92,156-173,249-279; utils/box_ops.py:9-61)
 * pred fp32 [F, Qtot, 4] cxcywh; queries [q0, q0+q) are matched.  raw_boxes fp32 [F, k, 4] xyxy pixels
 * (zero / degenerate = absent, prepare_targets semantics, img = 224).  Outputs (all device):
 *   tgt_cxcywh fp32 [F,k,4] compacted valid targets, tgt_count int32 [F],
 *   match_pred int64 [F,k], match_tgt int64 [F,k] (first min(q,count) entries valid, pred ascending), match_n int32 [F]
 * Exact shortest-augmenting-path LSAP (scipy.optimize.linear_sum_assignment semantics) in fp64 on the fp32 cost
 * C = w_l1*L1 - w_giou*GIoU, one thread per frame. q <= 16, k <= 16. */
/* If given_count != NULL, raw_boxes already holds prepared cxcywh targets (first given_count[f] rows valid) and the
 * prepare_targets step is skipped (HungarianMatcher.forward list API).
 * class_cost (optional, fp32 [F, q, k]): the exclude_class=False term of box_utils.py:83-85, i.e. -softmax(pred_logits)[query,
 * label of the frame's j-th kept target], added as  C += w_class * class_cost  after the box terms (reference op order). */
int hh_match_boxes(const float* pred, int Qtot, int q0, int q, const float* raw_boxes, const int32_t* given_count, int k, float img,
                   float w_l1, float w_giou, const float* class_cost, float w_class, float* tgt_cxcywh, int32_t* tgt_count,
                   int64_t* match_pred, int64_t* match_tgt, int32_t* match_n, int64_t F, hh_stream_t stream);
/* generic batched LSAP on fp32 costs [P, nr, nc] (word loss, loss.py:83-93): the rows with row_valid != 0
 * (kept in order) x all nc columns; out col_of_row int64 [P, nr] (assigned column per row, -1 for invalid /
 * unassigned rows). nr <= 16, nc <= 16 */
int hh_lsap_rows(const float* cost, const uint8_t* row_valid, int64_t* col_of_row, int64_t P, int nr, int nc,
                 hh_stream_t stream);
/* matched-pair losses: sums over all frames of L1 and (1-GIoU) (un-normalised), and d(pred) given upstream
 * scalars g_l1, g_giou (dL/d(sum_l1), dL/d(sum_giou)).  sums fp32 [2] must be zeroed by the caller. */
int hh_box_loss_fwd(const float* pred, int Qtot, int q0, const float* tgt_cxcywh, int k,
                    const int64_t* match_pred, const int64_t* match_tgt, const int32_t* match_n,
                    float* sums, int64_t F, hh_stream_t stream);
int hh_box_loss_bwd(const float* pred, int Qtot, int q0, const float* tgt_cxcywh, int k,
                    const int64_t* match_pred, const int64_t* match_tgt, const int32_t* match_n,
                    const float* g_l1, const float* g_giou, float* dpred, int64_t F, hh_stream_t stream);

/* The scalar tail of the step's two box losses in one launch (box_utils.py:142-173,445-461; run/train.py:161-183): per box type t in
 * {hand, object}, from sums_t = hh_box_loss_fwd's (sum L1, sum (1 - GIoU)) and num_boxes[t] (the clamped, world-averaged normaliser):
 *   loss_bbox_t = sums_t[0] / nb_t;  loss_giou_t = sums_t[1] / nb_t;  total_t = (w_l1 * loss_bbox_t + w_giou * loss_giou_t) / denom;
 *   cardinality_error_t = mean over frames of | #(argmax[f, q0_t .. q0_t + q_t) != no_object) - count_t[f] |   (argmax int64 [F, Q] or NULL).
 * out fp32 [8] = total_h, total_o, loss_bbox_h, loss_giou_h, loss_bbox_o, loss_giou_o, card_h, card_o;  coef fp32 [4] = d total_h / d sums_h[0],
 * d total_h / d sums_h[1], d total_o / d sums_o[0], d total_o / d sums_o[1].  hh_box_loss_bwd_scaled = hh_box_loss_bwd with upstream gradients
 * g[0] * coef_l1[0] and g[0] * coef_giou[0]; it ADDS into dpred (both box types write disjoint query slices of one zeroed buffer). */
int hh_box_tail_fwd(const float* sums_h, const float* sums_o, const float* num_boxes, const int32_t* count_h, const int32_t* count_o,
                    const int64_t* argmax, int Q, int q0_h, int q_h, int q0_o, int q_o, int64_t no_object, int64_t F, float w_l1, float w_giou,
                    float denom, float* out, float* coef, hh_stream_t stream);
int hh_box_loss_bwd_scaled(const float* pred, int Qtot, int q0, const float* tgt_cxcywh, int k, const int64_t* match_pred,
                           const int64_t* match_tgt, const int32_t* match_n, const float* g, const float* coef_l1, const float* coef_giou,
                           float* dpred, int64_t F, hh_stream_t stream);

/* ---- loss tail (csrc/loss.hip): the step's small fp32 reductions as a handful of launches instead of ~250 stock elementwise ops.
 * hh_rownorm_fwd/bwd: y = x / max(||x||_2, eps) per row, the operand normalisation of sim_matrix (model/metric.py:363-375); x fp32
 * [rows, cols] row stride ldx, y fp32 [rows, cols] dense, norm fp32 [rows] (saved for the backward); dx = d loss / d x.
 * hh_egonce_fwd: EgoNCE (model/loss.py:8-70) for the call of run/train.py:144-149 -- multi_pad_mask = pad[:, None].repeat(1, Bg),
 * strict_mask: x fp32 [R*Bg, Bg] (row stride ldx; row i = rephrase i % R of clip i / R), sim_v / sim_n fp32 [Bg, Bg] (either may be
 * NULL), pad fp32 [R*Bg] (0 = caption absent, row dropped); positives ((sim_v*sim_n)[i/R, j] + [i/R == j]) * pad_i > vn_threshold.
 * Writes loss[0] and grad fp32 [R*Bg, Bg] = d loss / d x; scratch: hh_workspace_bytes_egonce(R, Bg) bytes.
 * hh_masked_ce_fwd: the 582-way cross-entropy of WordContrastiveLoss (model/loss.py:95-104): sim fp32 [rows, V] (row stride ld),
 * noun_sim fp32 [V, V] (cosine similarity of the nouns; its diagonal is ignored), gt int64 [rows], valid uint8 [rows]; logits sim / T
 * with the columns k != gt of noun_sim[gt, k] > threshold replaced by -1 / T.  ce fp32 [rows] (0 on invalid rows), grad fp32 [rows, V]
 * = d ce[r] / d sim[r, :].  A valid row whose gt is outside [0, V) gets ce = NaN and a NaN gradient row (F.cross_entropy raises there).
 * hh_tv_accuracy: compute_tv_accuracy (model/metric.py:378-392): sim fp32 [Bg, Bg] (row stride ld), text_cos fp32 [Bg, Bg] (cosine
 * similarity of the first captions), sim_v, sim_n fp32 [Bg, Bg] -> out[0] = video->text, out[1] = text->video top-1 accuracy. */
/* hh_text_flags: per caption row of the token matrix text int64 [rows, L] (run/train.py:124,144): eot[r] = argmax_l text[r, l] (first index on
 * ties), pad[r] = 1.0 if the row holds anything but exactly two non-zero tokens ([SOT, EOT] = an empty rephrase slot), else 0.0. */
int hh_text_flags(const int64_t* text, int rows, int L, int64_t* eot, float* pad, hh_stream_t stream);
int hh_rownorm_fwd(const float* x, int64_t ldx, float* y, float* norm, int rows, int cols, float eps, hh_stream_t stream);
int hh_rownorm_bwd(const float* y, const float* norm, const float* dy, int64_t lddy, float* dx, int rows, int cols, float eps,
                   hh_stream_t stream);
int64_t hh_workspace_bytes_egonce(int R, int Bg);
int hh_egonce_fwd(const float* x, int64_t ldx, const float* sim_v, const float* sim_n, const float* pad, int R, int Bg,
                  float temperature, float vn_threshold, float* loss, float* grad, float* scratch, hh_stream_t stream);
int hh_masked_ce_fwd(const float* sim, int64_t ld, const float* noun_sim, const int64_t* gt, const unsigned char* valid, int rows, int V,
                     float temperature, float threshold, float* ce, float* grad, hh_stream_t stream);
int hh_tv_accuracy(const float* sim, int64_t ld, const float* text_cos, const float* sim_v, const float* sim_n, int Bg, float* out,
                   hh_stream_t stream);

/* ---- fused AdamW over a flat fp32 parameter arena (torch.optim.AdamW semantics; run/train.py:199-203,520)
 * p,g,m,v fp32 [n]; decay_mask uint8 [n] or per-segment handled by caller through two calls.  step >= 1. */
int hh_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, hh_stream_t stream);
/* The whole arena in one call, with the per-parameter decisions of torch.optim.AdamW taken on the DEVICE (no host-side launch plan,
 * no host synchronisation): the arena is n_seg segments [seg_off[s], seg_off[s+1]) (int64 [n_seg+1], 4-element aligned, seg_off[0]
 * = 0, seg_off[n_seg] = n), one per parameter.  A segment is updated iff seg_flag[s] > 0 (the parameter received a gradient this
 * step -- torch skips grad-less parameters entirely, no weight decay either; under data parallelism the caller all-reduces the
 * flags so that every rank decides alike); its step counter seg_step[s] (int32, in/out) is then incremented and supplies the bias
 * correction.  seg_decay[s] != 0 selects weight_decay, else 0 (utils/train_utils.py:28-48: the two param groups).  seg_coef: fp32
 * [2*n_seg] scratch.  zero_grads != 0 additionally clears g (the next step's optimizer.zero_grad(), run/train.py:199).  Updates
 * equal hh_adamw_step's on the same segment and step (same formulas; to fp32 rounding). */
int hh_adamw_arena_step(float* p, float* g, float* m, float* v, int64_t n, const int64_t* seg_off, const int* seg_decay,
                        int* seg_step, const float* seg_flag, float* seg_coef, int n_seg, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int zero_grads, hh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HH_H */
