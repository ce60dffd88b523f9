/* dsw_hip.h - C ABI of libdsw_hip.so: the MI355X (gfx950) hot path of DeepSphere-Weather.
 *
 * The reference is pure Python; its "FFI" for this path is the two stock PyTorch ops called from
 * modules/layers.py (torch.sparse.mm at :164, :167, :962 and Tensor.matmul at :177) plus the layout
 * copies around them (:158-160, :165, :168, :171-173).  The entry points below are what the
 * reference's conv_cheb / RemapBlock.forward (and their autograd backward) bind instead; the
 * Python-side binding (ctypes) is shown in INTEGRATION.md and lives in
 * deepsphere-weather_amd/dsw_amd/_native.py.
 *
 * Conventions
 *   - plain C, no exceptions; every call returns 0 on success or a negative DSW_ERR_* code
 *     (dsw_strerror() gives the text); the Python side turns it into RuntimeError.
 *   - all data pointers are DEVICE pointers owned by the caller (torch tensors); the library never
 *     allocates.  Scratch memory is passed in (dsw_cheb_bwd_workspace_bytes tells how much).
 *   - every call enqueues work on `stream` (a hipStream_t, e.g. torch.cuda.current_stream()) and
 *     returns immediately; calls are re-entrant and thread-safe (autograd runs backward on its
 *     own thread).
 *   - activations are node-major: [B, V, C] contiguous (sample, node, channel) - the layout
 *     ConvCheb.forward receives (layers.py:365-376) - never the reference's internal [V, Fin*B].
 *   - operators are CSR with int32 rowptr[v_out+1] / colind[nnz] and FP32 values (also for bf16
 *     activations: the operator and every accumulation stay fp32).
 *   - dtype: DSW_F32 (0) or DSW_BF16 (1) is the storage type of activations, weights and grads.
 */
#ifndef DSW_HIP_H
#define DSW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSW_F32 0
#define DSW_BF16 1

#define DSW_OK 0
#define DSW_ERR_BAD_ARG (-1)
#define DSW_ERR_BAD_DTYPE (-2)
#define DSW_ERR_WORKSPACE (-3)
#define DSW_ERR_LAUNCH (-4)
#define DSW_ERR_ALIGN (-5)

typedef void* dsw_stream_t; /* hipStream_t */

/* Tile plan of the fused two-hop SpMM (built once per operator on the host, see
 * deepsphere-weather_amd/dsw_amd/hop2.py): for every tile, the list of rows it gathers (tile rows, then its
 * 1-ring, then its 2-ring) and the CSR of tile + 1-ring rows with columns rewritten as positions in that list.
 * A tile is `tile_rows` CONSECUTIVE rows (explicit_tiles = 0: tile t = rows t * tile_rows ..., what HEALPix nested
 * order wants) or ANY set of <= tile_rows rows (explicit_tiles = 1: the first tile_meta[t][5] entries of its gather
 * list; the builder clusters the operator's graph, so that samplings whose row order is not 2-D local - equiangular
 * row-major, HEALPix ring order - still get compact neighbourhoods).  All pointers are device pointers. */
typedef struct dsw_hop2_plan {
    int32_t n_tiles, tile_rows, max_n1, max_n2, max_nnz;
    int32_t reserved;          /* longest local CSR row (the kernel's ELL width before rounding up to 4) */
    const int32_t* tile_meta;  /* [n_tiles][6]: s2_off, n1, n2, nnz_off, rp_off, rows of the tile (explicit_tiles) */
    const int32_t* s2_rows;    /* concatenated gather lists (global row ids) */
    const int32_t* lrowptr;    /* concatenated local row pointers, n1 + 1 per tile, tile-relative */
    const uint16_t* lcol;      /* concatenated local column positions */
    const float* lval;         /* concatenated values */
    int32_t explicit_tiles;
    int32_t hops;              /* 2 (or 0): plan of the fused two-hop kernel; 1: plan of the staged ONE-hop kernel - the
                                  gather list ends with the 1-ring, local CSR of the tile rows only (n1 = tile rows) */
    int32_t ell_w;             /* > 0 (hops == 1, <= 64 rows per tile, <= ell_w entries per row): the tile rows' stencils
                                  ALSO as a padded ELL image, 64 rows x ell_w entries per tile, and tile_meta[t][5] = the
                                  tile's longest row */
    const uint16_t* ell_pos;   /* [n_tiles][64][ell_w] list positions (padding: the row's own position) */
    const float* ell_val;      /* [n_tiles][64][ell_w] values (padding: 0) */
    /* hops == 2 (or 0), consecutive tiles (explicit_tiles == 0): the stencils of a tile's rows and one-ring as the padded ELL
     * image the one-launch kernels (dsw_fwd3.hip, dsw_bwd3d.hip) keep in LDS, so that a workgroup's prologue is ONE round of
     * loads instead of row pointers -> columns / values -> expansion: per tile max_n1 rows x W entries, W = (reserved + 3) & ~3,
     * values first, then the u8 list positions (padding: {own row, 0}), ell2_stride bytes apart; tile_meta[t][5] = the tile's
     * longest row.  NULL: the kernels expand the local CSR themselves. */
    const unsigned char* ell2;
    int64_t ell2_stride;
    int64_t struct_bytes;      /* sizeof(dsw_hop2_plan) of the header the CALLER was built with (since 0.1.1).  Every entry point
                                  that takes a plan refuses one of another size (DSW_ERR_BAD_ARG; the *_supported predicates
                                  say 0): the struct grew between versions, and a caller built against an older header would
                                  otherwise have the library read past the end of its struct. */
} dsw_hop2_plan;

/* Library version (major*10000 + minor*100 + patch). */
int dsw_version(void);

/* Text for an error code. */
const char* dsw_strerror(int code);

/* Diagnostics switches the library's objects were compiled with: 0 for the product build; bit 0 = -DDSW_DIAG (environment
 * overrides of the kernel selection), bit 1 = an ablation switch (-DDSW_ABLATION, -DDSW_ABL_*: wrong results by design).
 * dsw_amd/_native.py refuses a flagged library unless it was asked for by path (DSW_HIP_LIB). */
int dsw_build_flags(void);

/* Launch tracing: per-role durations of the kernels the entry points launch.  While a trace is open every kernel of the
 * library is launched with a start / stop event pair attached to its own dispatch (hipExtLaunchKernelGGL: the timestamps of
 * the kernel's completion signal, what a profiler's kernel trace reports - no marker packets between the kernels), and the
 * entry points note on the host which role the kernels belong to.  A benchmark opens a trace, runs steps the ordinary way
 * (eagerly: nothing is recorded inside a stream capture) and reads the roles back - every kernel timed INSIDE the step, with
 * the caches in the state the step leaves them, which back-to-back calls of one kernel are not (isolated timings
 * were 11-14 % off the in-step durations).  One trace at a time, process-wide.
 *   dsw_trace_begin(capacity)  room for `capacity` kernel launches; starts recording.
 *   dsw_trace_end(...)         waits for the recorded kernels and writes up to `cap` KERNEL records in launch order: index of
 *                              the role call the kernel belongs to, its role and three role-specific integers, the kernel's
 *                              duration in us, and the launch-site name of the kernel (names: cap x name_stride chars).
 *                              Returns the number of records (> cap: truncated), DSW_ERR_WORKSPACE if the capacity
 *                              overflowed, another error code on failure.
 * Kernels launched eagerly carry heavier release fences than the same kernels replayed from a HIP graph and run ~10 % longer
 * (measured); bench.py therefore uses the trace for the ORDER and ROLE of the kernels of a step and takes their durations
 * from a rocprofv3 kernel trace of the replayed graph.
 * aux: conv roles (V, Fin, Fout) [recurrences: (V, C, K)], SPMM roles (rows_out, rows_in, C), elementwise (n, kind, 0). */
#define DSW_ROLE_SPMM 1            /* dsw_spmm_csr / dsw_spmm_csr_ld: one product (interpolation pooling and its transpose) */
#define DSW_ROLE_SPMM2 2           /* dsw_spmm2_fused called directly */
#define DSW_ROLE_SPMM_STAGED 3     /* dsw_spmm_staged called directly */
#define DSW_ROLE_BASIS_FWD 4       /* forward recurrence launches (T_1 .. T_{K-1}) */
#define DSW_ROLE_BASIS_ADJ 5       /* adjoint recurrence launches (dgrad planes -> dX) */
#define DSW_ROLE_MIX_FWD 6         /* channel-mix GEMM of the forward */
#define DSW_ROLE_FWD_ONE_LAUNCH 7  /* whole forward (hops + channel mix + bias) in one launch */
#define DSW_ROLE_BWD_GEMM_FUSED 8  /* dW partials + db + dgrad planes in one pass over dY, + the partial reduce */
#define DSW_ROLE_BWD_DGRAD 9       /* dgrad GEMM (planes G_k, or dX of a mix-first layer) */
#define DSW_ROLE_BWD_WGRAD 10      /* dW / db (+ reduce) */
#define DSW_ROLE_ZMIX 11           /* mix-first forward: plane GEMM Z_k = X W_k */
#define DSW_ROLE_CLENSHAW_FWD 12   /* mix-first forward: Clenshaw recurrence on the output channels */
#define DSW_ROLE_ELEMENTWISE 13    /* relu mask (kind 1), ReZero residual forward (2) / backward (3), ReZero parameter gradients (4) */
#define DSW_ROLE_BWD_FUSED 14      /* (retired in round 6 with dsw_cheb_dx_one_launch: never emitted) */
#define DSW_ROLE_FWD_HOP2_MIX 16     /* hop 2 + channel mix + bias of the forward in one launch (one-hop plans) */
#define DSW_ROLE_BASIS_DUAL 15     /* mix-first backward: Chebyshev basis of dY under L^T (on the output channels) */
#define DSW_ROLE_BWD_DUAL 17       /* whole backward in one launch in the dual form (X, dY -> dX, dW, db), + the partial reduce */
int dsw_trace_begin(int capacity);
int dsw_trace_end(int32_t* call, int32_t* roles, int32_t* aux0, int32_t* aux1, int32_t* aux2, float* us, char* names,
                  int name_stride, int cap);

/* Sparse operator times node-major activations with a fused axpby epilogue, per sample:
 *     Y[b,r,:] = alpha * sum_p vals[p] * X[b,colind[p],:] + beta * Z[b,r,:] + gamma * Z2[b,r,:]
 * X: [B, v_in, C]; Y, Z, Z2: [B, v_out, C]; Z / Z2 may be NULL; Y may alias Z or Z2 (not X).
 * Replaces torch.sparse.mm at layers.py:164 (alpha=1), :167 fused with "2*(...) - x0"
 * (alpha=2, beta=-1), :962 (RemapBlock, rectangular operator), and their transposed autograd
 * counterparts (pass the CSR of the transposed operator). */
int dsw_spmm_csr(const int32_t* rowptr, const int32_t* colind, const float* vals,
                 int64_t v_out, int64_t v_in, int64_t nnz,
                 const void* X, void* Y, int64_t B, int64_t C,
                 float alpha, const void* Z, float beta, const void* Z2, float gamma,
                 int dtype, dsw_stream_t stream);

/* The same product on channel SLICES of wider node-major tensors: ldx / ldy / ldz = elements between consecutive rows
 * of X / Y / Z (>= C; Z2 stays dense [B, v_out, C]; ldz is ignored when Z is NULL).  The U-Net decoder's
 * `torch.cat((unpooled, skip), dim=2)` (my_models_graph.py:528-545) disappears with it: the unpooling writes the left
 * half of the concatenation buffer (ldy = its width), the encoder block wrote the right half
 * (dsw_rezero_residual_fwd_ld), the pooling reads that half (ldx), the backward of the unpooling reads its half of the
 * buffer's gradient (ldx), and the backward of the pooling adds the skip half of that gradient in its epilogue
 * (Z with ldz, beta = 1: the gradient accumulation of a tensor with two consumers without a separate add pass). */
int dsw_spmm_csr_ld(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t v_out, int64_t v_in,
                    int64_t nnz, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t B, int64_t C, float alpha,
                    const void* Z, int64_t ldz, float beta, const void* Z2, float gamma, int dtype, dsw_stream_t stream);

/* Plan of an interpolation-pooling (remap) matrix, built once per matrix on the host (dsw_amd/functional.py:
 * CsrOperator.remap_plan) - what lets RemapBlock's product (layers.py:956-964) and its transposed backward skip the generic
 * CSR walk:
 *   kind 1 (GROUPS)     every row has exactly m entries, in columns m r .. m r + m - 1 (v_in = m v_out): the pooling between two
 *                       levels of a regular hierarchy (HEALPix nested order: 4 children per parent), and the transposed
 *                       unpooling.  Pure streaming: m row loads, one store, weights = the CSR values (any values).
 *   kind 2 (BROADCAST)  every row has exactly one entry, in column r / m (v_out = m v_in): the unpooling of such a hierarchy
 *                       and the transposed pooling.  One row load, m stores.
 *   kind 0 (GENERIC)    any matrix (pooling between different samplings): the rows with more than long_thr entries (polar
 *                       cells of a cross-sampling matrix: up to hundreds) are LISTED, each gets whole waves; the launch needs
 *                       no scan of the row lengths and the lane-group-per-row path never walks a long row. */
typedef struct dsw_remap_plan {
    int32_t kind, m;
    int32_t long_thr, n_long;
    const int32_t* long_rows;   /* device pointer, n_long rows (kind 0); NULL otherwise */
    int32_t parts;              /* kind 0: lane groups that share a row (1, 2 or 4: rows of 6+ entries on average are split so
                                   that a row is not a chain of dependent index -> data round trips); 0 / 1 = one group per row */
    int32_t reserved;
} dsw_remap_plan;

/* Y[b,r,:] = sum_p vals[p] X[b,colind[p],:] + beta Z[b,r,:] for a remap matrix with a plan (NULL plan = dsw_spmm_csr_ld with
 * alpha = 1): row strides ldx / ldy / ldz as in dsw_spmm_csr_ld, Z optional.  The CSR arrays are always passed (kind 1 / 2 read
 * only `vals`); a regular plan whose m does not fit the shape / nnz returns DSW_ERR_BAD_ARG. */
int dsw_remap_csr(const dsw_remap_plan* plan, const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t v_out,
                  int64_t v_in, int64_t nnz, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t B, int64_t C,
                  const void* Z, int64_t ldz, float beta, int dtype, dsw_stream_t stream);

/* Two applications of one square operator A (V x V) in a single launch, per sample:
 *     Y1 = a1 * (A U)  + b1 * Z1 + d1 * Z1b
 *     Y2 = a2 * (A Y1) + b2 * U  + c2 * Z2
 * U, Z*, Y*: [B, V, C]; Z1 / Z1b / Z2 / Y1 may be NULL (Y1 then stays on chip).  Y2 may alias Z2;
 * Y1 must not alias Z1 / Z1b / U.  Returns DSW_ERR_BAD_ARG if dsw_spmm2_supported() is 0. */
int dsw_spmm2_fused(const dsw_hop2_plan* plan, int64_t V, const void* U, const void* Z1,
                    const void* Z1b, const void* Z2, void* Y1, void* Y2, int64_t B, int64_t C,
                    float a1, float b1, float d1, float a2, float b2, float c2,
                    int dtype, dsw_stream_t stream);

/* 1 if the fused two-hop kernel can take this plan and row size (LDS capacity, 16-byte lanes). */
int dsw_spmm2_supported(const dsw_hop2_plan* plan, int64_t C, int dtype);

/* One application of a square operator with the tile's neighbourhood staged in LDS (plan->hops == 1), per sample:
 *     Y = a * (A U) + b * Z + c * Z2
 * U, Z, Z2, Y: [B, V, C]; Z / Z2 may be NULL; Y may alias Z or Z2 (a tile reads them on its own rows only), not U.
 * The form the recurrences take on DENSE stencils (the reference's default k = 20 graph, utils_config.py:50: 21-23
 * entries per row), where the first-hop redundancy of the fused pair costs more than the round trip of the
 * intermediate plane it saves.  stream_out = 1: nontemporal stores (nothing gathers Y next).
 * Returns DSW_ERR_BAD_ARG if dsw_spmm_staged_supported() is 0. */
int dsw_spmm_staged(const dsw_hop2_plan* plan, int64_t V, const void* U, const void* Z, const void* Z2, void* Y,
                    int64_t B, int64_t C, float a, float b, float c, int dtype, dsw_stream_t stream, int stream_out);
int dsw_spmm_staged_supported(const dsw_hop2_plan* plan, int64_t C, int dtype);

/* Chebyshev basis T_1 .. T_{K-1} of X (T_0 = X is not copied):
 *     T_1 = L X,  T_k = 2 L T_{k-1} - T_{k-2}        (layers.py:163-169)
 * T: [K-1, B, V, C].  K <= 1 is a no-op.  With a (supported) plan of L, hops run pairwise fused (plan->hops == 2) or one
 * staged launch per hop (plan->hops == 1). */
int dsw_cheb_basis_fwd(const int32_t* rowptr, const int32_t* colind, const float* vals,
                       int64_t V, int64_t nnz, const void* X, void* T,
                       int64_t B, int64_t C, int64_t K, int dtype, dsw_stream_t stream,
                       const dsw_hop2_plan* plan);

/* Adjoint of the Chebyshev recurrence (what autograd derives for layers.py:163-169), in place on
 * the K gradient planes G_0 (= the dX buffer, [B,V,C]) and G_1..G_{K-1} (Grest, [K-1,B,V,C]):
 *     for j = K-1 .. 1:   G_{j-1} += (j > 1 ? 2 : 1) * L^T G_j - G_{j+1}        (G_K := 0)
 * after which G_0 holds dX.  rowptr_t/colind_t/vals_t: CSR of L^T.
 * With a (supported) plan of L^T the steps run pairwise fused; `spare` ([2,B,V,C], required when a
 * plan is given and K >= 4, else may be NULL) receives the intermediates that cannot be updated in
 * place. */
int dsw_cheb_basis_adj(const int32_t* rowptr_t, const int32_t* colind_t, const float* vals_t,
                       int64_t V, int64_t nnz, void* G0, void* Grest,
                       int64_t B, int64_t C, int64_t K, int dtype, dsw_stream_t stream,
                       const dsw_hop2_plan* plan_t, void* spare);

/* Channel mix on the matrix cores:  Y[n,o] = bias[o] + sum_{k,f} T_k[n,f] * W[f,k,o]
 * (layers.py:171-178 and the bias add at :375).  X = T_0 [N,Fin]; T = T_1.. [K-1,N,Fin];
 * W: [Fin,K,Fout] (the reference's parameter layout); bias [Fout] or NULL; Y: [N,Fout]. */
int dsw_cheb_mix_fwd(const void* X, const void* T, const void* W, const void* bias, void* Y,
                     int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype,
                     dsw_stream_t stream);

/* Evaluation order chosen for a layer shape (1 = mix-first, 0 = basis-first).  Layers that shrink the channel
 * count (2 * Fout <= Fin, K >= 2) evaluate  Y = sum_k T_k(L) (X W_k)  - channel mix first, then the recurrence
 * in Clenshaw form on the Fout channels (same value as layers.py:163-178 up to fp32 rounding; the SpMM hops, the
 * HBM-bound part, run on Fout instead of Fin channels).  Callers use it to size the tile plan they pass
 * (rows of Fout vs Fin channels) and to know whether T comes back as the basis. */
int dsw_cheb_mix_first(int64_t Fin, int64_t Fout, int64_t K);

/* Which launch sequence dsw_cheb_fwd takes for a layer shape and plan (pointer alignment aside) - what a profile of a
 * training step should be read against:
 *   DSW_FWD_PLAIN_HOPS   one dsw_spmm_csr launch per hop, then the channel-mix GEMM
 *   DSW_FWD_FUSED_PAIRS  hops pairwise in the two-hop kernel (plan->hops == 2), then the GEMM
 *   DSW_FWD_STAGED_HOPS  one staged launch per hop (plan->hops == 1: dense stencils), then the GEMM
 *   DSW_FWD_ONE_LAUNCH   both hops AND the channel mix in one launch (fp32, K = 3, Fin = 32, Fout 32 / 64, two-hop plan)
 *   DSW_FWD_MIX_FIRST    channel mix first, recurrence on the Fout channels (dsw_cheb_mix_first)
 *   DSW_FWD_HOP1_THEN_ONE_LAUNCH  staged hop 1, then hop 2 AND the channel mix in one launch (fp32, K = 3, Fin = 32, Fout 32 / 64,
 *                        one-hop plan: the reference's default k = 20 graph, equiangular; needs the basis buffer T) */
#define DSW_FWD_PLAIN_HOPS 0
#define DSW_FWD_FUSED_PAIRS 1
#define DSW_FWD_STAGED_HOPS 2
#define DSW_FWD_ONE_LAUNCH 3
#define DSW_FWD_MIX_FIRST 4
#define DSW_FWD_HOP1_THEN_ONE_LAUNCH 5
int dsw_cheb_fwd_path(const dsw_hop2_plan* plan, int64_t Fin, int64_t Fout, int64_t K, int dtype);

/* Whole ConvCheb.forward (layers.py:365-376) = dsw_cheb_basis_fwd + dsw_cheb_mix_fwd.
 * T ([K-1,B,V,Fin], may be NULL iff K == 1) receives the basis and is what backward needs.
 * When dsw_cheb_mix_first(Fin, Fout, K): T is required as SCRATCH of the same size (its content afterwards is
 * unspecified; backward then needs only X, W, dY and accepts T = NULL), and `plan` / `plan_t` must be built for
 * rows of Fout channels. */
int dsw_cheb_fwd(const int32_t* rowptr, const int32_t* colind, const float* vals,
                 int64_t V, int64_t nnz, const void* X, const void* W, const void* bias,
                 void* Y, void* T, int64_t B, int64_t Fin, int64_t Fout, int64_t K,
                 int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan);

/* ConvCheb.forward followed by the activation of the enclosing ConvBlock (my_models_graph.py:104-118: conv -> relu
 * when batch norm does not sit in between): act = DSW_ACT_RELU applies max(., 0) in the epilogue of the channel-mix
 * GEMM (after the bias; one in-place pass for mix-first layers, whose last stage is an SpMM) instead of a separate
 * read + write of Y.  act = DSW_ACT_NONE is dsw_cheb_fwd. */
#define DSW_ACT_NONE 0
#define DSW_ACT_RELU 1
int dsw_cheb_fwd_act(const int32_t* rowptr, const int32_t* colind, const float* vals,
                     int64_t V, int64_t nnz, const void* X, const void* W, const void* bias,
                     void* Y, void* T, int64_t B, int64_t Fin, int64_t Fout, int64_t K,
                     int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan, int act);

/* ConvCheb.forward followed by the tail of the enclosing ResBlock (my_models_graph.py:205-216:
 * `x_out *= self.rezero_weight; x_out += self.res_connection(x)`), in the epilogue of the channel-mix GEMM:
 *     Y = act( scale * (sum_k T_k W_k + bias) + R )
 * scale: ONE device scalar of the data dtype (the ReZero parameter) or NULL (= 1); R: [B*V, ldr >= Fout] or NULL; ldy:
 * elements between consecutive rows of Y (0 or Fout = dense; > Fout = a channel slice of a wider tensor, e.g. the
 * decoder's concatenation buffer - basis-first layers only, mix-first layers return DSW_ERR_BAD_ARG for it).  With
 * scale = R = NULL and ldy = 0 this is dsw_cheb_fwd_act.  Saves the separate read-read-write pass over the block's
 * output (and the write + read of the unscaled convolution in between). */
int dsw_cheb_fwd_res(const int32_t* rowptr, const int32_t* colind, const float* vals,
                     int64_t V, int64_t nnz, const void* X, const void* W, const void* bias,
                     void* Y, int64_t ldy, void* T, int64_t B, int64_t Fin, int64_t Fout, int64_t K,
                     int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan, int act,
                     const void* scale, const void* R, int64_t ldr);

/* dsw_cheb_fwd_res with caller-owned scratch (dsw_cheb_fwd_workspace_bytes; NULL / 0 = none): room for a per-call image of
 * the weights.  The channel-mix GEMM of WIDE fp32 layers streams W chunk by chunk next to the activations and evaluates
 * the fp32 product on the bf16 matrix pipe from three-term splits of both operands; with the scratch W is split ONCE per call
 * (1.5x its fp32 bytes) instead of once per workgroup and chunk, and mix-first layers on dense stencils fold plane K-1 into
 * the weights of plane K-3 there (one epilogue operand less in the recurrence).  The scratch also holds one partial output
 * tile per workgroup (+ 4 KiB of flags, cleared by the call): with it the streaming GEMM may divide its reduction steps
 * evenly over the workgroups, tiles cut between two of them and summed in a fixed order (dsw_gemm_x3s.hip).  Results with
 * and without scratch agree to fp32 rounding (the summation order of a cut tile differs), each bit-identical from run to run.
 * Calls that share a scratch buffer must be stream-ordered. */
int64_t dsw_cheb_fwd_workspace_bytes(int64_t B, int64_t V, int64_t Fin, int64_t Fout, int64_t K, int dtype);
int dsw_cheb_fwd_ws(const int32_t* rowptr, const int32_t* colind, const float* vals,
                    int64_t V, int64_t nnz, const void* X, const void* W, const void* bias,
                    void* Y, int64_t ldy, void* T, int64_t B, int64_t Fin, int64_t Fout, int64_t K,
                    int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan, int act,
                    const void* scale, const void* R, int64_t ldr, void* workspace, int64_t workspace_bytes);

/* Backward of that activation (autograd of F.relu, what the reference's ConvBlock records):
 *     dYm[i] = Y[i] > 0 ? dY[i] : 0          (Y = the layer's activated output; n elements; dYm may alias dY) */
int dsw_relu_bwd(const void* dY, const void* Y, void* dYm, int64_t n, int dtype, dsw_stream_t stream);

/* 0 if dsw_cheb_bwd of this layer shape works WITHOUT the forward's basis planes (T = NULL): mix-first layers, K = 1, and
 * layers whose whole backward runs in one launch in the dual form (dsw_bwd3d.hip: fp32, K = 3, 32 -> 64 or - round 6 -
 * 32 -> 32, two-hop plan of L^T, V % 64 == 0):
 *     U_k = T_k(L^T) dY on chip,   dX = sum_k U_k W_k^T,   dW_k = X^T U_k,   db = 1^T dY
 * - X and dY in, dX out, no T_1 / T_2 and no dgrad planes.  The forward of such a layer may then be called with T = NULL
 * (dsw_cheb_fwd: "inference" form, the basis is not stored).  1: T is required.  plan_t: the plan dsw_cheb_bwd will get. */
int dsw_cheb_bwd_needs_basis(const dsw_hop2_plan* plan_t, int64_t V, int64_t Fin, int64_t Fout, int64_t K, int dtype);

/* Scratch bytes dsw_cheb_bwd needs for this problem size. */
int64_t dsw_cheb_bwd_workspace_bytes(int64_t B, int64_t V, int64_t Fin, int64_t Fout, int64_t K,
                                     int dtype);

/* Backward of ConvCheb (closed form of what autograd derives from layers.py:113-180,375):
 *     dW[f,k,o] = sum_n T_k[n,f] dY[n,o];  db[o] = sum_n dY[n,o];
 *     G_k = dY W[:,k,:]^T;  for j=K-1..1: G_{j-1} += (j>1 ? 2 : 1) L^T G_j - G_{j+1};  dX = G_0
 * rowptr_t/colind_t/vals_t describe the CSR of L^T (for a symmetric L pass L itself).
 * T may be NULL where dsw_cheb_bwd_needs_basis() == 0 (mix-first layers; the one-launch dual form, which is taken whenever the
 * shape and plan_t allow it - with or without T - and for which plan_t is then REQUIRED).
 * dX / dW / db may be NULL to skip them (db is only produced together with dW).
 * plan_t: optional two-hop plan of L^T (NULL = one launch per adjoint step).
 * B * V == 0 (an empty batch shard) writes dW = 0 and db = 0. */
int dsw_cheb_bwd(const int32_t* rowptr_t, const int32_t* colind_t, const float* vals_t,
                 int64_t V, int64_t nnz, const void* X, const void* T, const void* W,
                 const void* dY, void* dX, void* dW, void* db, void* workspace,
                 int64_t workspace_bytes, int64_t B, int64_t Fin, int64_t Fout, int64_t K,
                 int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan_t);

/* dsw_cheb_bwd with the two things a fused residual block needs from it:
 *   scale   (device scalar or NULL): dX = scale * (closed form) - the backward of Y = scale * conv(X) takes dY as it
 *           arrives, no scaled copy of it is made; dW / db stay UNSCALED (dW_raw = sum_n T_k dY): see
 *           dsw_rezero_param_grads;
 *   dX_add  ([B*V, ld_add >= Fin] or NULL): added to dX in the epilogue of the dgrad GEMM - the gradient a second
 *           consumer of X sent (residual branch: my_models_graph.py:213), instead of an autograd `add` pass;
 *   accumulate_dw (0 / 1): dW += and db += instead of overwriting them - dW / db are then the parameter's gradient
 *           buffers, which already hold the contributions of its other uses (an autoregressive training step applies
 *           every layer once per forward: 7 uses -> 7 x 38 tiny `add` launches of autograd's AccumulateGrad otherwise).
 *           The summation order is the call order (deterministic). */
int dsw_cheb_bwd_res(const int32_t* rowptr_t, const int32_t* colind_t, const float* vals_t,
                     int64_t V, int64_t nnz, const void* X, const void* T, const void* W,
                     const void* dY, void* dX, void* dW, void* db, void* workspace,
                     int64_t workspace_bytes, int64_t B, int64_t Fin, int64_t Fout, int64_t K,
                     int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan_t,
                     const void* scale, const void* dX_add, int64_t ld_add, int accumulate_dw);

/* Parameter gradients behind Y = scale * conv(X) + R when the backward ran on dY (dsw_cheb_bwd_res):
 *     dW = scale * dW_raw,  db = scale * db_raw,  dscale = <W, dW_raw> + <bias, db_raw>   (= sum dY * conv(X))
 * n_w / n_b elements (n_b = 0: no bias); dW / db may alias dW_raw / db_raw; one launch, deterministic.
 * workspace: dsw_rezero_param_grads_workspace_bytes() bytes that are ZERO before the first use and belong to this entry
 * point from then on (it leaves them zeroed-where-it-matters for the next call; calls sharing a workspace must be
 * stream-ordered). */
int64_t dsw_rezero_param_grads_workspace_bytes(void);
int dsw_rezero_param_grads(const void* W, const void* bias, const void* dW_raw, const void* db_raw, const void* scale,
                           void* dW, void* db, void* dscale, int64_t n_w, int64_t n_b, void* workspace,
                           int64_t workspace_bytes, int dtype, dsw_stream_t stream);

/* Residual block epilogue (my_models_graph.py:205-216: `x_out *= rezero_weight; x_out += res_connection(x)`):
 *   forward   y[i] = w * c[i] + r[i]        (w: ONE device scalar of the data dtype - the ReZero parameter)
 *   backward  grad_c[i] = w * g[i] (grad_c may be NULL),  grad_w = sum_i g[i] * c[i]   (deterministic two-stage sum);
 *             the gradient of r is g itself.
 * n elements; 16-byte aligned tensors take the 16 B/lane path, any other alignment a scalar path of the same kernel;
 * y may alias c or r.  workspace: dsw_rezero_residual_workspace_bytes(). */
int64_t dsw_rezero_residual_workspace_bytes(void);
int dsw_rezero_residual_fwd(const void* c, const void* r, const void* w, void* y, int64_t n, int dtype,
                            dsw_stream_t stream);
int dsw_rezero_residual_bwd(const void* g, const void* c, const void* w, void* grad_c, void* grad_w,
                            void* workspace, int64_t workspace_bytes, int64_t n, int dtype, dsw_stream_t stream);
/* forward with a row stride on the output: y[row * ldy + j] = w * c[row * C + j] + r[row * C + j] (rows x C dense inputs,
 * ldy >= C elements; C, ldy multiples of 16 bytes and 16-byte aligned pointers, else DSW_ERR_ALIGN). */
int dsw_rezero_residual_fwd_ld(const void* c, const void* r, const void* w, void* y, int64_t rows, int64_t C, int64_t ldy,
                               int dtype, dsw_stream_t stream);

/* Max-value pooling over a sparse remap matrix (layers.py:1040-1079, GeneralMaxValPool.forward: the Python Counter
 * loop + torch.gather / argmax per coarse row):  for every sample b, coarse row d, channel f
 *     p* = argmax over the non-zeros p of row d of  vals[p] * X[b, colind[p], f]   (first maximum on ties)
 *     Y[b,d,f] = X[b, colind[p*], f]  (unweighted),   sel[b,d,f] = colind[p*]      (-1 for an empty row, Y = 0)
 * X: [B, v_in, C]; Y: [B, v_out, C]; sel: int32 [B, v_out, C] - the compact form of the reference's [2, B*C*v_out]
 * int64 index tensor (row = sel, column = f*B + b, listed column-major; see GeneralMaxValPool.reference_index). */
int dsw_maxval_pool_fwd(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t v_out, int64_t v_in,
                        const void* X, void* Y, int32_t* sel, int64_t B, int64_t C, int dtype, dsw_stream_t stream);

/* Its backward (autograd of the torch.gather at layers.py:1067):  dX[b,v,f] = sum over { d : sel[b,d,f] == v } dY[b,d,f],
 * evaluated as a gather over the CSR of the TRANSPOSED matrix (rowptr_t [v_fine+1], colind_t): deterministic, no atomics. */
int dsw_maxval_pool_bwd(const int32_t* rowptr_t, const int32_t* colind_t, int64_t v_fine, int64_t v_coarse,
                        const void* dY, const int32_t* sel, void* dX, int64_t B, int64_t C, int dtype,
                        dsw_stream_t stream);

/* Max-value unpooling (layers.py:1082-1103, GeneralMaxValUnpool.forward: torch.index_put into zeros, no accumulate):
 *     Y[b,v,f] = X[b,d,f] for the largest d with sel[b,d,f] == v, else 0      (last write wins, as on the CPU)
 * X: [B, v_coarse, C]; Y: [B, v_fine, C]; workspace: dsw_maxval_unpool_workspace_bytes(B, v_fine, C) bytes. */
int64_t dsw_maxval_unpool_workspace_bytes(int64_t B, int64_t v_fine, int64_t C);
int dsw_maxval_unpool_fwd(const int32_t* sel, const void* X, void* Y, void* workspace, int64_t workspace_bytes,
                          int64_t B, int64_t v_coarse, int64_t v_fine, int64_t C, int dtype, dsw_stream_t stream);

/* Its backward (autograd of index_put):  dX[b,d,f] = dY[b, sel[b,d,f], f]. */
int dsw_maxval_unpool_bwd(const int32_t* sel, const void* dY, void* dX, int64_t B, int64_t v_coarse, int64_t v_fine,
                          int64_t C, int dtype, dsw_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DSW_HIP_H */
