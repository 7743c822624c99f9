/* dpm_hip.h -- C ABI of libdpm_hip.so: the MI355X (gfx950) kernels behind DeepPointMap's
 * encode -> match -> register hot path.
 *
 * The reference has no FFI layer of its own: its lower boundary is the string-keyed operator
 * tables `Sampler(method)` / `Querier(method)` (network/encoder/utils.py:21-28,129-133), the
 * third-party pytorch3d ops they dispatch to (utils.py:12,94,102,115,278-283;
 * system/modules/utils.py:10,80) and plain torch modules.  Each entry point below names the
 * reference interface it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller, the
 *    library never allocates, frees or retains it; scratch is passed in as `workspace`.
 *  - all tensors are fp32, dense, "point-major": xyz (B,N,3), features (B,N,C); index
 *    tensors are int32; `lengths[b]` = number of valid (leading) points of frame b.
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant and
 *    free of global mutable state, so one Decoder may be driven from several host threads
 *    (reference system/core.py:54-57,93-103).
 *  - return value: 0 = ok, <0 = invalid argument / unsupported shape, >0 = hipError_t.
 */
#ifndef DPM_HIP_H
#define DPM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *dpm_stream_t;

#define DPM_OK 0
#define DPM_EINVAL (-1)
#define DPM_EUNSUPPORTED (-2)

#define DPM_ACT_NONE 0
#define DPM_ACT_RELU 1
#define DPM_ACT_SIGMOID 2

/* library / ABI version (major*1000+minor) and a printable name for a status code.  The shipped library reads no
 * environment variable in any entry point.  A build with -DDPM_EXPERIMENT (measurement switches that skip work or swap
 * kernel layouts, read with getenv; scripts/ only) ORs DPM_VERSION_EXPERIMENT into the version so that callers can
 * refuse it: bench.py does. */
#define DPM_VERSION_EXPERIMENT 0x40000000
int dpm_version(void);
const char *dpm_error_string(int status);

/* Encoder.forward input staging (network/encoder/encoder.py:52): channel-first (B,C,N) points +
 * bool padding (B,N; nonzero = padded) -> xyz (B,N,3), lengths (B,) = count of valid points.
 * Valid points must be the leading ones, as the reference's FPS assumes (utils.py:255). */
int dpm_prepare_points(const float *points_cf, const uint8_t *padding, int B, int C, int N,
                       float *xyz, int32_t *lengths, dpm_stream_t stream);

/* (B,R,C) point-major -> (B,C,R) channel-first (the layout Encoder.forward returns). */
int dpm_to_channel_first(const float *x, int B, int R, int C, float *out, dpm_stream_t stream);
/* The same with the output rows ldo >= R floats apart (batch elements C * ldo apart).  The decoder stages the reference's
 * channel-first descriptors (B,131,M) (odometry.py:47-49) as token rows of 132 floats, so that the 128 feature columns of
 * a row start 16-byte aligned for the projection GEMM (descriptor_attention.py:24-30). */
int dpm_to_channel_first_ld(const float *x, int B, int R, int C, float *out, int ldo, dpm_stream_t stream);

/* The encoder's return triple [coor (B,3,S), feat (B,C,S), padding (B,S; 1 = padded)] (network/encoder/encoder.py:51-69)
 * from the point-major level (xyz (B,S,3), fea (B,S,C), lengths), and -- desc != NULL -- the unified descriptor
 * (B,C+3,S) = [feat ; coor * coor_scale] of ExtractionThread.process (system/modules/odometry.py:47-49), in one pass. */
int dpm_emit_descriptors(const float *xyz, const float *fea, const int32_t *lengths, int B, int S, int C,
                         double coor_scale, float *coor, float *feat, uint8_t *padding, float *desc, dpm_stream_t stream);

/* Lower sampling levels as prefixes of the first level's picks (the four lower Sampler.fps calls of
 * network/encoder/pointnext.py:38-47 when npoint is descending): for level i with K = npoint[i] (host array),
 * xyz_out / idx_out hold the levels back to back ((B,K,3) / (B,K) each), len_out is (n_levels,B). */
int dpm_nested_levels(const float *xyz0, const int32_t *len0, int B, int K0, int n_levels, const int32_t *npoint,
                      float *xyz_out, int32_t *idx_out, int32_t *len_out, dpm_stream_t stream);

/* out[p, m, c] = src[index[p] * frame_stride + m * ld + c], p < n, m < rows, c < cols: per-pair copies of per-frame
 * rows (what torch.index_select did for the pair lists of odometry.py:103-127). */
int dpm_gather_frames(const float *src, long long frame_stride, int rows, int ld, int cols, const int32_t *index, int n,
                      float *out, dpm_stream_t stream);

/* Sampler.fps / Sampler.fps_t3d == pytorch3d.ops.sample_farthest_points
 * (network/encoder/utils.py:210-285): start index 0, dist = (dx*dx+dy*dy)+dz*dz evaluated in
 * fp32 without contraction, next pick = first argmax.  idx (B,K) is -1 where lengths[b] < K;
 * new_xyz (B,K,3) is zero there (masked_gather, utils.py:298-343); new_lengths[b] = number of
 * idx >= 0.  workspace: dpm_fps_workspace_bytes(B,N,K) bytes. */
size_t dpm_fps_workspace_bytes(int B, int N, int K);
int dpm_fps(const float *xyz, const int32_t *lengths, int B, int N, int K, int32_t *idx,
            float *new_xyz, int32_t *new_lengths, void *workspace, dpm_stream_t stream);
/* same, with the algorithm forced: 0 = auto (1 up to 16 384 points, 5 up to 65 536, 1 again beyond), 1 = register / brute force,
 * 2 = bucket-pruned over Z-ordered grid cells (N <= 65 536), 5 = the bucket kernel over a Sort-Tile-Recursive packing
 * (16 384 < N <= 65 536).  All give identical bits; the tests run all of them.  3, 4, 6, 7 (speculative multi-pick rounds,
 * one-wave tree: exact but slower, removed in round 3) return DPM_EUNSUPPORTED. */
int dpm_fps_ex(const float *xyz, const int32_t *lengths, int B, int N, int K, int32_t *idx,
               float *new_xyz, int32_t *new_lengths, void *workspace, int algo, dpm_stream_t stream);
/* `random_start_point=True` (utils.py:248): frame b starts from point start[b] (clamped to its valid points) instead of
 * point 0; the caller draws the indices (the reference: random.randint(0, lengths[n] - 1), one draw per frame in batch
 * order).  Default algorithm choice (1 / 5 / 2 by N). */
int dpm_fps_start(const float *xyz, const int32_t *lengths, const int32_t *start, int B, int N, int K, int32_t *idx,
                  float *new_xyz, int32_t *new_lengths, void *workspace, dpm_stream_t stream);

/* Querier.hybrid_query / hybrid_query_t3d == pytorch3d.ops.knn_points + radius mask
 * (network/encoder/utils.py:76-89,113-123): for each centre the K nearest valid points, slots
 * whose squared distance exceeds radius^2 replaced by the nearest index.  Slot 0 is the nearest
 * point.  idx (B,S,K).  Distances and tie handling reproduce the reference's CPU path exactly
 * (see csrc/knn.hip).  workspace: dpm_knn_workspace_bytes(B,N) bytes (0 for small N; NULL selects
 * the brute-force path). */
size_t dpm_knn_workspace_bytes(int B, int N);
int dpm_knn_hybrid(const float *points, const int32_t *lengths, const float *centers, int B, int N,
                   int S, int K, double radius, int32_t *idx, void *workspace, dpm_stream_t stream);

/* Same query when the centres are a SUBSET of the points (SetAbstraction centres are FPS picks of `points`) and the
 * self-query of ALL points with the same radius and K has already been answered (the preceding LocalAggregation):
 * center_src (B,S) = index of centre s in `points` (the FPS index, -1 for padded centres), reuse_idx (B,N,K) = that
 * earlier answer.  Rows with center_src >= 0 are copied (they are the identical computation), padded ones computed. */
int dpm_knn_hybrid_reuse(const float *points, const int32_t *lengths, const float *centers, int B, int N,
                         int S, int K, double radius, int32_t *idx, void *workspace, const int32_t *reuse_idx,
                         const int32_t *center_src, dpm_stream_t stream);

/* dpm_knn_hybrid in two calls sharing one workspace (N >= 1024, the grid path): the grid depends on the points and
 * the radius only, so a caller that pipelines frames builds it while the frames are still being sampled and runs the
 * search when the centres exist.  One search per build (the build resets the tie queue the search fills). */
int dpm_knn_build_grid(const float *points, const int32_t *lengths, int B, int N, double radius, void *workspace,
                       dpm_stream_t stream);
int dpm_knn_hybrid_prebuilt(const float *points, const int32_t *lengths, const float *centers, int B, int N,
                            int S, int K, double radius, int32_t *idx, void *workspace, dpm_stream_t stream);

/* Querier.ball_query / ball_query_t3d == pytorch3d.ops.ball_query (utils.py:57-73,99-110): the K
 * smallest indices among the valid points within `radius` (expanded-form distance, like the
 * reference), ascending, padded with the first of them.  idx (B,S,K). */
int dpm_ball_query(const float *points, const int32_t *lengths, const float *centers, int B, int N, int S,
                   int K, double radius, int32_t *idx, dpm_stream_t stream);

/* Sampler.voxel (network/encoder/utils.py:150-207): per frame, the point nearest the centre of every occupied voxel
 * (smallest index among equals) among the points within `sample_range` of the origin, in ascending voxel-id order; if
 * more than K voxels are occupied, the K most populated ones in the order torch.topk(counts, K) returns them (its CPU
 * kernel's choice and order among equal counts, replayed).  points (B,N,D) with D >= 3 (xyz first), padding (B,N)
 * uint8 (1 = padded: moved to 2*sample_range before the bounding box is taken, as the reference does).
 * Two calls, because the grid size is data dependent:
 *   dpm_voxel_sampler_bounds  -> hdr (B,8) floats: xyz_min[3], X, Y, Z (integer-valued, utils.py:159-161), 0, 0;
 *                                the caller reads X*Y*Z back and sizes the workspace for max_cells >= max_b X*Y*Z;
 *   dpm_voxel_sampler_select  -> sel (B,cap) original point indices in output order, -1 = padding rows
 *                                (utils.py:192-198), and n_unique (B) = occupied voxels.  K >= 1: cap == K;
 *                                K < 0 (the reference's K=None, B == 1 there): cap == N, all voxels.
 *                                A frame whose grid exceeds max_cells (or is not finite) gets hdr[6] = 1 and no points.
 *                                hdr[7] = 1 marks a frame where two points of a voxel were exactly equally near its
 *                                centre: the reference's pick then follows its unstable torch.sort, which one thread
 *                                replays over the frame's N distances (exact, slow: lattices / duplicated points only).
 * N <= 2^24.  workspace: dpm_voxel_sampler_workspace_bytes(B, N, max_cells). */
int dpm_voxel_sampler_bounds(const float *points, const uint8_t *padding, int B, int N, int D, double voxel_size,
                             double sample_range, float *hdr, dpm_stream_t stream);
size_t dpm_voxel_sampler_workspace_bytes(int B, int N, long long max_cells);
int dpm_voxel_sampler_select(const float *points, const uint8_t *padding, int B, int N, int D, double voxel_size,
                             double sample_range, float *hdr, long long max_cells, int K, int32_t *sel, int cap,
                             int32_t *n_unique, void *workspace, dpm_stream_t stream);

/* ---------------------------------------------------------------- pose graph (host) ---- */

/* HOST function (no GPU involved).  PoseGraph.__optim_open3d (system/modules/pose_graph.py:565-613):
 * open3d.pipelines.registration.global_optimization(graph, GlobalOptimizationLevenbergMarquardt(),
 * GlobalOptimizationConvergenceCriteria(), GlobalOptimizationOption(edge_prune_threshold=0, preference_loop_closure=2,
 * reference_node)) on a graph whose edges are all certain (pose_graph.py:597) -- a restatement of open3d 0.16's published
 * algorithm (parity unpinned: open3d is absent; see deeppointmap_amd/posegraph_optim.py).
 * poses (n,4,4) row-major doubles (scan -> world); edge k: src[k] -> dst[k], X (E,4,4) = the o3d PoseGraphEdge
 * transformation (source-scan coordinates into the target scan's frame), info (E,6,6) rotation block first.
 * criteria: NULL = open3d's defaults, else 8 doubles {max_iteration, min_relative_increment,
 * min_relative_residual_increment, min_right_term, min_residual, max_iteration_lm, upper_scale_factor,
 * lower_scale_factor}.  out_poses (n,4,4): refined poses, node `reference_node` unchanged.  stats (6): iterations,
 * residual at the start, residual at the end -- of the first and of the second Levenberg-Marquardt pass. */
int dpm_posegraph_optimize(const double *poses, int n, const int32_t *src, const int32_t *dst, const double *X,
                           const double *info, int E, int reference_node, const double *criteria, double *out_poses,
                           double *stats);

/* HOST function (no GPU involved): torch.topk(values, k, largest, sorted=True) on one row of n floats without NaNs, by
 * the step-by-step replay of its CPU kernel that the device code uses (csrc/topk_emulate.h, same source compiled for
 * the host) -- which elements survive and in which order among equal values.  out_idx (k).  Exists so that the replay
 * can be held to torch.topk by tests that run without a GPU. */
int dpm_host_topk_replay(const float *values, int n, int k, int largest, int32_t *out_idx);
/* Same for torch.sort(values, descending, stable=False) -> out_idx (n): std::sort over (value, index) pairs. */
int dpm_host_sort_replay(const float *values, int n, int descending, int32_t *out_idx);

/* SetAbstraction / LocalAggregation body (network/encoder/pointnext.py:52-61,97-107):
 * out[b,s,:] = max_k relu(LN(W [fea[idx[b,s,k]], (xyz[idx]-center)/radius] + bias)).
 * W (Cout, Cin+3) row-major exactly as the Conv2d weight (Cout,Cin+3,1,1): first Cin columns
 * features, last 3 relative xyz.  LN: eps 1e-5, biased variance, affine (gamma, beta).
 * fp32 MFMA for Cout in {32,64,128,256,512} and K in {16,32} (every shipped layer); the _generic
 * entry is the plain-VALU kernel for any other shape (same contract). */
int dpm_group_mlp_max_generic(const float *xyz, const float *fea, const float *centers,
                              const int32_t *idx, const float *W, const float *bias, const float *gamma,
                              const float *beta, int B, int N, int S, int K, int Cin, int Cout, double radius,
                              float *out, dpm_stream_t stream);
int dpm_group_mlp_max(const float *xyz, const float *fea, const float *centers, const int32_t *idx,
                      const float *W, const float *bias, const float *gamma, const float *beta,
                      int B, int N, int S, int K, int Cin, int Cout, double radius, float *out,
                      dpm_stream_t stream);

/* First-stage SetAbstraction with Encoder.point_mlp0 (encoder.py:25,53) folded in: the input
 * features are W0 xyz + b0 (W0 (Cin,3) = the Conv1d weight, b0 (Cin)), evaluated inside the
 * gather, so the (B,N,Cin) level-0 feature tensor is never written or read.  Cout in {32,64,128},
 * K in {16,32}; otherwise DPM_EUNSUPPORTED (callers then materialise the features with dpm_linear). */
int dpm_group_mlp_max_from_xyz(const float *xyz, const float *W0, const float *b0, const float *centers,
                               const int32_t *idx, const float *W, const float *bias, const float *gamma,
                               const float *beta, int B, int N, int S, int K, int Cin, int Cout,
                               double radius, float *out, dpm_stream_t stream);

/* The same layer with the projection hoisted out of the gather ("project before gather", the encoder's default):
 * W [fea_n ; rel] + b = (W_f fea_n + b) + W_r rel, so P = fea W_f^T + b (B,N,Cout) is computed once per POINT with
 * dpm_linear (x = fea, W = the first Cin columns, ldw = Cin+3) and this kernel does the per-(centre, neighbour) rest:
 * out[b,s,:] = max_k relu(LN(P[idx[b,s,k]] + W_rel (xyz[idx]-center)/radius)).  W_rel points at column Cin of the
 * Conv2d weight, ldw_rel = Cin+3.  Cout in {32,64,128,256,512}; P and out 16-byte aligned.
 * dpm_group_affine_ln_max: first-stage variant, P is affine in the point itself (P = A xyz + c with A (Cout,3) =
 * W_f W0 and c (Cout) = W_f b0 + b, both produced with dpm_linear) and is evaluated on the fly; Cout in {32,64,128}. */
int dpm_group_gather_ln_max(const float *P, const float *xyz, const float *centers, const int32_t *idx,
                            const float *W_rel, int ldw_rel, const float *gamma, const float *beta, int B, int N,
                            int S, int K, int Cout, double radius, float *out, dpm_stream_t stream);
int dpm_group_affine_ln_max(const float *A, const float *cvec, const float *xyz, const float *centers,
                            const int32_t *idx, const float *W_rel, int ldw_rel, const float *gamma,
                            const float *beta, int B, int N, int S, int K, int Cout, double radius, float *out,
                            dpm_stream_t stream);
/* The folded form of dpm_group_gather_ln_max (round 5): the relative-coordinate term W_rel (p - c) / radius is linear in the point
 * and in the centre separately, so P' = P + xyz (W_rel / radius)^T is made once per point by the projection's epilogue
 * (dpm_linear_bf16x3_rank3) and this kernel computes out[b,s,:] = max_k relu(LN(P'[idx[b,s,k]] - (W_rel / radius) center)):
 * four subtractions instead of fifteen operations per gathered row and lane, no coordinates gathered.  Same shapes and alignment
 * as dpm_group_gather_ln_max.  |W_rel p / radius| exceeds the term it replaces by up to |p| / radius, so the pre-LayerNorm values
 * carry ~1e-6 relative rounding error instead of ~1e-7 (DESIGN.md section 4 has the measured effect on descriptors and poses).
 * dpm_group_affine_ln_max folds the same way internally (A + W_rel / radius). */
int dpm_group_gather_ln_max_folded(const float *P, const float *centers, const int32_t *idx, const float *W_rel, int ldw_rel,
                                   const float *gamma, const float *beta, int B, int N, int S, int K, int Cout, double radius,
                                   float *out, dpm_stream_t stream);
/* The folded forms for a layer whose LayerNorm mean removal has been moved into its weights (round 5): the caller passes
 * W' = (I - 11^T / Cout) W -- every column of [W_f | W_rel], the bias, and for the affine form the columns of A and the vector
 * cvec, have zero mean over the Cout output channels -- so that every pre-LayerNorm row (network/encoder/pointnext.py:52-61: the
 * grouped features after the 1x1 Conv2d) has zero mean by construction and the kernels compute the variance from the rows as they
 * are (no row sum, no subtraction: about a third of the per-row instructions).  The caller THEN multiplies channel c of the layer
 * (its row of [W_f | W_rel], bias_c; A, cvec) by sign(gamma_c) (+1 for 0): gamma y + beta = |gamma| (sign(gamma) y) + beta is then
 * non-decreasing in the gathered value and commutes with the maximum over the neighbours bit for bit, so |gamma|, beta and the ReLU
 * are applied once per centre (gamma is passed as stored; the kernels take its magnitude; the signs leave the sum of squares, all the
 * variance needs, as it was).  tests/test_centred_algebra.py states the identities in fp64.  PRECONDITIONS, not checked: with other
 * weights the result is a LayerNorm without its mean removal / with |gamma| for gamma.  Otherwise as dpm_group_gather_ln_max_folded / dpm_group_affine_ln_max. */
int dpm_group_gather_ln_max_centred(const float *P, const float *centers, const int32_t *idx, const float *W_rel, int ldw_rel,
                                    const float *gamma, const float *beta, int B, int N, int S, int K, int Cout, double radius,
                                    float *out, dpm_stream_t stream);
int dpm_group_affine_ln_max_centred(const float *A, const float *cvec, const float *xyz, const float *centers,
                                    const int32_t *idx, const float *W_rel, int ldw_rel, const float *gamma,
                                    const float *beta, int B, int N, int S, int K, int Cout, double radius, float *out,
                                    dpm_stream_t stream);

/* 1x1 Conv1d / nn.Linear (build_mlp, network/encoder/utils.py:358-389; decoder heads):
 * out[r, :Cout] = act(x[r,:Cin] W^T + bias + residual[r]); W (Cout,Cin) row-major with leading
 * dimension ldw; x/out/residual have leading dimensions ldx/ldo/ldr (rows R). bias, residual
 * may be NULL. */
int dpm_linear(const float *x, int ldx, const float *W, int ldw, const float *bias,
               const float *residual, int ldr, float *out, int ldo, int R, int Cin, int Cout, int act,
               dpm_stream_t stream);
/* same for `batch` independent problems: operand b lives at ptr + b*stride (strides in floats; a
 * stride of 0 shares the operand).  With W = the second descriptor set this is the M x N similarity
 * contraction of Decoder._descriptor_pairing (decoder.py:185).  fp32 MFMA (exact fp32). */
int dpm_linear_batched(const float *x, int ldx, long long sx, const float *W, int ldw, long long sw,
                       const float *bias, const float *residual, int ldr, long long sr, float *out,
                       int ldo, long long so, int batch, int R, int Cin, int Cout, int act,
                       dpm_stream_t stream);

/* LayerNorm1d / nn.LayerNorm over the channel axis (network/encoder/utils.py:392-402,
 * descriptor_attention.py:20-22): out = act(LN(x + pre)*gamma + beta + post); pre/post NULL-able,
 * all (R,C) with leading dimension = C except x (ldx) and out (ldo). */
int dpm_layernorm(const float *x, int ldx, const float *pre, const float *gamma, const float *beta,
                  const float *post, float *out, int ldo, int R, int C, int act, dpm_stream_t stream);

/* Conv1d(k=1)/nn.Linear followed by LayerNorm1d (network/encoder/utils.py:358-413 build_mlp; the post-norm blocks of
 * network/decoder/descriptor_attention.py:31-48) as ONE kernel: out = act(LN(x W^T + bias + pre) * gamma + beta + post),
 * eps 1e-5, biased variance.  Cout in {32,64,128,256}; pre / post packed (R,Cout) or NULL.  Returns DPM_EUNSUPPORTED
 * for other widths / unaligned operands (run dpm_linear + dpm_layernorm then). */
int dpm_linear_layernorm(const float *x, int ldx, const float *W, int ldw, const float *bias, const float *pre,
                         const float *gamma, const float *beta, const float *post, float *out, int ldo, int R, int Cin,
                         int Cout, int act, dpm_stream_t stream);

/* FeaturePropagation interpolation (network/encoder/pointnext.py:199-216): for each fine point
 * the 3 nearest valid coarse points (expanded-form distance), w_j = (1/max(d_j,1e-8))/sum;
 * out[b,n,:] = cat[fea1[b,n,:D1], sum_j w_j fea2[b,idx_j,:D2]].  S==1 broadcasts fea2. */
int dpm_three_interp_cat(const float *xyz1, const float *xyz2, const int32_t *lengths2,
                         const float *fea1, const float *fea2, int B, int N, int S, int D1, int D2,
                         float *out, dpm_stream_t stream);

/* ---------------------------------------------------------------- decoder -------------- */

/* PositionEmbeddingCoordsSine.forward (network/decoder/descriptor_attention.py:66-83):
 * xyz rows (R, leading dim ld, metres) -> out (R,E); dim_t (F) is the reference's table
 * temperature**(2*(i//2)/F); channels >= 3F are zero. */
int dpm_posemb(const float *xyz, int ld, const float *dim_t, int F, int E, int R, float *out,
               dpm_stream_t stream);

/* Scaled-dot-product core of nn.MultiheadAttention (descriptor_attention.py:14-15,35-45):
 * out[b,m,h*d:(h+1)*d] = softmax(Q_h K_h^T / sqrt(d)) V_h; no masks, dropout 0.  head_dim 32 (every shipped config:
 * model_channel 256, 8 heads) runs on the matrix cores; 8, 16, 64 and 128 (other Decoder(args)) through a generic kernel;
 * other widths return DPM_EUNSUPPORTED.  At head_dim 32 the score product Q K^T is computed on the bf16 matrix pipe from exact
 * three-way bf16 splits of both operands (six term products, fp32 accumulate: fp32-accumulation accuracy; dpm_linear_bf16x3
 * below has the arithmetic), the product with V in exact fp32 -- in every dpm_attention* entry point alike.
 * Q/K/V/out: row leading dims ld*, batch strides s* (in floats). */
int dpm_attention(const float *Q, int ldq, long long sq, const float *K, int ldk, long long sk,
                  const float *V, int ldv, long long sv, float *out, int ldo, long long so, int B, int M,
                  int N, int heads, int head_dim, dpm_stream_t stream);
/* The same with batch element b reading the keys / values of element (b + kv_shift) mod B: the source and target
 * tokens of P pairs stacked as B = 2P sequences with kv_shift = P make both directions of DescriptorAttentionLayer's
 * cross attention (descriptor_attention.py:41-44) one launch. */
int dpm_attention_shifted(const float *Q, int ldq, long long sq, const float *K, int ldk, long long sk,
                          const float *V, int ldv, long long sv, float *out, int ldo, long long so, int B, int M,
                          int N, int heads, int head_dim, int kv_shift, dpm_stream_t stream);

/* ... and with nn.MultiheadAttention's key_padding_mask (descriptor_attention.py:33-42): key_mask (B,N) bytes, non-zero
 * = key n of sequence b is padding and takes no part in the softmax (sequence b reads row (b + kv_shift) mod B of the
 * mask, like its keys); NULL = no mask. */
int dpm_attention_masked(const float *Q, int ldq, long long sq, const float *K, int ldk, long long sk,
                         const float *V, int ldv, long long sv, float *out, int ldo, long long so, int B, int M,
                         int N, int heads, int head_dim, int kv_shift, const uint8_t *key_mask, dpm_stream_t stream);

/* dpm_attention_shifted over batch elements DRAWN from a smaller set of stored sequences: element b's queries are stored
 * sequence seq_index[b] (Q + seq_index[b] * sq), its keys / values stored sequence seq_index[(b + kv_shift) mod B]
 * (K / V + ... * sk / sv); out is per batch element.  The consecutive-frame registrations of a batch (odometry.py:103-127)
 * use every frame as a source and as a target, and the q | k | v projection of the first cross-attention block
 * (descriptor_attention.py:41-44) depends on the frame alone: it is computed once per frame and attended through this
 * entry point -- row-wise kernels give the same rows whatever the row count, so the result equals the per-pair form bit
 * for bit.  head_dim 32 only; seq_index (B) int32 on the device. */
int dpm_attention_indexed(const float *Q, int ldq, long long sq, const float *K, int ldk, long long sk,
                          const float *V, int ldv, long long sv, float *out, int ldo, long long so, int B, int M,
                          int N, int heads, int head_dim, int kv_shift, const int32_t *seq_index, dpm_stream_t stream);

/* Key-split form of dpm_attention_shifted for FEW queries against MANY keys (scan-to-map registration: 256 scan tokens
 * attending a 4096-token map tile, mapping.py:153-155 -> descriptor_attention.py:41-44): the keys are cut into `nsplit`
 * ranges of whole 64-key tiles that run as separate workgroups, and a second kernel joins the ranges (rescaled to the
 * common maximum, fixed order).  Same result up to the rounding of that join.  2 <= nsplit <= 64, every range non-empty;
 * head_dim 32 only; workspace: dpm_attention_split_workspace_bytes(B, M, heads, head_dim, nsplit). */
size_t dpm_attention_split_workspace_bytes(int B, int M, int heads, int head_dim, int nsplit);
int dpm_attention_split(const float *Q, int ldq, long long sq, const float *K, int ldk, long long sk, const float *V,
                        int ldv, long long sv, float *out, int ldo, long long so, int B, int M, int N, int heads,
                        int head_dim, int kv_shift, int nsplit, void *workspace, dpm_stream_t stream);

/* nn.MultiheadAttention's in_proj followed by its attention (descriptor_attention.py:33-44) with K and V handed over as the
 * attention kernel's operand planes instead of fp32 rows (round 5).  dpm_linear_bf16x3_kvplanes is dpm_linear_bf16x3 for a
 * q | k | v projection over sequences of `tokens` rows: columns [0, kv_col0) go to `out` as fp32 rows (Q), columns [kv_col0, Cout) =
 * K (heads x 32) then V (heads x 32) go to kv_planes as one 24 576-byte image per (sequence = row / tokens, head, 64-key tile):
 * the three bf16 planes of the K tile in the swizzled rows the score product reads, then the three planes of the V tile
 * transposed in the order the P V product reads -- exactly what the attention kernel's own staging makes of the fp32 rows, so
 * dpm_attention_planes returns dpm_attention_shifted / _indexed's result bit for bit while each K / V element is split once
 * instead of once per 64-query block that reads it.  kv_planes: dpm_attention_planes_bytes(sequences, tokens, heads) bytes,
 * 16-byte aligned.  Needs tokens % 64 == 0, R % tokens == 0, kv_col0 % 64 == 0, Cout - kv_col0 == 64 * heads, 16-byte aligned
 * x / bias / out with ldx % 4 == 0; DPM_EUNSUPPORTED otherwise (callers then run dpm_linear_bf16x3 + dpm_attention_*).
 * dpm_attention_planes: head_dim 32, N % 64 == 0, no key mask; seq_index as in dpm_attention_indexed or NULL. */
size_t dpm_attention_planes_bytes(int n_sequences, int N, int heads);
int dpm_linear_bf16x3_kvplanes(const float *x, int ldx, const void *w_planes, int ldw, long long plane_stride,
                               const float *bias, float *out, int ldo, int R, int Cin, int Cout, int kv_col0, int tokens,
                               int heads, void *kv_planes, dpm_stream_t stream);
int dpm_attention_planes(const float *Q, int ldq, long long sq, const void *kv_planes, float *out, int ldo, long long so,
                         int B, int M, int N, int heads, int kv_shift, const int32_t *seq_index, dpm_stream_t stream);

/* The same contraction as dpm_linear (Conv1d(k=1) / nn.Linear: network/encoder/utils.py:358-389, the decoder's projections
 * and heads) on the bf16 matrix pipe with every fp32 operand split exactly into three bf16 terms and six of the nine term
 * products accumulated in fp32 (the three dropped ones are below 2^-23 of the product): fp32-accumulation accuracy at 3/8 of
 * the matrix-pipe time of the exact-fp32 instruction (csrc/gemm_b3.hip).  dpm_split_bf16x3 makes the weight planes once per
 * weight version: planes = 3 x n bf16 (hi | mid | lo), plane p of element i at planes[p * n + i].  dpm_linear_bf16x3 takes a row
 * block of such planes: w_planes -> plane 0 of the first weight row, rows ldw elements apart, planes plane_stride elements
 * apart.  DPM_EUNSUPPORTED for Cin % 32 != 0, Cout % 4 != 0 or unaligned bias / residual / output (x may have any row stride);
 * a caller that falls back to dpm_linear then must do so for EVERY call of that layer (the two kernels differ in the last bits). */
int dpm_split_bf16x3(const float *W, long long n, void *planes, dpm_stream_t stream);
int dpm_linear_bf16x3(const float *x, int ldx, const void *w_planes, int ldw, long long plane_stride, const float *bias,
                      const float *residual, int ldr, float *out, int ldo, int R, int Cin, int Cout, int act,
                      dpm_stream_t stream);
/* dpm_linear_bf16x3 plus a rank-3 term added in fp32 in the epilogue, before residual and activation:
 * out[r, c] += scale * (x3[r, 0:3] . w3[c, 0:3]), x3 (R,3) packed, w3 rows ldw3 floats apart.  It is the POINT half of a grouping
 * layer's relative-coordinate columns (network/encoder/pointnext.py:52-56: W [fea ; (p - c) / r] = W_f fea + W_r p / r - W_r c / r),
 * with x3 = the points' coordinates, w3 = W_r, scale = 1 / r; dpm_group_gather_ln_max_folded subtracts the centre half. */
int dpm_linear_bf16x3_rank3(const float *x, int ldx, const void *w_planes, int ldw, long long plane_stride, const float *bias,
                            const float *residual, int ldr, float *out, int ldo, int R, int Cin, int Cout, int act,
                            const float *x3, const float *w3, int ldw3, double scale, dpm_stream_t stream);
/* dpm_linear_layernorm (Conv1d(k=1) / Linear + LayerNorm1d, network/encoder/utils.py:358-413, descriptor_attention.py:36-48)
 * on the bf16x3 product, weights as planes of dpm_split_bf16x3; rows identical to dpm_linear_bf16x3 followed by dpm_layernorm.
 * DPM_EUNSUPPORTED for Cout outside {32, 64, 128, 256}, Cin % 32 != 0 or unaligned operands. */
int dpm_linear_layernorm_bf16x3(const float *x, int ldx, const void *w_planes, int ldw, long long plane_stride,
                                const float *bias, const float *pre, const float *gamma, const float *beta, const float *post,
                                float *out, int ldo, int R, int Cin, int Cout, int act, dpm_stream_t stream);

/* InvResMLP's point-wise pair (network/encoder/pointnext.py:118-138: pw_conv = Conv1d(C, 4C, 1) -> LayerNorm1d -> ReLU ->
 * Conv1d(4C, C, 1) -> LayerNorm1d, then the residual and ReLU) as ONE kernel for C = 32, H = 4C = 128 (the first level, whose
 * 4C-wide intermediate is otherwise written to and read back from HBM): out = relu(LN2(relu(LN1(x W1^T + b1)) W2^T + b2) + post).
 * x (R,32) rows ldx floats apart; post / out packed (R,32); W1 (128,32) and W2 (32,128) as bf16x3 planes of dpm_split_bf16x3
 * (plane p of element i at planes[p * plane_stride + i]), W2's COLUMNS permuted before the split: stored column 32 s + 8 g + e
 * holds original column 16 (2 s + (e >> 2)) + 4 g + (e & 3), s < 4, g < 4, e < 8 (the order in which the kernel's accumulator
 * registers hold the intermediate).  Same bf16x3 arithmetic as dpm_linear_layernorm_bf16x3 twice; the row statistics are summed
 * in another association (results agree to rounding).  DPM_EUNSUPPORTED for other widths or unaligned operands. */
int dpm_pwconv_pair_bf16x3(const float *x, int ldx, const void *w1_planes, long long plane_stride1, const float *b1,
                           const float *gamma1, const float *beta1, const void *w2_planes_kperm, long long plane_stride2,
                           const float *b2, const float *gamma2, const float *beta2, const float *post, float *out, int R,
                           int C, int H, dpm_stream_t stream);

/* F.normalize(x, p=2, dim=-1) (decoder.py:185): x / max(||x||, 1e-12), rows (R,C). */
int dpm_l2_normalize(const float *x, int R, int C, float *out, dpm_stream_t stream);

/* Decoder._descriptor_pairing from the L2-normalised head outputs on (decoder.py:185-191) as ONE operator: a (batch,M,C),
 * b (batch,N,C) row-major -> S = a b^T (fp32 MFMA), P = softmax_row(S/tau) * softmax_col(S/tau), the k largest entries of
 * the flattened P sorted descending (ties: smaller flat index first): out_val (batch,k), out_idx (batch,k) with
 * row = idx / N, col = idx % N.  The M x N matrix never exists in memory: two launches over row strips of 64, the strip
 * recomputed in the second (csrc/match.hip).  DPM_EUNSUPPORTED for N > 256, k > 2048, C % 32 != 0 or unaligned operands:
 * the caller then runs dpm_linear_batched + dpm_dual_softmax_topk (same values up to the last bit of the column sums). */
size_t dpm_match_workspace_bytes(int batch, int M, int N, int k);
int dpm_match_topk(const float *a, const float *b, int batch, int M, int N, int C, double tau, int k, float *out_val,
                   int32_t *out_idx, void *workspace, dpm_stream_t stream);

/* Decoder._descriptor_pairing tail (decoder.py:186-191): S (M,N) similarity, overwritten with
 * P = softmax_row(S/tau) * softmax_col(S/tau); then the k largest entries of the flattened P,
 * sorted descending: out_val (k), out_idx (k) flat indices (row = idx / N, col = idx % N).
 * `batch` independent (M,N) problems are laid out back to back (S (batch,M,N), outputs (batch,k)). */
size_t dpm_pairing_workspace_bytes(int batch, int M, int N);
int dpm_dual_softmax_topk(float *S, int batch, int M, int N, double tau, int k, float *out_val,
                          int32_t *out_idx, void *workspace, dpm_stream_t stream);

/* Decoder._get_corres_sets input assembly (decoder.py:204-205): for the k flat indices,
 * X[0:k] = [x[src] | y[dst]], X[k:2k] = [y[dst] | x[src]] (rows of 2E), and the decoded
 * src_idx/dst_idx (k). x (batch,M,E), y (batch,N,E), X (batch,2k,2E). */
int dpm_gather_pairs(const float *x, const float *y, const int32_t *flat_idx, int batch, int k, int M,
                     int N, int E, float *X, int32_t *src_idx, int32_t *dst_idx, dpm_stream_t stream);

/* torch.mean over the points of each batch element (OverlapHead, heads.py:64-65):
 * x (B,R,C) -> out[b, 0:C] with row stride ldo. */
int dpm_mean_rows(const float *x, int B, int R, int C, float *out, int ldo, dpm_stream_t stream);

/* Decoder._get_corres_sets + _solve_transformation_SVD (decoder.py:208-265): offsets (2k,3)
 * [first k: src->dst, next k: dst->src], keypoint coordinates, pair indices and confidences ->
 * result[0:9] R row-major, [9:12] T, [12] rmse, [13] #correspondences, [14] #inliers,
 * [15] iterations, [16] mean of the first 30 inlier confidences (simvec_to_num,
 * system/modules/utils.py:18), [17:20] reserved, [20:20+#inliers] inlier confidences in
 * correspondence order.  result holds 20 + 2k floats; `header` (NULL-able) receives a copy of
 * result[0:20].  R = V U^T of the fp64 SVD, no reflection fix.
 * offsets == NULL: src_xyz/dst_xyz/conf are taken as k ready-made correspondences (rows) and
 * only _solve_transformation_SVD runs.  `batch` independent pairs: offsets (batch,2k,3), indices
 * and conf (batch,k), coordinates at src_xyz + b*stride_src, result (batch, 20+2k), header rows
 * header_stride floats apart.  Limit: the 2k weights and the torch.topk replay scratch live in LDS (24 B per pair next to
 * the kernel's static arrays, together at most the CU's 160 KB: k up to about 6000 on gfx950); beyond that
 * DPM_EUNSUPPORTED. */
size_t dpm_kabsch_workspace_bytes(int batch, int k);
int dpm_corr_kabsch(const float *offsets, const float *src_xyz, int ld_src, long long stride_src,
                    const float *dst_xyz, int ld_dst, long long stride_dst, const int32_t *src_idx,
                    const int32_t *dst_idx, const float *conf, int batch, int k, double eps_offset,
                    int num_iter, double std_ratio, void *workspace, float *result, float *header,
                    int header_stride, dpm_stream_t stream);

/* ---------------------------------------------------------------- scan pre-processing ---- */

/* VoxelSample(voxel_size, 'first') -> DistanceSample(min_dis, max_dis) -> CoordinatesNormalization(ratio)
 * (dataloader/transforms.py:322-356,387-397,400-407): xyz = N raw points, `stride` floats apart (3 for packed
 * xyz, 4 for KITTI .bin records).  Output: the kept points in ascending voxel-id order, divided by ratio
 * (out_xyz (out_capacity,3)), their original indices (out_idx, NULL-able), status[0] = number kept,
 * status[1] = 1 if the voxel grid (X*Y*Z cells) exceeded max_cells (nothing is written then).
 * workspace: dpm_preprocess_workspace_bytes(max_cells). */
size_t dpm_preprocess_workspace_bytes(long long max_cells);
int dpm_preprocess_scan(const float *xyz, int N, int stride, double voxel_size, double min_dis, double max_dis,
                        double ratio, long long max_cells, float *out_xyz, int32_t *out_idx, int out_capacity,
                        int32_t *status, void *workspace, dpm_stream_t stream);

/* Building blocks of OutlierFilter / LowPassFilter (dataloader/transforms.py:230-289), which the reference runs
 * through pytorch3d.knn_points and open3d.estimate_normals between DistanceSample and CoordinatesNormalization:
 *
 * dpm_knn_self: exact K nearest OTHER points of every point of one cloud xyz (N,3) (== knn_points(p, p, K+1)
 *   with column 0 dropped; rows ordered by (distance, index); direct-form squared distances).  Any of idx (N,K),
 *   dist2 (N,K), mean_dist (N) [mean over the K columns of sqrt(dist2), transforms.py:240-241] may be NULL.
 *   `cell` = edge of the search grid in the units of xyz (a few times the typical neighbour spacing).
 * dpm_point_normals: unit normal of every point = eigenvector of the smallest eigenvalue of the covariance of
 *   the points within `radius` (itself included), (0,0,1) when fewer than 3 are in range (transforms.py:268-271).
 * dpm_lowpass_similarity: sim[i] = sum of the `flux` largest |n_i . n_j| over the K neighbours idx[i,:]
 *   (transforms.py:279-281).
 * dpm_stat_filter: mean / unbiased std of stat (N); mode 0 keeps stat <= mean + k_std*std (OutlierFilter,
 *   transforms.py:242-246), mode 1 keeps stat > mean - k_std*std (LowPassFilter, transforms.py:282); survivors
 *   are compacted in order into xyz_out (coordinates divided by `ratio`: CoordinatesNormalization folded into the
 *   last filter, 1.0 = untouched) / idx_out (idx_in NULL: positions), their number into n_out[0].
 * workspace for the first two: dpm_knn_self_workspace_bytes(N). */
size_t dpm_knn_self_workspace_bytes(int N);
int dpm_knn_self(const float *xyz, int N, int K, double cell, int32_t *idx, float *dist2, float *mean_dist,
                 void *workspace, dpm_stream_t stream);
int dpm_point_normals(const float *xyz, int N, double radius, float *normals, void *workspace, dpm_stream_t stream);
int dpm_lowpass_similarity(const float *normals, const int32_t *idx, int N, int K, int flux, float *sim,
                           dpm_stream_t stream);
int dpm_stat_filter(const float *stat, int N, double k_std, int mode, double ratio, const float *xyz_in,
                    const int32_t *idx_in, float *xyz_out, int32_t *idx_out, int32_t *n_out, dpm_stream_t stream);

/* ---------------------------------------------------------------- map tiles ------------- */

/* PoseGraph.__global_mapping + centring of global_map_query_graph (system/modules/pose_graph.py:373-409,
 * 504-510): key_points (n_scans,C,S) device-resident unified descriptors (last three rows xyz in metres),
 * select (K) scan indices in tile order (NULL: 0..K-1), poses (n_scans,12) [R row-major, T] = SE3_pred,
 * centering (12) -> out (C, K*S): features copied, xyz -> R_c^T ((R_k x + t_k) - t_c). */
int dpm_map_tile(const float *key_points, const int32_t *select, const float *poses, const float *centering,
                 int C, int S, int K, float *out, dpm_stream_t stream);

/* ---------------------------------------------------------------- registration edge ---- */

/* calculate_information_matrix_from_pcd (system/modules/utils.py:60-113), pytorch3d branch:
 * pcd1 (3,N1), pcd2 (3,N2) channel-first metres; Rt = 12 floats (R row-major, then T);
 * out6x6 = sum over source points whose transformed nearest target lies within `radius` of the
 * G^T G of that target point.  Exact nearest neighbours via a uniform grid. */
size_t dpm_infomat_workspace_bytes(int n_pairs, int N1, int N2);
/* Rt may point into a dpm_corr_kabsch result (its first 12 floats are R row-major, T). */
int dpm_information_matrix(const float *pcd1, int N1, const float *pcd2, int N2, const float *Rt,
                           double radius, float *out6x6, void *workspace, dpm_stream_t stream);
/* n_pairs edges in one pass: pcd (F,3,N) scans in metres, pair p = (src_frame[p], dst_frame[p]);
 * pose p at Rt + p*rt_stride (12 floats), output p at out + p*out_stride (36 floats). */
int dpm_information_matrix_batched(const float *pcd, int N, const int32_t *src_frame,
                                   const int32_t *dst_frame, int n_pairs, const float *Rt, int rt_stride,
                                   double radius, float *out, int out_stride, void *workspace,
                                   dpm_stream_t stream);
/* The same computation in two calls sharing one workspace (dpm_infomat_workspace_bytes(n_pairs, N, N)):
 * the target grids depend only on the scans, so a caller that pipelines frames builds them before the
 * poses exist (next to the encoder) and runs only the search after the registration
 * (system/modules/odometry.py:116-118 calls the function right after registration_forward). */
int dpm_infomat_build_grids(const float *pcd, int N, const int32_t *dst_frame, int n_pairs, double radius,
                            void *workspace, dpm_stream_t stream);
int dpm_infomat_search_grids(const float *pcd, int N, const int32_t *src_frame, const int32_t *dst_frame,
                             int n_pairs, const float *Rt, int rt_stride, double radius, float *out,
                             int out_stride, void *workspace, dpm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DPM_HIP_H */
