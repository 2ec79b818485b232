/*
 * sqfa_hip.h -- C ABI of libsqfa_hip.so, the MI355X (gfx950) implementation of the
 * SQFA pairwise SPD-distance hot path.
 *
 * Plain C, plain pointers and sizes; no torch / C++ types cross this boundary.
 * All data pointers are DEVICE pointers (HBM) unless stated otherwise; all work is
 * enqueued on `stream` (a hipStream_t passed as void*), nothing synchronises.
 *
 * What each entry point replaces in the reference (paths relative to the
 * reference repository root):
 *
 *   sqfa_airm_pairwise   the whole chain executed per closure evaluation
 *       spd_inv_sqrt               src/sqfa/linalg.py:144-162   (per-class whitening)
 *       conjugate_matrix           src/sqfa/linalg.py:19-45     (all-pairs W_j A_i W_j^T)
 *       generalized_eigenvalues    src/sqfa/linalg.py:48-70     (batched eigvalsh + flip)
 *       affine_invariant_sq        src/sqfa/distances.py:46-67  (sum log^2)
 *       affine_invariant           src/sqfa/distances.py:70-89  (sqrt(. + 1e-6))
 *       fisher_rao_lower_bound[_sq] src/sqfa/distances.py:177-237 (scale = 1/2 on embeddings)
 *       closure loss               src/sqfa/_optim.py:88-96     (-mean over i>j) via uniform_weight
 *       check_distances_valid      src/sqfa/_optim.py:16-30     via nonfinite_out
 *       autograd backward of all of the above (torch LinalgEighBackward0 x2, BmmBackward)
 *                                  via gradA_out / gradB_out (closed form)
 *
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef SQFA_HIP_H
#define SQFA_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types */
#define SQFA_F32 0
#define SQFA_F64 1

/* status codes (return values) */
#define SQFA_OK                 0
#define SQFA_ERR_BAD_ARGUMENT  -1   /* null pointer, negative size, bad dtype, bad shard ... */
#define SQFA_ERR_UNSUPPORTED_M -2   /* m outside [1, sqfa_hip_max_dim()] */
#define SQFA_ERR_WORKSPACE     -3   /* workspace_bytes smaller than sqfa_airm_workspace_bytes() */
#define SQFA_ERR_LAUNCH        -4   /* a HIP launch failed; see sqfa_hip_last_error() */

/* Library / build identification. */
int sqfa_hip_version(void);            /* 1000*major + minor */
const char *sqfa_hip_arch(void);       /* "gfx950" */
int sqfa_hip_max_dim(void);            /* largest matrix size m handled natively */
const char *sqfa_hip_last_error(void); /* text of the last HIP error seen by this library (host thread local) */

/*
 * Tile geometry used for a given problem (informational): tiles form an
 * n_tiles_i x n_tiles_j grid over (A classes) x (B classes); tile (bi,bj) is
 * processed by a call iff (bi + bj) % shard_count == shard_index (and, in self mode,
 * it contains a pair i>j).  tile_j / n_tiles_j describe the WIDEST tiling; a call
 * whose shard would not fill the GPU halves the tile width (possibly repeatedly), which
 * changes which pairs a shard evaluates but never the union over the shards of one
 * shard_count.  Returns SQFA_OK or an error code.
 */
int sqfa_airm_tiling(int nA, int nB, int m, int dtype,
                     int *tile_i, int *tile_j, int *n_tiles_i, int *n_tiles_j, int *padded_m);

/* Bytes of device workspace sqfa_airm_pairwise[_opt] needs for this problem with ANY shard_count and ANY options (0 on error). */
size_t sqfa_airm_workspace_bytes(int nA, int nB, int m, int dtype);
/* The same for calls with this shard_count and geometry policy only (the lane geometry and the tile width, hence the slab,
 * depend on them): at C=1000, m=16 55 MB for shard_count 1 (the slab holds exactly the tiles a shard owns) instead of the
 * bound above.  geometry_policy: as in sqfa_airm_options (0 for sqfa_airm_pairwise). */
size_t sqfa_airm_workspace_bytes_sharded(int nA, int nB, int m, int dtype, int shard_count, int geometry_policy);

/* Per-call options of sqfa_airm_pairwise_opt / sqfa_airm_eigenvalues_backward (NULL = all defaults).  There is no
 * process-wide policy state in the library: what a call does depends on its arguments only.
 *   geometry_policy      launches with few pairs run on "small-launch" lane geometries (more lanes per pair, same padded
 *                        sizes: configs.hpp, SQFA_CONFIGS_F32_SMALL): 0 = by pair count (default), 1 = wherever such a row
 *                        exists, -1 = never (tests run both kinds of row; A/B timing)
 *   class_factor_policy  the class factor pass K0b (pair_kernel.hpp, class_factor_kernel) runs for launches with enough
 *                        pairs to pay for its latency: 0 = by pair count (default), 1 = always, -1 = never.  Results agree
 *                        to rounding either way; every shard of a job must pass the same value.
 *   sweep_counter        NULL, or a device buffer of two uint64 {sum of Jacobi sweeps, number of wave rounds} the tile
 *                        kernel adds to atomically (introspection for benchmarks)
 *   mean_metric_policy   1 = when the factor pass runs, sizes m <= 17 and m = 25..32 orthogonalise the factor columns in the
 *                        metric of the MEAN class (class_factor_mean_kernel) instead of the plain inner product: 0.2-0.8
 *                        fewer sweeps (-8 % at m = 16 / 17) for classes that share a dominant covariance, as real class
 *                        statistics do; +1-3 % on classes scattered around a multiple of I (BASELINE's synthetic
 *                        generator), hence opt-in.  0 (default) / -1 = off.  Every shard of a job passes the same value. */
typedef struct sqfa_airm_options {
  int geometry_policy;
  int class_factor_policy;
  unsigned long long *sweep_counter;
  int mean_metric_policy;
} sqfa_airm_options;

/*
 * Pairwise affine-invariant distances between two batches of SPD matrices, the
 * weighted sum of those distances, and the gradient of that sum.
 *
 *   A        (nA, m, m) row-major contiguous SPD matrices (dtype)
 *   B        (nB, m, m) or NULL.  NULL (with nB == 0) selects SELF mode: B is A and
 *            only the unordered pairs i > j are evaluated (the reference evaluates
 *            all ordered pairs and then keeps i > j, src/sqfa/_optim.py:94).
 *   scale    d2 = scale * sum_k log(lambda_k)^2      (1 for AIRM, 0.5 for Calvo-Oller)
 *   eps      D = sqrt(d2 + eps) when sqrt_mode != 0, else D = d2
 *   pair_weights  NULL, or (nA, nB) row-major (dtype): w_ij.  In SELF mode the weight
 *            of the unordered pair {i,j} is w_ij + w_ji.
 *   uniform_weight  used when pair_weights == NULL: every evaluated pair has this
 *            weight (e.g. -1/P for the closure loss, P the GLOBAL pair count).
 *   shard_index, shard_count   tile shard processed by this call (0,1 = everything).
 *   loss_out      (1) dtype: sum over evaluated pairs of w * D          (may be NULL)
 *   gradA_out     (nA, m, m) dtype: d loss / d A (SELF mode: full gradient wrt the
 *                 shared batch).  NULL = forward only (no eigenvectors kept).
 *   gradB_out     (nB, m, m) dtype, cross mode only (ignored / may be NULL in SELF mode)
 *   dist_out      (nA, nB) dtype: D_ij for evaluated pairs; SELF mode also writes D_ji
 *                 and the diagonal (sqrt(eps) or 0).  Entries of tiles outside the shard
 *                 are left untouched.                                       (may be NULL)
 *   eig_out       (nA, nB, m) dtype: generalized eigenvalues of (A_i, B_j), UNSORTED;
 *                 SELF mode mirrors 1/lambda into (j,i) and writes ones on the diagonal.
 *                                                                           (may be NULL)
 *   nonfinite_out (2) int32: {#pairs with D = NaN, #pairs with D = +-inf} among the
 *                 evaluated pairs                                           (may be NULL)
 *   workspace     device scratch of at least sqfa_airm_workspace_bytes() bytes
 *   stream        hipStream_t
 *
 * Outputs of this shard only; a multi-GPU caller sums loss/grad/nonfinite over shards.
 * Results are bitwise reproducible run to run for fixed inputs and sharding.
 */
int sqfa_airm_pairwise(const void *A, int nA, const void *B, int nB, int m, int dtype,
                       double scale, double eps, int sqrt_mode,
                       const void *pair_weights, double uniform_weight,
                       int shard_index, int shard_count,
                       void *loss_out, void *gradA_out, void *gradB_out,
                       void *dist_out, void *eig_out, int *nonfinite_out,
                       void *workspace, size_t workspace_bytes, void *stream);
/* The same call with explicit options (NULL = sqfa_airm_pairwise). */
int sqfa_airm_pairwise_opt(const void *A, int nA, const void *B, int nB, int m, int dtype,
                           double scale, double eps, int sqrt_mode,
                           const void *pair_weights, double uniform_weight,
                           int shard_index, int shard_count,
                           void *loss_out, void *gradA_out, void *gradB_out,
                           void *dist_out, void *eig_out, int *nonfinite_out,
                           void *workspace, size_t workspace_bytes, void *stream,
                           const sqfa_airm_options *options);

/*
 * Backward of the generalized eigenvalues themselves (the reference's generalized_eigenvalues,
 * src/sqfa/linalg.py:48-70, is autograd-transparent and its tutorial builds custom distance_funs
 * on it, docs/source/tutorials/distances.md:127-178): gradient of
 *     sum_{i,j,k} eig_weights[i,j,k] * lambda_k(A_i, B_j)
 * with respect to A and B, in closed form (d lambda_k/dA = u_k u_k^T, d lambda_k/dB = -lambda_k u_k u_k^T
 * for the generalized eigenvectors U, U^T B U = I).
 *   eig_weights  (nA, nB, m) dtype, indexed like sqfa_airm_pairwise's eig_out of the SAME inputs
 *                (its unsorted column order is a deterministic function of the inputs; a caller that
 *                sorts the eigenvalues scatters its upstream gradient back through the permutation)
 *   gradA_out (nA,m,m), gradB_out (nB,m,m; cross mode); SELF mode (B NULL, nB 0): eig_weights[j,i,k]
 *                weighs the mirrored value 1/lambda_k and the result is the gradient wrt the shared batch.
 *   workspace as for sqfa_airm_pairwise; options: those of the call that produced eig_out (the column order depends on
 *   the lane geometry), NULL = defaults.
 */
int sqfa_airm_eigenvalues_backward(const void *A, int nA, const void *B, int nB, int m, int dtype,
                                   const void *eig_weights, void *gradA_out, void *gradB_out,
                                   void *workspace, size_t workspace_bytes, void *stream,
                                   const sqfa_airm_options *options);

/*
 * T_c = Psi_c F^T for c = 0..C-1: the streaming half of the projection S_c = F Psi_c F^T of the
 * class scatter matrices into feature space.  Replaces conjugate_matrix
 * (src/sqfa/linalg.py:19-45) as called by transform_scatters (src/sqfa/model.py:172-188):
 * Psi (C,D,D) is read from HBM exactly once; S_c = F T_c and, in the backward pass,
 * dL/dF = sum_c (G_c + G_c^T) T_c^T need only T (C,D,K).  Psi_c is assumed symmetric
 * (covariance / second-moment matrices).
 *   F (K,D) row-major, Psi (C,D,D), T_out (C,D,K) row-major; float32 or float64, D % 4 == 0, K <= 64
 *   (SQFA_ERR_UNSUPPORTED_M otherwise: the caller keeps its own path for those shapes).
 */
int sqfa_project_scatters(const void *F, int K, int D, const void *Psi, int C, int dtype, void *T_out,
                          void *stream);

/*
 * The same product from BLOCK-TRIANGULAR PACKED statistics: symmetric Psi_c stored once (per fit) as its lower block
 * triangle -- row blocks of 16 rows, each holding the 16 x 64 tiles of the column stripes up to and including its 64-wide
 * diagonal block, contiguous -- so that every closure streams 51-54 % of the bytes of the full tensor (D = 784: 54.1 %,
 * 2048: 51.6 %, 3072: 51.0 %) and the statistics need half the memory.  Both halves of the symmetric product come from
 * the same tile (project_packed_kernel.hip); results agree with sqfa_project_scatters to rounding (other summation order).
 *   sqfa_packed_scatter_elems(D)    elements per class of the packed form (0: D not supported)
 *   sqfa_pack_scatters              Psi (C,D,D) -> packed_out (C, sqfa_packed_scatter_elems(D)); reads the LOWER triangle
 *                                   and the diagonal blocks of Psi
 *   sqfa_project_scatters_packed    T_out (C,D,K) = Psi_c F^T from the packed form
 * float32, D % 16 == 0, 16 <= D <= 4096, K <= 64 (SQFA_ERR_UNSUPPORTED_M otherwise: keep the full tensor and
 * sqfa_project_scatters).  One workgroup per class: meant for C >= a few hundred classes.
 */
size_t sqfa_packed_scatter_elems(int D);
int sqfa_pack_scatters(const void *Psi, int C, int D, int dtype, void *packed_out, void *stream);
int sqfa_project_scatters_packed(const void *F, int K, int D, const void *packed, int C, int dtype, void *T_out,
                                 void *stream);

/*
 * The two small products around it, each reading T (C,D,K) once:
 *   sqfa_feature_scatters           S_c = F T_c            -> S_out (C,K,K)   (the projected scatters;
 *                                   noise / embedding are added by the caller)
 *   sqfa_feature_scatters_backward  P_g = sum over the classes c = g, g + n_groups, ... of
 *                                   (G_c + G_c^T) T_c^T  -> partial_out (n_groups,K,D); G (C,K,K) is
 *                                   the gradient wrt S; dL/dF = sum_g P_g (fixed order: reproducible)
 * They replace the einsum of conjugate_matrix (src/sqfa/linalg.py:41) and its autograd backward.
 * float32 or float64, K <= 64; forward needs D % 4 == 0
 * (SQFA_ERR_UNSUPPORTED_M otherwise: the caller keeps its own expression).
 */
int sqfa_feature_scatters(const void *F, int K, int D, const void *T, int C, int dtype, void *S_out, void *stream);
int sqfa_feature_scatters_backward(const void *G, const void *T, int C, int D, int K, int dtype, int n_groups,
                                   void *partial_out, void *stream);

/*
 * Closure glue: the same two products with the elementwise steps around them folded in, and the
 * parametrization, so that one closure evaluation is 8 (SecondMomentsSQFA) / 11 (SQFA) launches.
 *   sqfa_feature_scatters_ex        as sqfa_feature_scatters, plus `noise` added to the diagonal (the
 *                                   feature_noise regulariser, src/sqfa/model.py:537-538) and, when means_f
 *                                   (C,K) (the projected class means) is not NULL, the Calvo-Oller embedding
 *                                   [[S + m m^T, m], [m^T, 1]] (src/sqfa/distances.py:141-174) written as
 *                                   (C, K+1, K+1) into S_out
 *   sqfa_feature_scatters_backward_ex  as sqfa_feature_scatters_backward for G stored with row pitch / class
 *                                   size ldg (K, or K+1 to read the top-left block of a gradient wrt the embedding);
 *                                   g_symmetric != 0: the caller guarantees G_c = G_c^T (true for the gradients
 *                                   sqfa_airm_pairwise writes), G_c + G_c^T is then read as 2 G_c along rows only
 *   sqfa_embed_backward_means       gm_out (C,K) = (G + G^T) m + gE[:K,K] + gE[K,:K]: gradient wrt the projected means
 *   sqfa_sphere_forward             F = X / ||X||_row, norms_out (K)    (Sphere.forward, src/sqfa/constraints.py:37)
 *   sqfa_sphere_backward            grad_out (K,D) = gloss * (gF - F (F.gF)) / ||X||  with
 *                                   gF = extra + sum_g partials[g]  (partials (n_groups,K,D) from the backward
 *                                   product; extra (K,D) optional; gloss: device scalar or NULL = 1;
 *                                   norms NULL = no constraint: grad_out = gloss * gF)
 */
int sqfa_feature_scatters_ex(const void *F, int K, int D, const void *T, int C, int dtype, double noise,
                             const void *means_f, void *S_out, void *stream);
int sqfa_feature_scatters_backward_ex(const void *G, int ldg, const void *T, int C, int D, int K, int dtype,
                                      int n_groups, int g_symmetric, void *partial_out, void *stream);
int sqfa_embed_backward_means(const void *gE, const void *means_f, int C, int K, int dtype, void *gm_out, void *stream);
int sqfa_sphere_forward(const void *X, int K, int D, int dtype, void *F_out, void *norms_out, void *stream);
int sqfa_sphere_backward(const void *X, const void *norms, int K, int D, int dtype, const void *partials, int n_groups,
                         const void *extra, const void *gloss, void *grad_out, void *stream);

/*
 * L-BFGS search direction in compact form for optimizer state kept on the device (the reference optimises
 * with torch.optim.LBFGS, src/sqfa/_optim.py:78-82; this evaluates the same two-loop recursion as two
 * triangular solves and four (history x n) products in six launches, sqfa_amd/_lbfgs.py).
 *   S, Y (h, n): ring buffers of steps and gradient differences; SY (h, h): SY[i][j] = s_i . y_j; h <= 128
 *   sqfa_lbfgs_push        writes (s, y) into ring row `slot` and refreshes row and column `slot` of SY
 *   sqfa_lbfgs_direction   d_out (n) = -H g for the k pairs listed (HOST array `slots`, chronological order);
 *                          H_diag: device scalar (initial Hessian scale) or NULL = 1
 *   work                   scratch of sqfa_lbfgs_work_elems(h, n) elements of the dtype, shared by both calls
 *                          (partial dot products, solve vectors, one n-vector)
 */
int sqfa_lbfgs_max_history(void);
size_t sqfa_lbfgs_work_elems(int h, int n);
int sqfa_lbfgs_push(void *S, void *Y, void *SY, int h, int n, int slot, const void *s, const void *y, void *work,
                    int dtype, void *stream);
/* y_out = g - g_prev, s_out = t d, scalars_out (5) = [max|g|, max|s|, y.s, y.y, y.s / y.y]: the element-wise part of an
 * iteration and its stopping-rule scalars in two launches; work: sqfa_lbfgs_work_elems(h, n) >= 1024 elements */
int sqfa_lbfgs_step_stats(const void *g, const void *g_prev, const void *d, double t, int n, void *y_out, void *s_out,
                          void *scalars_out, void *work, int dtype, void *stream);
int sqfa_lbfgs_direction(const void *S, const void *Y, const void *SY, int h, int n, const int *slots, int k,
                         const void *g, const void *H_diag, void *d_out, void *work, int dtype, void *stream);

/*
 * Per-pair Gaussian terms behind the reference's other distance_fun operators -- bhattacharyya
 * (src/sqfa/distances.py:240-280), mahalanobis[_sq] (:283-361), hellinger (:364-393),
 * fisher_rao_same_cov (:396-432) -- which all reduce to, with Sbar_ij = (Sigma_i + Sigma_j)/2 and
 * delta_ij = mu_i - mu_j:
 *     Q_ij  = delta^T Sbar^-1 delta          LD_ij = log det Sbar
 * The reference materialises the (nA,nB,K,K) tensor of mean covariances and runs batched inv / logdet
 * on it; here one lane group factorises one pair at a time and only (nA,nB) outputs exist.
 *   muA (nA,m), covA (nA,m,m), muB (nB,m), covB (nB,m,m): row-major, dtype; m <= 64
 *   Q_out, LD_out   (nA,nB) dtype, either may be NULL
 *   gQ, gLD         (nA,nB) dtype upstream gradients (either may be NULL), used when gcovA_out != NULL:
 *   gmuA_out (nA,m), gcovA_out (nA,m,m): gradient of sum_ij (gQ_ij Q_ij + gLD_ij LD_ij) wrt muA / covA
 *                   (full symmetric matrices).  Both NULL = forward only.
 * The B-side gradient is the same call with A and B swapped and gQ / gLD transposed; when B is A the
 * caller passes gQ + gQ^T, gLD + gLD^T and takes the A side as the total.  Deterministic (no atomics).
 */
int sqfa_gauss_pair_terms(const void *muA, const void *covA, int nA, const void *muB, const void *covB, int nB,
                          int m, int dtype, const void *gQ, const void *gLD, void *Q_out, void *LD_out,
                          void *gmuA_out, void *gcovA_out, void *stream);

/*
 * Matrix functions of SPD matrices, f(S) = Q f(Lambda) Q^T per class, and their backward -- spd_log and spd_sqrt of the
 * reference (src/sqfa/linalg.py:165-183, 121-141: torch.linalg.eigh + einsum), as used by log_euclidean[_sq]
 * (src/sqfa/distances.py:92-138).  The eigen-decomposition is one-sided Jacobi, run to convergence, on the Cholesky
 * factor of each class (S = L L^T, L V = Q Sigma  =>  lambda = sigma^2 with the relative accuracy log needs), in double
 * whatever the dtype; results are bitwise reproducible.
 *   S (n,m,m) dtype, F_out (n,m,m) dtype or NULL; U_out (n,m,m) and lam_out (n,m): FLOAT64, eigenvectors as columns,
 *   eigenvalues unsorted -- what sqfa_spd_function_backward needs (a non-SPD class yields NaN there and in F_out)
 *   kind: SQFA_SPD_LOG, SQFA_SPD_SQRT, SQFA_SPD_INV_SQRT (the symmetric inverse root)
 *   workspace: sqfa_spd_function_workspace_bytes(n, m, dtype) bytes; m <= sqfa_hip_max_dim()
 * sqfa_spd_function_backward: gradS_out (n,m,m) dtype = Q [(Q^T sym(G) Q) o Gamma] Q^T for the upstream gradient
 *   G (n,m,m) dtype wrt F (Daleckii-Krein; Gamma = divided differences of f, evaluated in forms that stay finite for
 *   repeated eigenvalues, where torch's eigh backward -- the reference's autograd -- returns inf / NaN).
 */
#define SQFA_SPD_LOG 0
#define SQFA_SPD_SQRT 1
#define SQFA_SPD_INV_SQRT 2
size_t sqfa_spd_function_workspace_bytes(int n, int m, int dtype);
int sqfa_spd_function(const void *S, int n, int m, int dtype, int kind, void *F_out, double *U_out, double *lam_out,
                      void *workspace, size_t workspace_bytes, void *stream);
int sqfa_spd_function_backward(const double *U, const double *lam, const void *G, int n, int m, int dtype, int kind,
                               void *gradS_out, void *stream);

/* Introspection (benchmarks / development; not needed by a reference-side binding).
 *
 * sqfa_airm_profile(1): every following sqfa_airm_pairwise call brackets its pair tile kernel
 *   with hipEvents recorded on the caller's stream.  sqfa_airm_profile_read synchronises on
 *   those events, returns the summed kernel time [ms] and the number of launches, and
 *   releases them (call it outside any timed region).  The switch is an atomic flag; it changes what is
 *   RECORDED around a launch, never which kernel runs. */
int sqfa_airm_profile(int enable);
int sqfa_airm_profile_read(double *tile_kernel_ms_total, int *launches);
int sqfa_project_profile_read(double *kernel_ms_total, int *launches);  /* same, for sqfa_project_scatters */

#ifdef __cplusplus
}
#endif
#endif /* SQFA_HIP_H */
