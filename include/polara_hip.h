/*
 * polara_hip.h — C ABI of libpolarahip.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * factorization-and-scoring hot path of evfro/polara (PureSVD + CoFFee).
 *
 * The reference is pure Python and has NO FFI of its own; each entry point below replaces one
 * NumPy/SciPy/numba call site on the hot path (cited as file:line relative to the reference
 * tree).  INTEGRATION.md shows the ctypes binding a Polara maintainer would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - All `*_dev` pointers are DEVICE pointers (HBM) owned by the caller (the host layer allocates
 *    them through torch, cupy, hipMalloc — anything).  The library never allocates device memory
 *    and keeps no global state; scratch space is passed explicitly (`work`, sized by the matching
 *    pk_*_work_bytes function).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Every call only ENQUEUES
 *    work on that stream and returns; the caller synchronises.
 *  - Return value: 0 = ok, negative = error class (PK_E_*).  pk_last_error() returns a
 *    thread-local message for the last failing call on the calling thread.
 *  - Matrices are row-major with an explicit leading dimension (in elements).
 *  - Index dtypes: row pointers int64, column indices int32, results int64 (like the reference's
 *    `top_recs`, models.py:400).
 */
#ifndef POLARA_HIP_H
#define POLARA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PK_OK 0
#define PK_E_INVALID (-1)  /* bad argument                                  */
#define PK_E_LAUNCH (-2)   /* HIP launch / runtime error                    */
#define PK_E_UNSUPPORTED (-3)
#define PK_E_NOCONV (-4)   /* pk_svd_build: stopped at max_outer without converging (ARPACK's ArpackNoConvergence) */

#define PK_VAL_F32 0
#define PK_VAL_F64 1

const char *pk_last_error(void);
int pk_version(void);
/* Process-wide options, set by explicit calls — the library reads NO environment variable (round 6).  They exist for the
 * tests, which force a code path on inputs too small to take it by themselves:
 *   "score_boot_tiles"     tiles of the sweep's threshold bootstrap (default 16; 0 = off)
 *   "score_head_tiles"     head of the two-phase sweep whatever the user count (default: pk_score_two_phase_plan's rule; 0 = off)
 *   "score_phase2_splits"  item splits of its second phase
 * unset != 0 returns the option to its default.  Unknown names: PK_E_INVALID. */
int pk_set_option(const char *name, int32_t value, int32_t unset);
/* hipMemcpyAsync device -> host on `stream` (pinned destination: asynchronous): the hand-over of a pass's lists as one more
 * call of the pass (the reference returns host arrays: models.py:400-405). */
int pk_copy_to_host_async(void *stream, void *dst_host, const void *src_dev, int64_t bytes);
/* number of visible HIP devices (>=1 required by every compute call); fills name of `device`. */
int pk_device_info(int device, char *name, int name_len, int *cu_count, int64_t *hbm_bytes);
/* Loads every code object of the library on the current device (the runtime would otherwise load each translation
 * unit's at the first launch of one of its kernels — inside the caller's first build).  Called by pk_ctx_create and by
 * the Python layer's HipOps(); idempotent, a few milliseconds once per process and device. */
int pk_warm_up(void);

/* ------------------------------------------------------------------------------------------
 * K1 / K4.  CSR x dense  (fp64 accumulate):   out[r, 0:nc] = sum_p vals[p] * X[indices[p], 0:nc]
 *
 * Replaces: scipy `csr_matvec(s)` inside ARPACK's reverse-communication loop for
 * `svds(A, k)` (models.py:841-844: operator XH_X = A^T(A x)), and `test_matrix.dot(v)`
 * (models.py:860, fold-in E = A_test V).  A^T products use the same kernel on the CSC arrays.
 *
 * Work is described by a task list built once per matrix by the host layer (polara_amd/csr.py):
 * task t covers nnz range [task_begin[t], task_end[t]) of row task_row[t]; rows longer than the
 * split threshold are cut into several tasks whose partial sums go to `partial[task_slot[t]]`
 * (task_slot >= 0) and are added in slot order by a fix-up pass (deterministic, no atomics).
 * ------------------------------------------------------------------------------------------ */
int pk_spmm_csr_f64(void *stream,
                    int64_t n_tasks, const int32_t *task_row_dev, const int64_t *task_begin_dev,
                    const int64_t *task_end_dev, const int32_t *task_slot_dev,
                    int64_t n_long, const int32_t *long_row_dev, const int32_t *long_slot_begin_dev,
                    const int32_t *long_slot_end_dev,
                    const int32_t *indices_dev, const void *vals_dev, int val_kind,
                    const double *X_dev, int64_t ldx, int32_t nc,
                    double *out_dev, int64_t ldo, double *partial_dev /* [n_slots x nc] or NULL */);
/* Same product with the dense block X given in fp32 (x_kind = PK_VAL_F32; needs nc % 4 == 0, ldx % 4 == 0,
 * 16-byte aligned X) or fp64 (PK_VAL_F64 = pk_spmm_csr_f64).  Accumulation and output are fp64.  Used for the
 * approximate fold-in of the scoring pass: E' = A_test fl32(V) moves half the gather bytes of models.py:860's
 * left factor; the re-scoring kernel certifies the result against the rounding of V (pk_rescore_topk_f64). */
int pk_spmm_csr_x(void *stream,
                  int64_t n_tasks, const int32_t *task_row_dev, const int64_t *task_begin_dev,
                  const int64_t *task_end_dev, const int32_t *task_slot_dev,
                  int64_t n_long, const int32_t *long_row_dev, const int32_t *long_slot_begin_dev,
                  const int32_t *long_slot_end_dev,
                  const int32_t *indices_dev, const void *vals_dev, int val_kind,
                  const void *X_dev, int x_kind, int64_t ldx, int32_t nc,
                  double *out_dev, int64_t ldo, double *partial_dev);
/* The general form behind both: task rows are numbered from `row_base` (task t writes output row task_row[t] -
 * row_base — a row block of a larger plan runs as its own launch), and with accumulate != 0 the result is ADDED
 * to out.  Used for the transposed product of the build, Z = A^T Y (models.py:844's rmatvec), cut into user
 * blocks: the CSC of every block is one slice of a (block, item)-ordered plan (pk_csr_transpose with
 * rows_per_block > 0), block b adds its share to Z in launch order (deterministic), and the rows of Y that one
 * launch gathers stay cache-resident instead of spanning the whole user range. */
int pk_spmm_csr_ex(void *stream,
                   int64_t n_tasks, const int32_t *task_row_dev, const int64_t *task_begin_dev,
                   const int64_t *task_end_dev, const int32_t *task_slot_dev,
                   int64_t n_long, const int32_t *long_row_dev, const int32_t *long_slot_begin_dev,
                   const int32_t *long_slot_end_dev,
                   const int32_t *indices_dev, const void *vals_dev, int val_kind,
                   const void *X_dev, int x_kind, int64_t ldx, int32_t nc,
                   double *out_dev, int64_t ldo, double *partial_dev, int64_t row_base, int32_t accumulate,
                   int64_t x_rows /* rows of X (= columns of the sparse matrix), or 0 if unknown: with fewer than 2^24 rows
                   and X below 4 GiB the kernels address X with 32-bit offsets (a quarter of the address arithmetic) */);

/* ------------------------------------------------------------------------------------------
 * Ingest.  The index work around the hot path, on the device:
 * `coo_matrix((val, (rows, cols))).tocsr()` of get_training_matrix / get_test_matrix
 * (models.py:172-175, 198-203 <- data.py:794-817, 835-862: duplicates summed, rows sorted by column),
 * the CSC image behind the transposed products of `svds` (models.py:844), and the wave-task plans.
 * Building blocks: a stable LSD radix sort of (key, u32 payload) pairs and an exclusive scan.
 * ------------------------------------------------------------------------------------------ */
/* out[0..n] = exclusive prefix sums of in[0..n-1] (out[n] = total).  work >= pk_scan_work_bytes(n). */
int64_t pk_scan_work_bytes(int64_t n);
int pk_exclusive_scan_i32(void *stream, int64_t n, const int32_t *in_dev, int64_t *out_dev, void *work_dev);
/* Stable ascending sort of n (key, payload) pairs by the low key_bits bits of the key (key_bytes = 4 or 8).
 * The sorted pairs end in (keys, vals) when *result_in_tmp == 0, else in (keys_tmp, vals_tmp).
 * work >= pk_radix_work_bytes(n). */
int64_t pk_radix_work_bytes(int64_t n);
int pk_radix_sort_pairs(void *stream, int64_t n, int32_t key_bytes, void *keys_dev, uint32_t *vals_dev,
                        void *keys_tmp_dev, uint32_t *vals_tmp_dev, int32_t key_bits, void *work_dev,
                        int32_t *result_in_tmp);
/* The serving order of the catalogue: rows of V [n x K] by DESCENDING Euclidean norm, ties by row id (the order of
 * `np.argsort(-np.linalg.norm(V, axis=1), kind='stable')`; the reference keeps V as built, models.py:849 — the re-indexing
 * serves the sweep's pruning bound).  order_dev[p] = the row at position p, rank_dev[row] = its position, V_sorted_dev (or
 * NULL) [n x K] = the rows in that order.  work >= pk_row_norm_order_work_bytes(n). */
int64_t pk_row_norm_order_work_bytes(int64_t n);
int pk_row_norm_order_f64(void *stream, int64_t n, int32_t K, const double *V_dev, int64_t ldv, int32_t *order_dev,
                          int32_t *rank_dev, double *V_sorted_dev, void *work_dev);
/* COO triplets (device arrays, any order, duplicates allowed; entry i has row rows_dev[i * idx_stride] and column
 * cols_dev[i * idx_stride] — idx_stride = 2 reads the interleaved int64 [nnz x 2] index array of `to_coo`,
 * data.py:794-817, as it is) -> canonical CSR: indptr int64[n_rows + 1],
 * indices int32 / values (val_kind) sized for nnz entries, of which the first *n_unique_dev are used (duplicates
 * are summed in their original order).  err_dev[0] != 0 afterwards = an index was out of range.
 * work >= pk_coo_to_csr_work_bytes(nnz). */
int64_t pk_coo_to_csr_work_bytes(int64_t nnz);
int pk_coo_to_csr(void *stream, int64_t nnz, const int64_t *rows_dev, const int64_t *cols_dev, int64_t idx_stride,
                  const void *vals_dev,
                  int val_kind, int64_t n_rows, int64_t n_cols, int64_t *indptr_dev, int32_t *indices_dev,
                  void *values_dev, int64_t *n_unique_dev, int32_t *err_dev, void *work_dev);
/* CSR -> CSC.  rows_per_block = 0: t_indptr int64[n_cols + 1].  rows_per_block > 0: the rows are cut into
 * ceil(n_rows / rows_per_block) blocks and t_indptr has n_blocks * n_cols + 1 entries — entry b * n_cols + c starts
 * column c restricted to the rows of block b (row ids stay global, ascending within a column segment). */
int64_t pk_csr_transpose_work_bytes(int64_t nnz);
int pk_csr_transpose(void *stream, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *indptr_dev,
                     const int32_t *indices_dev, const void *values_dev, int val_kind, int64_t rows_per_block,
                     int64_t *t_indptr_dev, int32_t *t_indices_dev, void *t_values_dev, void *work_dev);
/* Column renaming j -> col_map[j] with every row re-sorted by the new ids (row pointers unchanged). */
int64_t pk_csr_relabel_work_bytes(int64_t nnz);
int pk_csr_relabel_sorted(void *stream, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *indptr_dev,
                          const int32_t *indices_dev, const void *values_dev, int val_kind, const int32_t *col_map_dev,
                          int32_t *indices_out_dev, void *values_out_dev, void *work_dev);
/* The rows of a CSR ordered by DESCENDING stored-entry count (stable: ties by row id): perm_out[r] = the input row that
 * becomes row r.  The scoring pass groups 32 consecutive users per wave and a group sweeps the catalogue until its LAST
 * user is done; users of similar activity finish at similar depths (ML-20M-shaped: 12.3 -> 9.5 % of the tiles scored). */
int64_t pk_csr_rows_by_length_work_bytes(int64_t n_rows);
int pk_csr_rows_by_length(void *stream, int64_t n_rows, int64_t nnz, const int64_t *indptr_dev, const int32_t *indices_dev,
                          const void *values_dev, int val_kind, int32_t *perm_out_dev, int64_t *new_indptr_dev,
                          int32_t *indices_out_dev, void *values_out_dev, void *work_dev);
/* counts[k] = number of keys equal to k (item popularity: the internal item order of the device path) */
int pk_count_i32(void *stream, int64_t n, const int32_t *keys_dev, int64_t n_bins, int32_t *counts_dev);
/* vals_out[p] = (row_scale[row of p] * vals[p]) * col_scale[indices[p]] (fp64): the diagonal rescaling A' = D_r A D_c of
 * ScaledMatrixMixin (models.py:864-895, preprocessing/matrices.py:71-93) applied to the device CSR. */
int pk_csr_scale_f64(void *stream, int64_t n_rows, const int64_t *indptr_dev, const int32_t *indices_dev, const void *vals_dev,
                     int val_kind, const double *row_scale_dev, const double *col_scale_dev, double *vals_out_dev);
/* Wave-task plan of a CSR (one 64-lane wave per task, rows longer than `split` cut into near-equal tasks whose
 * partial results are added in slot order): phase 1 leaves counts_dev[0..2] = (tasks, long rows, slots), phase 2
 * fills the arrays sized from them.  row_first_task / row_long_index (int64[n_rows + 1], optional) = first task of
 * a row / number of long rows before it, so that a row range can run as its own launch. */
int64_t pk_row_plan_work_bytes(int64_t n_rows);
int pk_row_plan_count(void *stream, int64_t n_rows, const int64_t *indptr_dev, int32_t split, int64_t *counts_dev,
                      void *work_dev);
int pk_row_plan_fill(void *stream, int64_t n_rows, const int64_t *indptr_dev, const void *work_dev,
                     int32_t *task_row_dev, int64_t *task_begin_dev, int64_t *task_end_dev, int32_t *task_slot_dev,
                     int32_t *long_row_dev, int32_t *long_slot_begin_dev, int32_t *long_slot_end_dev,
                     int64_t *row_first_task_dev, int64_t *row_long_index_dev);

/* ------------------------------------------------------------------------------------------
 * K2.  Dense tall-skinny fp64 pieces of the block eigensolver / HOOI.
 * Replace: ARPACK's Fortran re-orthogonalisation + LAPACK QR/SVD inside `svds`
 * (models.py:844; lib/tensor.py:71,75,79) and numpy `qr` (tensor.py:61,63).
 * ------------------------------------------------------------------------------------------ */
/* G[la x lb] = A^T B, A: n x la, B: n x lb.  work >= pk_gram_work_bytes(n, la, lb). */
int64_t pk_gram_work_bytes(int64_t n, int32_t la, int32_t lb);
int pk_gram_f64(void *stream, int64_t n, int32_t la, int32_t lb, const double *A_dev, int64_t lda,
                const double *B_dev, int64_t ldb, double *G_dev, int64_t ldg, void *work_dev);
/* out[n x lout] = X[n x lin] * C[lin x lout]   (out must not alias X) */
int pk_tsmm_f64(void *stream, int64_t n, int32_t lin, int32_t lout, const double *X_dev, int64_t ldx,
                const double *C_dev, int64_t ldc, double *out_dev, int64_t ldo);
/* out[n x lout] = Z[n x lout] - X[n x lin] * C[lin x lout]  (out may alias Z, not X): the projection X - V (V^T X) of the
 * solvers in one pass */
int pk_tsmm_sub_f64(void *stream, int64_t n, int32_t lin, int32_t lout, const double *X_dev, int64_t ldx,
                    const double *C_dev, int64_t ldc, const double *Z_dev, int64_t ldz, double *out_dev, int64_t ldo);
/* out = alpha X C + beta Z1 + gamma Z2 in one pass (Z1 / Z2 may be NULL): one step of the Chebyshev recurrence on a small
 * dense operator — the projected problems of the block Lanczos build (ARPACK solves the same projected problem inside
 * svds, models.py:844) — in ONE launch instead of a Gram product, its reduction and the recurrence. */
int pk_tsmm_axpby_f64(void *stream, int64_t n, int32_t lin, int32_t lout, const double *X_dev, int64_t ldx,
                      const double *C_dev, int64_t ldc, double alpha, double beta, const double *Z1_dev, int64_t ldz1,
                      double gamma, const double *Z2_dev, int64_t ldz2, double *out_dev, int64_t ldo);
/* The bookkeeping of an orthonormalisation pass on the device, one launch: flags[0] += sum |info[i]| (Cholesky verdicts),
 * flags[1] = max(flags[1], max |G - I|) (NaN / inf count as 1), info[0..n_info) zeroed for the next block. */
int pk_orth_check_f64(void *stream, int32_t l, const double *G_dev, int64_t ldg, int32_t *info_dev, int32_t n_info,
                      double *flags_dev);
/* Symmetric positive semi-definite eigen-decomposition by one-sided Jacobi, single workgroup.
 * S (n x n, destroyed) -> evals[n] descending, evecs (n x n, ROW i = i-th eigenvector).
 * info_dev[0] = sweeps used, info_dev[1] = 1 if converged.  n <= 1024. */
int pk_eigh_psd_f64(void *stream, int32_t n, double *S_dev, int64_t lds_, double *evecs_dev, int64_t ldv,
                    double *evals_dev, int32_t max_sweeps, double tol, int32_t *info_dev);
/* Same solve; beyond 136 columns (block Jacobi) every round is a launch of its own instead of ONE cooperative launch with
 * grid barriers: the form to re-run when pk_eigh_psd_f64 reports info_dev[1] = 0 there (a barrier that did not complete). */
int pk_eigh_psd_rounds_f64(void *stream, int32_t n, double *S_dev, int64_t lds_, double *evecs_dev, int64_t ldv,
                           double *evals_dev, int32_t max_sweeps, double tol, int32_t *info_dev);
/* The r LEADING eigenpairs of a symmetric PSD matrix in one launch (Householder tridiagonalisation, Sturm-count
 * multisection, inverse iteration, back-transformation; csrc/eigh_top.hip) — what the HOOI unfoldings need of their Gram
 * matrices (the k = r of `svds(unfolding, k=r)`, lib/tensor.py:70-80).  S (n x n) is read only; evals[0..r) descending,
 * ROW j of evecs = j-th eigenvector.  The kernel checks its result against S: info_dev[0] = 1 when it passed, 0 when
 * nothing usable was written (degenerate or pathological input) — the caller then uses pk_eigh_psd_f64.
 * pk_eigh_top_supported: 8 <= n <= 176, 1 <= r <= min(32, n).  work: pk_eigh_top_work_bytes(n) bytes. */
int pk_eigh_top_supported(int32_t n, int32_t r);
int64_t pk_eigh_top_work_bytes(int32_t n);
int pk_eigh_top_f64(void *stream, int32_t n, const double *S_dev, int64_t lds_, int32_t r, double *evecs_dev, int64_t ldv,
                    double *evals_dev, void *work_dev, int32_t *info_dev);
/* Cholesky of a small SPD Gram matrix and the inverse of its factor: G + shift_rel*trace(G)*I = R^T R, Rinv = R^-1 (upper
 * triangular), so that X <- X Rinv is orthonormal (CholeskyQR; replaces LAPACK's QR inside svds / numpy.qr,
 * models.py:844, tensor.py:61).  info_dev[0] = 0 or (column + 1) of the first non-positive pivot.
 * work: pk_chol_work_bytes(n) bytes (0 for n <= 136). */
int64_t pk_chol_work_bytes(int32_t n);
int pk_chol_rinv_f64(void *stream, int32_t n, const double *G_dev, int64_t ldg, double shift_rel, double *Rinv_dev,
                     int64_t ldr, void *work_dev, int32_t *info_dev);
/* The same on the column-scaled matrix D G D, D = diag(G)^-1/2 (unit diagonal): Rinv = D R'^-1, so X Rinv is orthonormal all the
 * same, but the factorisation sees the conditioning of the scaled block (columns of very different norm that are otherwise
 * nearly orthogonal — filtered Ritz vectors — need one pass where the unscaled form needs two).  shift_rel is relative to
 * trace(D G D) = n.  A non-positive diagonal entry of G is reported like a non-positive pivot. */
int pk_chol_rinv_scaled_f64(void *stream, int32_t n, const double *G_dev, int64_t ldg, double shift_rel, double *Rinv_dev,
                     int64_t ldr, void *work_dev, int32_t *info_dev);
/* out = alpha*Z + beta*Y + gamma*X over n_elems (Chebyshev three-term recurrence); Y/X may be NULL */
int pk_axpbypcz_f64(void *stream, int64_t n_elems, double alpha, const double *Z_dev, double beta,
                    const double *Y_dev, double gamma, const double *X_dev, double *out_dev);
/* partial[b, j] = sum over row block b of (Z[i,j] - theta[j]*X[i,j])^2 ; nblocks = pk_resid_blocks(n) */
int32_t pk_resid_blocks(int64_t n);
int pk_resid_colnorm2_f64(void *stream, int64_t n, int32_t l, const double *Z_dev, int64_t ldz,
                          const double *X_dev, int64_t ldx, const double *theta_dev, double *partial_dev);
/* small general C = op(A) op(B) (any shape, one thread per output; for l x l glue only) */
int pk_dgemm_small_f64(void *stream, int transA, int transB, int32_t M, int32_t N, int32_t K,
                       const double *A_dev, int64_t lda, const double *B_dev, int64_t ldb,
                       double *C_dev, int64_t ldc);
/* X[i, j] *= s[j] */
int pk_scale_cols_f64(void *stream, int64_t n, int32_t l, double *X_dev, int64_t ldx, const double *s_dev);

/* ------------------------------------------------------------------------------------------
 * K3.  Fused scoring: scores = E V^T (fp32 MFMA) + seen-item masking + per-user top-k,
 * then exact fp64 re-scoring of the surviving candidates.  No dense score matrix ever exists.
 *
 * Replaces, per user chunk: `(test_matrix.dot(v)).dot(v.T)` (models.py:860),
 * `downvote_seen_items` (models.py:494-519) and `get_topk_elements`/`topsort`
 * (models.py:488-491, 561-563), i.e. the body of `_slice_recommender` (models.py:359-371).
 * ------------------------------------------------------------------------------------------ */
/* number of float elements of the MFMA-fragment-packed image of an [n x K] factor matrix */
int64_t pk_pack_elems(int64_t n, int32_t K);
int32_t pk_pack_kq(int32_t K); /* K padded to a multiple of 8, divided by 8 */
/* src f64 [n x K] (ld) -> packed f32 fragments (32 rows per tile, zero padded) */
int pk_pack_frag_f32(void *stream, int64_t n, int32_t K, const double *src_dev, int64_t ld, float *dst_dev);
/* candidate capacity (power of two in {16,32,64}) used for a given topk; 0 if topk unsupported */
int32_t pk_candidate_capacity(int32_t topk);
/* Streams all item tiles against 32-user groups; keeps the KC best (fp32 score, item) pairs of
 * each user among items NOT in the user's seen list (seen_ptr == NULL: no filtering).
 * cand_* are [splits x n_users_pad x KC] with n_users_pad = 32*ceil(n_users/32); unused slots have
 * idx -1.  `splits` = S > 1 deals the 32-item tiles of the catalogue round-robin to S sweeps per user group
 * (split h takes tiles h, h+S, ...), each with its own wave and threshold (more, smaller work items when
 * there are few users; interleaved so that every sweep meets the high-norm head first and the exact pruning
 * keeps working); the per-split lists are merged by pk_rescore_topk_f64.
 * seen lists must be sorted ascending per user (CSR canonical form).
 * The item range is swept in chunks of `tiles_per_chunk` 32-item tiles, one launch per chunk, so the
 * packed item factors of a chunk stay resident in the 4 MiB per-XCD L2 while every workgroup streams
 * them; the per-user selection state is parked in `state_dev` between launches.  0 = auto: L2-sized chunks for a
 * full sweep, ONE launch for a pruned one (its groups are spread over the head of the catalogue whatever the launches
 * do, and every boundary parks and restores the lists; pk_score_chunk_launches tells the count).
 * Threshold bootstrap: a sweep that starts cold first scores its first 16 tiles WITHOUT selecting — every lane keeps the
 * KC / 2 largest group maxima in a sorted register list — and starts from the smallest of a user's KC values, a lower
 * bound of its final KC-th best score: a quarter of the pushes and flush sorts of a cold start (pk_set_option
 * "score_boot_tiles" overrides; 0 = off).  Should such a sweep end without a full list, the last slot of the list holds idx -2 and
 * pk_rescore_topk_* sends the user to the exact path. */
int64_t pk_score_state_bytes(int64_t n_users, int32_t splits);
/* recommended number of item splits for this many users (1 when the users alone fill the chip) */
int32_t pk_score_splits(int64_t n_users, int32_t KC);
/* Seen-item lists (the rows of the test CSR: everything downvote_seen_items masks, models.py:494-519)
 * folded into ONE 64-bit record per 32-item tile a user has seen items in: (tile << 32) | item mask.
 * tiles_dev has the capacity of seen_idx_dev and is addressed by the same seen_ptr_dev; the records
 * of user u are tiles_dev[seen_ptr[u] .. seen_ptr[u] + ntiles_dev[u]).  rows_sorted = 1: every row ascending
 * by item (canonical CSR); 0: any order within a row (columns only renamed, e.g. to a serving index). */
int pk_seen_tiles_build(void *stream, int64_t n_users, const int64_t *seen_ptr_dev, const int32_t *seen_idx_dev,
                        int32_t rows_sorted /* 0: rows in arbitrary item order, sorted in LDS */,
                        int64_t max_row_len /* only read when rows_sorted == 0: <= pk_seen_tiles_max_unsorted_row() */,
                        uint64_t *tiles_dev, int32_t *ntiles_dev);
int32_t pk_seen_tiles_max_unsorted_row(void);
int pk_score_candidates_f32(void *stream, int64_t n_users, int64_t n_items, int32_t K,
                            const float *Vp_dev, const float *Ep_dev,
                            const int64_t *seen_ptr_dev, const uint64_t *seen_tiles_dev,
                            const int32_t *seen_ntiles_dev /* all three NULL: nothing is masked */,
                            int32_t KC, int32_t splits /* splits*KC <= 64 */,
                            float *cand_score_dev, int32_t *cand_idx_dev /* [splits][n_users_pad][KC] */,
                            void *state_dev /* pk_score_state_bytes(n_users, splits) */,
                            int32_t tiles_per_chunk /* 0 = auto */,
                            const float *user_bound_dev /* [n_users] or NULL */,
                            const float *tile_bound_dev /* [ceil(n_items/32)] or NULL */,
                            const uint32_t *seen_dense_dev, const int32_t *seen_skip_dev,
                            int32_t dense_tiles /* pk_seen_dense_build output, or NULL, NULL, 0: the stream serves every tile */);
/* The same sweep with the users' side read from the fp64 ROWS of E = test_matrix.dot(v) (models.py:860) as the fold-in
 * leaves them: every wave builds its 32 users' MFMA fragments and pruning bounds in its prologue — the arithmetic of
 * pk_pack_frag_bound_f32, bit for bit: bound_u = ||E[u, :K]|| (1 + 1e-6) + extra_scale * extra[u * extra_ld] — so a pass
 * needs neither the packing launch nor a packed copy of E.  E_dev 16-byte aligned, lde even and >= K; extra_dev: the
 * error weight column of an approximate fold-in, or NULL; tile_bound_dev == NULL: full sweep (no pruning). */
/* 1 when the sweeps of this library take the users' side from the rows of E (the two *_rows_f32 entries below); a probe build of
 * csrc/experiments/ returns 0 and its callers pack fragments first */
int pk_sweep_takes_rows(void);
int pk_score_candidates_rows_f32(void *stream, int64_t n_users, int64_t n_items, int32_t K, const float *Vp_dev,
                                 const double *E_dev, int64_t lde, const double *extra_dev, int64_t extra_ld,
                                 double extra_scale, const int64_t *seen_ptr_dev, const uint64_t *seen_tiles_dev,
                                 const int32_t *seen_ntiles_dev, int32_t KC, int32_t splits, float *cand_score_dev,
                                 int32_t *cand_idx_dev, void *state_dev, int32_t tiles_per_chunk /* 0 = auto */,
                                 const float *tile_bound_dev /* or NULL */, const uint32_t *seen_dense_dev,
                                 const int32_t *seen_skip_dev, int32_t dense_tiles);
/* The pruned sweep in TWO PHASES (replaces the same reference lines: the chunk loop body of models.py:359-371, with
 * downvote_seen_items :494-519 and topsort :488-491 fused in).  A launch of the single sweep lasts as long as its slowest
 * wave — the groups of the heaviest users stay ~270 tiles in the sweep, everybody else 70-130 — so the catalogue is
 * cut: tiles [0, head_tiles) are swept once per group (lists + thresholds in slot 0), the remaining tiles are dealt
 * round-robin to `splits` sweeps per group that all START from the head's threshold (a score below it cannot be among
 * the KC best, whatever a split's own list holds), and the splits + 1 lists of a user are merged into ONE list of KC
 * candidates with exact comparisons: the dependent chain of a group is head + tail / splits tiles at the same number
 * of tile-waves.  The merged list is what pk_rescore_topk_* takes with splits = 1.
 *   work_*   [(splits + 1)][n_users_pad][KC] raw lists;  cand_*  [n_users_pad][KC] merged;
 *   state    pk_score_state_bytes(n_users, splits + 1): records of slot 0 = the head, slots 1.. = the splits (first int64
 *            of a record: the tile at which the group left that sweep; a head exit below head_tiles means the group was
 *            pruned inside the head and its splits did not run).
 * pk_score_two_phase_plan: the default (head_tiles, splits) for a user set and a catalogue; head_tiles = 0: use the single
 * sweep (user sets that fill the chip's wave slots on their own: there the sweep is bound by the work of the head tiles,
 * not by its longest chain, and the scheme was measured slower). */
int pk_score_two_phase_plan(int64_t n_users, int64_t n_items, int32_t KC, int32_t *head_tiles, int32_t *splits);
int pk_score_two_phase_f32(void *stream, int64_t n_users, int64_t n_items, int32_t K,
                           const float *Vp_dev, const float *Ep_dev,
                           const int64_t *seen_ptr_dev, const uint64_t *seen_tiles_dev, const int32_t *seen_ntiles_dev,
                           int32_t KC, int32_t head_tiles, int32_t splits /* (splits + 1) * KC <= 256 */,
                           float *work_score_dev, int32_t *work_idx_dev, float *cand_score_dev, int32_t *cand_idx_dev,
                           void *state_dev, int32_t tiles_per_chunk /* 0 = auto */,
                           const float *user_bound_dev, const float *tile_bound_dev /* both required */,
                           const uint32_t *seen_dense_dev, const int32_t *seen_skip_dev, int32_t dense_tiles);
/* ... with the users' side from the rows of E (see pk_score_candidates_rows_f32; tile_bound_dev required) */
int pk_score_two_phase_rows_f32(void *stream, int64_t n_users, int64_t n_items, int32_t K, const float *Vp_dev,
                                const double *E_dev, int64_t lde, const double *extra_dev, int64_t extra_ld,
                                double extra_scale, const int64_t *seen_ptr_dev, const uint64_t *seen_tiles_dev,
                                const int32_t *seen_ntiles_dev, int32_t KC, int32_t head_tiles, int32_t splits,
                                float *work_score_dev, int32_t *work_idx_dev, float *cand_score_dev, int32_t *cand_idx_dev,
                                void *state_dev, int32_t tiles_per_chunk, const float *tile_bound_dev,
                                const uint32_t *seen_dense_dev, const int32_t *seen_skip_dev, int32_t dense_tiles);
/* Dense seen masks for the first dense_tiles tiles of the catalogue — where the sweep spends its time, and where a user
 * has a record in nearly every tile: dense_dev[(u / 32 * dense_tiles + tile) * 32 + u % 32] = the user's 32-bit mask in
 * that tile (one coalesced 128-byte load per tile and wave instead of a cursor walk with a scattered 8-byte load per
 * lane), skip_dev[u] = the user's stream records below dense_tiles.  dense_dev: pk_seen_dense_bytes(n_users, dense_tiles). */
int64_t pk_seen_dense_bytes(int64_t n_users, int32_t dense_tiles);
int pk_seen_dense_build(void *stream, int64_t n_users, const int64_t *seen_ptr_dev, const uint64_t *seen_tiles_dev,
                        const int32_t *seen_ntiles_dev, int32_t dense_tiles, uint32_t *dense_dev, int32_t *skip_dev);
/* launches pk_score_candidates_f32 issues for these arguments (fixed item chunks; doubling chunks when the
 * pruning bounds are passed) */
int32_t pk_score_chunk_launches(int64_t n_items, int32_t K, int32_t splits, int32_t tiles_per_chunk, int32_t pruned);
/* Exact pruning bounds for pk_score_candidates_f32 (Cauchy-Schwarz: |E_u . V_i| <= ||E_u|| ||V_i||).
 * The reference scores every item for every user (models.py:860); the sweep may instead stop, per
 * group of 32 users, at the first tile from which  ||E_u|| * max_{i >= tile} ||V_i||  can no longer
 * beat the user's current KC-th best score.  Both bounds are rounded UP, so the candidate lists — and
 * the certification done by pk_rescore_topk_f64 — are exactly those of the full sweep.
 *   pk_row_norm_bound_f32:  out[r] >= ||src[r,:]||_2                       (user_bound from E)
 *   pk_tile_norm_bound_f32: out[t] >= max_{i >= 32 t} ||V[i,:]||_2         (tile_bound; work: float[n_items])
 * The bound falls fastest when the internal item order is by descending popularity or norm. */
/* After the pass, the state buffer holds, for split h and user group g (32 users), 64 records of 16
 * bytes starting at byte ((h * n_groups + g) * 64) * 16; the first int64 of each record is the tile at
 * which that group left the sweep (absolute tile index; >= n_tiles - S + 1 when it was never pruned): split h
 * scored ceil((that - h) / S) tiles, for the roofline accounting of bench.py. */
int pk_row_norm_bound_f32(void *stream, int64_t n, int32_t K, const double *src_dev, int64_t ld, float *out_dev);
/* The fp32 image of the item factors the approximate fold-in gathers from: out [n x ld32] (ld32 > K) = fl32(V) in columns
 * 0..K-1, bound_dev[r] (pk_row_norm_bound_f32) in column K, zeros beyond.  stat_dev[2] (uint32): [0] = the bits of the largest
 * bound, [1] != 0 when an entry of V or a bound is not finite — the range check of the image in the same launch. */
int pk_v32_image_f32(void *stream, int64_t n, int32_t K, int32_t ld32, const double *V_dev, int64_t ldv, const float *bound_dev,
                     float *out_dev, uint32_t *stat_dev);
/* pk_pack_frag_f32 and the row bound in one pass over the block (the user side of a scoring pass):
 * bound[r] >= ||src[r,:]||_2 + extra_scale * extra[r * extra_ld]   (extra_dev may be NULL). */
int pk_pack_frag_bound_f32(void *stream, int64_t n, int32_t K, const double *src_dev, int64_t ld, float *dst_dev,
                           float *bound_dev, const double *extra_dev, int64_t extra_ld, double extra_scale);
int pk_tile_norm_bound_f32(void *stream, int64_t n_items, int32_t K, const double *V_dev, int64_t ld,
                           float *work_dev, float *out_dev);
/* Exact fp64 re-scoring + final ordering (score desc, item asc).  Writes topk item ids (int64) and
 * optionally their fp64 scores.  flags[u] != 0 marks users whose result is NOT guaranteed exact by
 * the fp32 candidate pass (bit0: candidate margin below the fp32 error bound; bit1: fewer than
 * topk unseen items) — the host re-runs those through pk_score_exact_rows_f64.
 * v_row_norm_max = max_i ||V[i,:]||_2 (<= 1 for orthonormal factors) enters the fp32 error bound. */
int pk_rescore_topk_f64(void *stream, int64_t n_users, int64_t n_items, int32_t K,
                        const double *V_dev, int64_t ldv, const double *E_dev, int64_t lde,
                        const int64_t *seen_ptr_dev, int32_t KC, int32_t splits,
                        const float *cand_score_dev, const int32_t *cand_idx_dev, int32_t topk,
                        double v_row_norm_max,
                        int64_t *out_idx_dev, double *out_score_dev /* or NULL */, int32_t *flags_dev);
/* General form: the r-th row processed is user rows_dev[r] (NULL: user r, n_rows = n_users) — used to re-do a
 * list of users; e_err_dev (or NULL): per-user bound w_u such that the given E rows satisfy
 * ||E'_u - E_u|| <= 2^-24 w_u (the approximate fold-in against fl32(V): w_u = sum_j |a_uj| ||V_j||).  Scores are
 * then within delta_u = 2^-24 w_u max||V_i|| of the exact ones, and flags bit2 (value 4) marks the users whose
 * order is NOT certified at that accuracy (two consecutive scores of the top-k, or the k-th and the best
 * excluded item, closer than 2 delta_u): the host recomputes their E rows exactly and calls again with
 * e_exact = 1 (the same e_err_dev: the non-candidates are still only known through the approximate sweep).
 * V32_dev (or NULL): fl32 image of the item factors [n_items x ldv32]; used INSTEAD of V_dev in a first call over
 * approximate E rows (e_err_dev given, e_exact = 0) — half the cache lines per gathered row; its rounding,
 * 2^-24 ||E'_u|| max||V_i||, is added to delta_u.  Ignored when the E rows are exact. */
int pk_rescore_topk_rows_f64(void *stream, int64_t n_rows, const int32_t *rows_dev,
                             const int32_t *n_rows_dev /* or NULL; else the list length is min(*n_rows_dev, n_rows) */,
                             int64_t n_users, int64_t n_items,
                             int32_t K, const double *V_dev, int64_t ldv,
                             const float *V32_dev /* or NULL */, int64_t ldv32,
                             const double *E_dev, int64_t lde,
                             const double *e_err_dev, int64_t e_err_ld /* e_err of user u at e_err_dev[u * e_err_ld] */,
                             int32_t e_exact /* 1: these E rows are exact, the candidates came
                             from a sweep over approximate ones (second call) */,
                             const int64_t *seen_ptr_dev, int32_t KC, int32_t splits,
                             const float *cand_score_dev, const int32_t *cand_idx_dev, int32_t topk,
                             double v_row_norm_max,
                             int64_t *out_idx_dev, double *out_score_dev /* or NULL */, int32_t *flags_dev);
/* The same, and the users it flags are appended to a device-side list while it runs: flagged_list_dev[old *count ...] =
 * flagged_offset + user for every user whose flag is not 0 (order arbitrary), *flagged_count_dev advanced by an atomic —
 * what pk_flag_compact(mask = all) makes of the flags afterwards, without its two launches.  The counter is NOT zeroed
 * here (several calls may append to one list: user batches): pk_zero_i32 (n <= 2^20 counters, a kernel) first.
 * list / count both NULL: plain pk_rescore_topk_rows_f64. */
int pk_rescore_topk_rows_list_f64(void *stream, int64_t n_rows, const int32_t *rows_dev,
                             const int32_t *n_rows_dev /* or NULL; else the list length is min(*n_rows_dev, n_rows) */,
                             int64_t n_users, int64_t n_items,
                             int32_t K, const double *V_dev, int64_t ldv,
                             const float *V32_dev /* or NULL */, int64_t ldv32,
                             const double *E_dev, int64_t lde,
                             const double *e_err_dev, int64_t e_err_ld /* e_err of user u at e_err_dev[u * e_err_ld] */,
                             int32_t e_exact /* 1: these E rows are exact, the candidates came
                             from a sweep over approximate ones (second call) */,
                             const int64_t *seen_ptr_dev, int32_t KC, int32_t splits,
                             const float *cand_score_dev, const int32_t *cand_idx_dev, int32_t topk,
                             double v_row_norm_max,
                             int64_t *out_idx_dev, double *out_score_dev /* or NULL */, int32_t *flags_dev,
                             int32_t *flagged_list_dev, int32_t *flagged_count_dev, int32_t flagged_offset);
/* The same with the items' own norm bounds: item_norm_dev[i] >= ||V_i|| (fp32, e.g. pk_row_norm_bound_f32) — the order of
 * two neighbouring entries is then certified against cu (||V_i|| + ||V_i'||) instead of 2 cu max ||V||
 * (cu = 2^-24 (e_err[u] + ||E'_u||)): fewer users to re-fold for the same proof.  NULL: pk_rescore_topk_rows_list_f64. */
int pk_rescore_topk_rows_norms_f64(void *stream, int64_t n_rows, const int32_t *rows_dev,
                             const int32_t *n_rows_dev, int64_t n_users, int64_t n_items,
                             int32_t K, const double *V_dev, int64_t ldv,
                             const float *V32_dev, int64_t ldv32,
                             const double *E_dev, int64_t lde,
                             const double *e_err_dev, int64_t e_err_ld, int32_t e_exact,
                             const int64_t *seen_ptr_dev, int32_t KC, int32_t splits,
                             const float *cand_score_dev, const int32_t *cand_idx_dev, int32_t topk,
                             double v_row_norm_max,
                             int64_t *out_idx_dev, double *out_score_dev, int32_t *flags_dev,
                             int32_t *flagged_list_dev, int32_t *flagged_count_dev, int32_t flagged_offset,
                             const float *item_norm_dev);
int pk_zero_i32(void *stream, int32_t *p_dev, int32_t n);
/* The re-do of flagged users without a host round trip: pk_flag_compact lists the users with (flags & mask) != 0
 * (list capacity n, *count_dev = list length), pk_fold_rows_f64 recomputes the listed rows of E = A_test V in fp64
 * straight from the CSR (row = row_offset + list[r]), pk_rescore_topk_rows_f64 re-scores them (rows_dev = the list,
 * n_rows_dev = count_dev).  All three take the list length from device memory. */
int pk_flag_compact(void *stream, int64_t n, const int32_t *flags_dev, int32_t mask, int32_t *list_dev,
                    int32_t *count_dev);
int pk_fold_rows_f64(void *stream, int64_t cap, const int32_t *list_dev, const int32_t *count_dev, int64_t row_offset,
                     const int64_t *indptr_dev, const int32_t *indices_dev, const void *vals_dev, int val_kind,
                     const double *V_dev, int64_t ldv, int32_t K, double *E_dev, int64_t lde);

/* The product restricted to FLAGGED rows: only the tasks of rows r with (row_flags_dev[r] & flag_mask) != 0 run, every
 * other row of `out` is left alone.  The exact re-fold of the users an ids-only pass could not certify at the accuracy
 * of its approximate fold-in (pk_rescore_topk_rows_list_f64 leaves the flags on the device): same plan, mapping and
 * summation order as pk_spmm_csr_ex with an fp64 dense block, so a re-folded row has the bits of the exact product.
 * Even nc and ldx, X 16-byte aligned (odd ranks: pk_fold_rows_f64). */
int pk_spmm_csr_flagged_f64(void *stream, int64_t n_tasks, const int32_t *task_row_dev, const int64_t *task_begin_dev,
                            const int64_t *task_end_dev, const int32_t *task_slot_dev, int64_t n_long,
                            const int32_t *long_row_dev, const int32_t *long_slot_begin_dev,
                            const int32_t *long_slot_end_dev, const int32_t *indices_dev, const void *vals_dev,
                            int val_kind, const double *X_dev, int64_t ldx, int32_t nc, double *out_dev, int64_t ldo,
                            double *partial_dev, int64_t x_rows, const int32_t *row_flags_dev, int32_t flag_mask);

/* The same product on LISTED rows: list_dev[0 .. min(*count_dev, cap)) names the rows (row = row_offset + list[i]; the list
 * the re-scoring kernel appends the uncertified users to), a wave walks the tasks [row_first_task[row], row_first_task[row + 1])
 * of a listed row — the task arrays are those of the WHOLE plan (pk_row_plan_fill), the long-row arrays those of the row
 * range whose split rows may be listed (their partial sums are added under the same flag predicate as above).  Costs what the
 * listed rows cost: the flag-predicated form launches a wave per task of every row (0.23 ms of early exits on S-1M). */
int pk_spmm_csr_rows_list_f64(void *stream, int64_t cap, const int32_t *list_dev, const int32_t *count_dev, int64_t row_offset,
                              const int64_t *row_first_task_dev, const int32_t *task_row_dev, const int64_t *task_begin_dev,
                              const int64_t *task_end_dev, const int32_t *task_slot_dev, int64_t n_long,
                              const int32_t *long_row_dev, const int32_t *long_slot_begin_dev, const int32_t *long_slot_end_dev,
                              const int32_t *indices_dev, const void *vals_dev, int val_kind, const double *X_dev, int64_t ldx,
                              int32_t nc, double *out_dev, int64_t ldo, double *partial_dev, int64_t x_rows,
                              const int32_t *row_flags_dev, int32_t flag_mask);

/* ------------------------------------------------------------------------------------------
 * K4q.  The approximate fold-in of an ids-only scoring pass against a PACKED image of the item factors
 * (csrc/foldq.hip): E'[u, 0:K] = sum_j a_uj decode(image row j), E'[u, K] = w_u = sum_j a_uj D_j with
 * ||E'_u - E_u|| <= 2^-24 w_u PROVEN (D_j is computed by the encoder from the bits it wrote), columns K+1 .. Kx-1 = 0.
 * Replaces `test_matrix.dot(v)` (models.py:857-860) wherever the pass certifies its lists against that bound
 * (pk_rescore_topk_rows_list_f64 with e_err = column K; users it cannot certify are re-folded in fp64 by
 * pk_fold_rows_f64).  A rank-K row takes pk_q20_lanes(K) * 16 bytes (128 for K <= 50: ONE cache line per gathered
 * entry where the fp32 image takes two); ranks above 202 have no packed image (pk_q20_lanes returns 0).
 * pk_q20_encode_f64: V [n x K] fp64 row-major -> image (pk_q20_image_bytes, 128-byte aligned), the bracket scale table
 * tab_dev (96 doubles), info_dev (int32: 0 = ok, else the factors are outside the format's range: use the fp32 image);
 * work_dev: 768 bytes.  pk_q20_decode_f64: the rows exactly as the fold-in reads them, [n x (K + 1)], column K = D_j
 * (tests, diagnostics).  pk_fold_q20: the row-task plan of pk_spmm_csr_ex; non-negative values only (w_u is a bound
 * for a_uj >= 0); the partial buffer holds Kx doubles per slot. */
int32_t pk_q20_lanes(int32_t K);
double pk_q20_kappa(int32_t K);
int64_t pk_q20_image_bytes(int64_t n, int32_t K);
int pk_q20_encode_f64(void *stream, int64_t n, int32_t K, const double *V_dev, int64_t ldv, void *img_dev,
                      double *tab_dev, void *work_dev, int32_t *info_dev);
int pk_q20_decode_f64(void *stream, int64_t n, int32_t K, const void *img_dev, const double *tab_dev, double *out_dev,
                      int64_t ldo);
int pk_fold_q20(void *stream, int64_t n_tasks, const int32_t *task_row_dev, const int64_t *task_begin_dev,
                const int64_t *task_end_dev, const int32_t *task_slot_dev, int64_t n_long,
                const int32_t *long_row_dev, const int32_t *long_slot_begin_dev, const int32_t *long_slot_end_dev,
                const int32_t *indices_dev, const void *vals_dev, int val_kind, const void *img_dev,
                const double *tab_dev, int64_t n_items, int32_t K, int32_t Kx, double *out_dev, int64_t ldo,
                double *partial_dev);
/* dst[perm_dev[r], :] = src_dev[r, :] for rows of `width` int64 (perm_dev == NULL: a plain copy): the lists of a pass whose
 * users were grouped by activity go back to the caller's user order.  `dst` is device memory or MAPPED PINNED HOST memory:
 * the host-side [n_users x topk] int64 array of get_recommendations (models.py:400-405) can be written by the kernel
 * itself, without a copy-engine transfer behind the pass. */
int pk_scatter_rows_i64(void *stream, int64_t n_rows, int32_t width, const int64_t *src_dev, const int64_t *perm_dev,
                        int64_t *dst);
/* dst[e] = src_dev[e] >= 0 ? table_dev[src_dev[e]] : -1: internal item positions -> the item ids of the caller's index
 * (the renaming get_recommendations applies to its result, models.py:400-405 return ids of data.index.itemid), on the
 * device; `dst` is device or mapped pinned host memory. */
int pk_map_ids_i64(void *stream, int64_t n, const int64_t *src_dev, const int64_t *table_dev, int64_t n_table, int64_t *dst);
/* Brute-force exact path for a list of users: all n_items fp64 scores, two-class key
 * (unseen above seen, then score; the reference's downvote semantics, models.py:510-519), top-k.
 * Outputs are compact [n_rows x topk] (row r belongs to user rows_dev[r]).
 * work: pk_exact_work_bytes(n_rows, n_items) bytes. */
int64_t pk_exact_work_bytes(int32_t n_rows, int64_t n_items);
int pk_score_exact_rows_f64(void *stream, int32_t n_rows, const int32_t *rows_dev, int64_t n_items,
                            int32_t K, const double *V_dev, int64_t ldv, const double *E_dev, int64_t lde,
                            const int64_t *seen_ptr_dev, const int32_t *seen_idx_dev, int32_t topk,
                            int64_t *out_idx_dev, double *out_score_dev, void *work_dev);
/* The same rows for a DEVICE-side list (pk_flag_compact output): users list_dev[0 .. *count_dev); results go to the rows
 * of those users in the [n_users x topk] outputs.  n_wg workgroups walk the list (work >= pk_exact_work_bytes(n_wg,
 * n_items)); the list never visits the host, so a scoring pass needs no synchronisation. */
int pk_score_exact_list_f64(void *stream, int32_t n_wg, const int32_t *list_dev, const int32_t *count_dev, int64_t n_items,
                            int32_t K, const double *V_dev, int64_t ldv, const double *E_dev, int64_t lde,
                            const int64_t *seen_ptr_dev, const int32_t *seen_idx_dev, int32_t topk, int64_t *out_idx_dev,
                            double *out_score_dev, void *work_dev);
/* Evaluation support (models.py:408-485, evaluation.py:24-44 `build_rank_matrix` restricted to the holdout):
 * rank_out[e] = 1-based position of hold_item[e] in row hold_row[e] of the device-resident [n_users x topk]
 * recommendation array, 0 if absent.  Every hit/rank metric is a reduction of this holdout-sized vector. */
int pk_eval_ranks(void *stream, int64_t n_holdout, const int64_t *recs_dev, int32_t topk,
                  const int64_t *hold_row_dev, const int64_t *hold_item_dev, int32_t *rank_out_dev);
/* evaluate() on the device (models.py:408-485; formulas of recommender/evaluation.py:90-253).  One row of 16 doubles
 * per test user: tp, fp, tn, fn (evaluation.py:176-205), precision, recall, fallout, specifity, miss_rate (:208-236),
 * arhr, mrr (:108-118), map (:120-133), ndcg, ndcl (:136-173), valid recommendations, holdout items.
 * recs: [n_users x ld] int64 (negative = padding); the holdout is CSR-like over the SAME user rows (hold_ptr int64
 * [n_users + 1], items in the id space of recs, hold_rel = feedback or NULL (= ones: ignore_feedback), hold_pos =
 * feedback >= switch_positive per entry or NULL (no positive/negative split)).  pk_eval_reduce adds the columns up
 * in a fixed order (work >= pk_eval_reduce_work_bytes); the means are sums / n_users. */
int32_t pk_eval_cols(void);
int pk_eval_user_metrics(void *stream, int64_t n_users, int32_t topk, const int64_t *recs_dev, int64_t ld,
                         const int64_t *hold_ptr_dev, const int64_t *hold_item_dev, const double *hold_rel_dev,
                         const unsigned char *hold_pos_dev, double not_rated_penalty, double switch_positive,
                         int32_t alternative, double *out_dev);
int64_t pk_eval_reduce_work_bytes(int64_t n_users);
int pk_eval_reduce(void *stream, int64_t n_users, const double *table_dev, double *sums_dev, void *work_dev);
/* coverage (evaluation.py:239-242): count_dev[0] = number of distinct ids in [0, n_bins) (+1 when any id is negative: np.unique
 * counts the padding constant too); flags_dev int32[n_bins + 1] scratch */
int pk_unique_count_i64(void *stream, int64_t n, const int64_t *ids_dev, int64_t n_bins, int32_t *flags_dev,
                        int64_t *count_dev);
/* Dense fp64 score rows (kept for `slice_recommendations`/`_user_scores`, models.py:277-291):
 * out[r, :] = E[r, :] V^T for r in [0, n_rows). */
int pk_dense_scores_f64(void *stream, int32_t n_rows, int64_t n_items, int32_t K, const double *V_dev,
                        int64_t ldv, const double *E_dev, int64_t lde, double *out_dev, int64_t ldo);
/* Top-k columns of dense score rows by (score descending, column ascending): get_topk_elements / topsort on an array
 * (models.py:488-491, 561-563; where scores tie the reference's argpartition order is implementation-defined).
 * out_idx_dev int64 [n_rows x topk]. */
int pk_topk_rows_f64(void *stream, int64_t n_rows, int64_t n_cols, const double *scores_dev, int64_t ld, int32_t topk,
                     int64_t *out_idx_dev);

/* ------------------------------------------------------------------------------------------
 * K5.  Sparse tensor-times-matrix (CoFFee / HOOI).
 * Replaces numba `dttm_seq` / `dttm_par` (lib/sparse.py:203-234) called from `ttm3d_seq`
 * (lib/tensor.py:7-19):  res[i0, j, k] += val * u[i1, j] * v[i2, k].
 * nnz are pre-sorted by the output mode by the host layer; tasks as in pk_spmm_csr_f64, with
 * idx1/idx2 the two contracted-mode indices of every nnz and `vals` optional (NULL = ones,
 * data.py:805).  res is [n0 x (ra*rb)] row-major.
 * ------------------------------------------------------------------------------------------ */
int pk_ttm_f64(void *stream,
               int64_t n_tasks, const int32_t *task_row_dev, const int64_t *task_begin_dev,
               const int64_t *task_end_dev, const int32_t *task_slot_dev,
               int64_t n_long, const int32_t *long_row_dev, const int32_t *long_slot_begin_dev,
               const int32_t *long_slot_end_dev,
               const int32_t *idx1_dev, const int32_t *idx2_dev, const double *vals_dev,
               const double *u_dev, int64_t ldu, int32_t ra, const double *v_dev, int64_t ldv_, int32_t rb,
               double *res_dev, int64_t ldr, double *partial_dev);
/* CoffeeModel.predict_feedback (models.py:1068-1091): for each of n holdout (user, item) pairs the index of the feedback
 * level f with the largest reconstructed score  sum_abc core[a, b, c] u[user, a] v[item, b] w[f, c]  (first maximum, as
 * np.argmax).  u [n_users x r0], v [n_items x r1], w [L x r2] row-major fp64, core [r0 x r1 x r2] C-ordered, r2 <= 16.
 * pred_dev int64[n]; scores_dev [n x L] or NULL. */
int pk_tucker_predict_f64(void *stream, int64_t n, const int64_t *users_dev, const int64_t *items_dev,
                          const double *u_dev, int64_t ldu, const double *v_dev, int64_t ldv_, const double *w_dev,
                          int64_t ldw, const double *core_dev, int32_t r0, int32_t r1, int32_t r2, int32_t L,
                          int64_t *pred_dev, double *scores_dev);


/* ------------------------------------------------------------------------------------------
 * Coarse entry points: what a host in ANY language binds to replace the two hot calls of the
 * reference without re-writing the solver or the scoring pipeline — `SVDModel.build` (models.py:835-855: svds of the
 * training matrix) and `get_recommendations` (models.py:391-405 with 857-861, 494-519, 488-491).  Host pointers in,
 * caller-allocated HOST buffers out; device memory lives behind opaque handles owned by the library; every call
 * on one context is serialised by a mutex inside it (the reference's thread pool, models.py:374-382, is safe),
 * distinct contexts are independent; no torch, no Python.  Errors: negative return code + pk_ctx_error(ctx).
 * (The kernel-granular functions above stay the interface of polara_amd's own Python layer, which keeps its device
 * arrays in torch tensors and shards over RCCL; the coarse calls are single-GPU.)
 * ------------------------------------------------------------------------------------------ */
typedef struct pk_ctx pk_ctx;   /* one device + one stream + one mutex */
typedef struct pk_mat pk_mat;   /* a CSR matrix resident on the device with its task plan (and, lazily, its user-blocked transpose) */
typedef struct pk_build_stats {
    int32_t outer;              /* outer iterations (Rayleigh-Ritz + filter) */
    int32_t gramian_steps;      /* applications of A^T A to the block */
    int32_t block;              /* block width */
    int32_t converged;          /* 1 = every leading pair met tol * sigma_1^2 */
    double final_rel_residual;  /* worst ||A^T A v - sigma^2 v|| / sigma_1^2 of the leading k pairs */
} pk_build_stats;
int pk_ctx_create(int32_t device, pk_ctx **ctx_out);
void pk_ctx_destroy(pk_ctx *ctx);
const char *pk_ctx_error(pk_ctx *ctx);
/* host CSR (indptr int64[n_rows + 1], indices int32, values f32 | f64) -> device matrix.  The rows of a TRAINING matrix
 * are users (models.py:160-177); for pk_score_topk the rows are the test users' known interactions with explicit
 * zeros kept — they contribute nothing to the fold-in but still count as seen (models.py:198-203, 494-519). */
int pk_mat_from_csr(pk_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *indptr_host,
                    const int32_t *indices_host, const void *values_host, int32_t val_kind, pk_mat **mat_out);
/* host COO triplets, any order, duplicates summed (what `coo_matrix(...).tocsr()` does, models.py:172-175); entry i has
 * row rows_host[i * idx_stride], column cols_host[i * idx_stride] (idx_stride = 2 with cols = rows + 1 reads the
 * interleaved [nnz x 2] index array of `to_coo`, data.py:794-817, as it is) */
int pk_mat_from_coo(pk_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *rows_host,
                    const int64_t *cols_host, int64_t idx_stride, const void *values_host, int32_t val_kind,
                    pk_mat **mat_out);
void pk_mat_free(pk_ctx *ctx, pk_mat *mat);
int64_t pk_mat_nnz(const pk_mat *mat);   /* stored entries (after duplicate sums) */
/* Top-k singular triplets of A (models.py:841-853): sigma_out[k] descending, V_out[n_cols * k] COLUMN-major (the
 * F-ordered `vh.T` of models.py:849: column j = right singular vector j), U_out[n_rows * k] column-major or NULL
 * (`return_factors`, models.py:841,853).  block = 0: k + max(14, 0.28 k) rounded up to 8; tol <= 0: 1e-12 (relative to
 * sigma_1^2, on ||A^T A v - sigma^2 v||); max_outer <= 0: 200.  Returns PK_E_NOCONV (with the best available factors
 * written) when it stops unconverged — the reference's ARPACK raises there. */
int pk_svd_build(pk_ctx *ctx, pk_mat *A, int32_t k, int32_t block, double tol, int32_t max_outer, uint64_t seed,
                 double *sigma_out, double *V_out, double *U_out, pk_build_stats *stats_out);
/* The same build with the USERS sharded over `comm->world` ranks — one process (or thread) and one pk_ctx per GPU, the
 * multi-device form of the coarse API (SURVEY 8b sketched `pk_ctx_create(device_ids, n_dev)`; with one process per GPU
 * the devices meet through a communicator instead).  `A_local` holds this rank's rows (users) of the training matrix,
 * all n_cols columns.  What is exchanged is exactly north_star's "all-reduce only for the Gramian step": after every
 * Z = A_p^T (A_p X) the [n_cols x l] block is summed over ranks, and so is the l x l Rayleigh-Ritz matrix (A_p X)^T (A_p X);
 * everything on the item side (orthonormalisation, filter recurrences, eigen-decompositions, locking decisions) is
 * computed identically on every rank from all-reduced data and the same seed.  sigma_out / V_out are global and equal on
 * all ranks; U_out (or NULL) holds this rank's rows.  Scoring needs no collective: every rank calls pk_score_topk /
 * pk_serving_* on its own users with the replicated V.
 * The library does not link a collective library: the host passes its communicator as a callback — with RCCL
 *     int ar(void *user, void *buf, int64_t n, void *stream) {
 *         return ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, (ncclComm_t)user, (hipStream_t)stream) == ncclSuccess ? 0 : -1; }
 * `stream` is the HIP stream the call must be ordered on (enqueue there, or block until done): the context's stream, or
 * — when the exchange of a Gramian product is long enough to be worth hiding (modelled all-reduce >= 0.4 ms; context option
 * "dist_overlap") — a side stream of the library for the FIRST of the product's two column panels, whose sum then travels
 * while the second panel is computed (the same rule and the same two panels as polara_amd/solver.py::ItemRows.product).
 * Every rank issues its calls in the same order. */
typedef struct pk_comm {
    int32_t rank, world;
    int (*allreduce_sum_f64)(void *user, void *buf_dev, int64_t count, void *stream);   /* in place, DEVICE pointer; 0 = ok */
    void *user;
} pk_comm;
int pk_svd_build_sharded(pk_ctx *ctx, pk_mat *A_local, const pk_comm *comm, int32_t k, int32_t block, double tol,
                         int32_t max_outer, uint64_t seed, double *sigma_out, double *V_out, double *U_out_local,
                         pk_build_stats *stats_out);
/* The k leading eigenpairs of a small dense symmetric PSD matrix T [n x n] (DEVICE, row-major, leading dimension ldt) — the
 * projected problem Q^T (A^T A) Q of the block Lanczos build (polara_amd/solver.py::_block_lanczos; inside the reference's
 * `svds` ARPACK solves the same kind of projected problem, models.py:844).  Filtered subspace iteration with locking on
 * the operator T, block width l (k <= l <= min(n, 1024)), run on `stream` (NULL: the context's) with temporaries from the
 * context's pool.  X0_dev: orthonormal start block [x0_rows x l] (rows x0_rows..n-1 are taken as zero: the warm start from
 * the pairs of a leading principal submatrix), or NULL = the first l unit vectors.  Convergence: ||T y - theta y|| <= tol *
 * theta_1 for the k leading pairs.  Outputs: basis_out_dev [n x l] (locked vectors, then the active block; column j belongs
 * to lam_out_host[j]), res_out_host[0 .. counts[1]) = residual norms of the active block, counts_out[5] = {locked vectors,
 * width of the active block, converged, outer iterations, products with T}.  Synchronises `stream` before it returns.
 * Round 6: the filter runs in SEGMENTS with a CholeskyQR between them and one Rayleigh-Ritz step per round (two host reads);
 * the iteration with locking takes over when that hands over (degenerate bounds, a Cholesky breakdown, stagnation).
 * lam0_host (or NULL): the Ritz values of the l columns of X0_dev when those are the pairs of a leading principal submatrix of
 * T (they keep their Rayleigh quotients), r0_rel: the worst relative residual of the first k of them w.r.t. THIS T (< 0:
 * unknown) — a warm look then needs no Rayleigh-Ritz step before its filter. */
int pk_sym_eig_topk_f64(pk_ctx *ctx, void *stream, int32_t n, const double *T_dev, int64_t ldt, int32_t k, int32_t l,
                        const double *X0_dev, int64_t ldx0, int32_t x0_rows, double tol, int32_t max_outer, uint64_t seed,
                        double *basis_out_dev, int64_t ldb, double *lam_out_host, double *res_out_host, int32_t *counts_out,
                        const double *lam0_host, double r0_rel);
/* What a host may choose about the builds of a context — explicit calls, not environment variables (round 6):
 *   "svd_method"   0 = the cost model (solver.py::choose_method restated), 1 = block Lanczos, 2 = filtered subspace iteration
 *   "krylov_block" 0 = the cost model (solver.py::choose_krylov_block restated), else the width of a Krylov block (<= block)
 *   "dist_overlap" two-panel exchange of a sharded product: 0 = never, 1 = by the cost model, 2 = whenever possible (tests)
 *   "hooi_ttm"     1 = pk_hooi runs the per-entry mode products (pk_ttm_f64) instead of the factored form
 *   "time_spmm"    1 = HIP events around every SpMM launch of the context's builds, read with pk_ctx_spmm_timings
 * Unknown names and values out of range are PK_E_INVALID. */
int pk_ctx_set_option(pk_ctx *ctx, const char *name, int32_t value);
/* The SpMM launches recorded since the last call (option "time_spmm"; bench.py's roofline of the build): waits for them,
 * writes min(count, cap) records — ms_out[i] = duration in ms, meta_out[6 i ..] = {rows written, rows gathered from, stored
 * entries, columns, bytes per stored value, bytes per element of the dense block} — forgets them all, returns the count. */
int64_t pk_ctx_spmm_timings(pk_ctx *ctx, double *ms_out, int64_t *meta_out, int64_t cap);
/* The recurrence of the block Lanczos build for a host layer that keeps the looks (polara_amd/solver.py::_block_lanczos; the
 * reference's `svds` call, models.py:841-844, runs ARPACK's recurrence in its place).  All three run on `stream` (a
 * hipStream_t; 0 = the null stream) with temporaries from the context's pool — use ONE stream per context.
 * pk_mat_wrap_device: a NON-OWNING matrix over CSR arrays that already live in HBM (indptr int64[n_rows + 1], indices int32,
 * values f32 | f64 — they must outlive the handle), with its task plan and the user-blocked transpose image sized for
 * products with `block_cols`-column blocks (0: 64).  pk_mat_free releases the handle, never the borrowed arrays.
 * pk_lanczos_steps: steps j0 + 1 .. j0 + m of the recurrence  W = A^T A Q_j;  T[:, j] = Q^T W;  Q_(j+1) R = W - Q T[:, j]
 * (shifted CholeskyQR3, re-projected against the whole basis in every pass) on the caller's buffers: Q_dev [n_cols x >=
 * (j0 + m + 1) b] with block j in columns [(j - 1) b, j b), T_dev [>= (j0 + m) b square] (block column j AND its mirror
 * image are written), S_out_dev [b x b] = W_perp^T W_perp of the LAST step run (the coupling behind the residual
 * estimates of the Ritz pairs), flags_dev[2] += Cholesky verdicts / max= distance of a last pass's Gram matrix from I.
 * last_closes: the last step of the call does not compute a next block (the space is full).  No host synchronisation.
 * rounded != 0: both sparse products of these steps gather fp32 images of their dense blocks (half the bytes per gathered
 * row, fp64 accumulation: a product rounded to ~6e-8 of its norm) and only the BAND of the block column of T is written —
 * for the late steps of a build only, when the pairs are within ~1e-7 of convergence (the error of a rounded product
 * enters a pair's residual times the pair's coefficients in that block; solver.py: products='relaxed').
 * pk_gramian_apply_f64: Z = A^T (A X), X / Z [n_cols x nc] with leading dimensions ldx / ldz (the verification product). */
int pk_mat_wrap_device(pk_ctx *ctx, void *stream, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *indptr_dev,
                       const int32_t *indices_dev, const void *values_dev, int32_t val_kind, int32_t block_cols,
                       pk_mat **mat_out);
int pk_lanczos_steps(pk_ctx *ctx, void *stream, pk_mat *A, int32_t b, int32_t j0, int32_t m, int32_t last_closes,
                     double *Q_dev, int64_t ldq, double *T_dev, int64_t ldt, double *S_out_dev, double *flags_dev,
                     int32_t rounded);
int pk_gramian_apply_f64(pk_ctx *ctx, void *stream, pk_mat *A, int32_t nc, const double *X_dev, int64_t ldx, double *Z_dev,
                         int64_t ldz);
/* The two halves of step j (j >= 1) for a host layer that shards the USERS itself — one process per GPU, item side replicated:
 * pk_lanczos_products writes W = A_p^T (A_p Q_j) of THIS rank's rows into W_out_dev [n_cols x b] (contiguous); the host sums W
 * over the ranks (ONE all-reduce of n_cols x b per step, e.g. ncclAllReduce on `stream`); pk_lanczos_orth then does on every
 * rank what pk_lanczos_steps does after its products: block column j of T and its mirror image, the next block by shifted
 * CholeskyQR3 re-projected in every pass, S_out_dev and flags_dev as there (`rounded`: band-only block column, see there). */
int pk_lanczos_products(pk_ctx *ctx, void *stream, pk_mat *A, int32_t b, int32_t j, const double *Q_dev, int64_t ldq,
                        double *W_out_dev, int32_t rounded);
int pk_lanczos_orth(pk_ctx *ctx, void *stream, int64_t n_items, int32_t b, int32_t j, int32_t last_closes, double *Q_dev, int64_t ldq,
                    double *T_dev, int64_t ldt, const double *W_dev, double *S_out_dev, double *flags_dev, int32_t rounded);
/* the context's HIP stream (hipStream_t as void*): what a pk_comm callback is handed, for hosts that create the
 * communicator's work on it */
void *pk_ctx_stream(pk_ctx *ctx);
/* Recommendations for the users of T (rows: test users, columns: the n_items items): scores = (T V) V^T, seen items
 * (every stored entry of T) pushed below all unseen ones when filter_seen, topk item ids per user by descending score
 * -> out_idx[n_users * topk] int64 row-major (the `top_recs` of models.py:400-405; -1 pads a user with fewer than topk
 * candidates when filter_seen = 0 is impossible: with filter_seen seen items re-enter after the unseen ones exactly as
 * models.py:510-519 orders them).  V_host: [n_items * K] column-major.  out_scores: fp64 scores of those items or NULL. */
int pk_score_topk(pk_ctx *ctx, int64_t n_items, int32_t K, const double *V_host, pk_mat *T, int32_t topk,
                  int32_t filter_seen, int64_t *out_idx, double *out_scores);
/* The same with the model-side state kept between calls — what a RecommenderModel keeps between get_recommendations()
 * calls (models.py:391-405: the factors and the test data do not change from call to call): pk_serving_create orders the
 * catalogue by factor norm, uploads the factors and their images, renames and re-sorts the rows of T and plans them;
 * pk_serving_score runs one scoring pass (any topk / filter_seen) and writes the caller's item ids; pk_score_topk =
 * create + score + free.  The handle belongs to `ctx`; T may be freed after pk_serving_create. */
typedef struct pk_serving pk_serving;
int pk_serving_create(pk_ctx *ctx, int64_t n_items, int32_t K, const double *V_host, pk_mat *T, pk_serving **out);
int pk_serving_score(pk_ctx *ctx, pk_serving *serving, int32_t topk, int32_t filter_seen, int64_t *out_idx,
                     double *out_scores);
void pk_serving_free(pk_ctx *ctx, pk_serving *serving);
/* Tucker / HOOI of a sparse 3-way tensor (`CoffeeModel.build` -> `hooi`, models.py:1009-1024, lib/tensor.py:37-96):
 * idx_host [nnz x 3] row-major (user, item, feedback level), vals_host or NULL (= ones, data.py:805), shape[3],
 * mlrank[3].  u1_start [n1 x r1] / u2_start [n2 x r2] (row-major, orthonormal columns): the start the reference draws
 * from NumPy's RandomState + LAPACK QR (tensor.py:57-63) — pass it to follow the reference's iteration exactly; both
 * NULL: a seeded device generator.  Outputs (host, row-major, C order like the reference's factors): u0 [n0 x r0],
 * u1 [n1 x r1], u2 [n2 x r2], core [r0 x r1 x r2], trace_out[num_iters] = the core norm after every iteration
 * (tensor.py:82-88), *iters_out = iterations run (stops when the core's growth falls below growth_tol). */
int pk_hooi(pk_ctx *ctx, int64_t nnz, const int64_t *idx_host, const double *vals_host, const int64_t *shape,
            const int32_t *mlrank, int32_t num_iters, double growth_tol, const double *u1_start, const double *u2_start,
            uint64_t seed, double *u0_out, double *u1_out, double *u2_out, double *core_out, double *trace_out,
            int32_t *iters_out);

#ifdef __cplusplus
}
#endif
#endif /* POLARA_HIP_H */
