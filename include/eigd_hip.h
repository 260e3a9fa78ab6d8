/*
 * eigd_hip.h -- C ABI of libeigd_hip.so, the MI355X (gfx950) implementation of the
 * numerical layer under smdogroup/eigd's adjoint eigenvector-derivative path.
 *
 * The reference is pure Python; every flop of its hot path is reached through
 * scipy/numpy natives.  Each entry point below replaces one of those call-site
 * classes (citations: file:line in the reference, eigd/eigenvector_derivatives.py
 * unless another file is named).  The Python host layer (package eigd_amd) binds
 * these with ctypes and keeps eigd's own class/function surface on top.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (EIGD_E_*); the message of
 *     the last error on the calling thread is eigd_last_error().
 *   - `dptr` arguments are DEVICE pointers obtained from eigd_malloc; `h*`
 *     arguments are HOST pointers borrowed for the duration of the call.
 *   - dense blocks are row-major "n x k, leading dimension ld" (numpy C order), so a
 *     numpy (n, N) array is one memcpy away from its device image.
 *   - a "stack" is `ns` such blocks at a fixed element stride (slab stride): the
 *     Krylov histories W, Z of sibk/pgmres and the Lanczos basis V (k = 1).
 *   - one ctx = one device + one stream; calls on one ctx are ordered; functions
 *     that return host results synchronise the stream before returning.
 *   - thread-compatible: distinct ctx objects may be used from distinct threads.
 */
#ifndef EIGD_HIP_H
#define EIGD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EIGD_OK 0
#define EIGD_E_INVALID (-1)  /* bad argument / shape mismatch            */
#define EIGD_E_HIP (-2)      /* HIP runtime error (no device, OOM, ...)  */
#define EIGD_E_NOTSPD (-3)   /* zero / non-finite pivot: singular shift  */
#define EIGD_E_INTERNAL (-4) /* violated internal invariant              */

typedef struct eigd_ctx eigd_ctx;
typedef struct eigd_mat eigd_mat;           /* device CSR matrix                     */
typedef struct eigd_symbolic eigd_symbolic; /* host ordering + symbolic factorisation */
typedef struct eigd_factor eigd_factor;     /* device numeric factor                 */
typedef struct eigd_lane eigd_lane;         /* extra sweep workspaces of a factor, bound to another stream */

const char* eigd_last_error(void);
int eigd_version(void);

/* ---- context, memory, transfers ---------------------------------------- */
int eigd_device_count(int* count);
int eigd_ctx_create(int device, eigd_ctx** out);
/* a second context (own stream, events, scratch) on the parent's device: independent mode groups of the
 * lock-step solvers run on forked contexts so that their launch chains overlap */
int eigd_ctx_fork(eigd_ctx* parent, eigd_ctx** out);
int eigd_ctx_destroy(eigd_ctx* ctx);
/* make the context's device the calling host thread's current HIP device (call once at the start of every worker
 * thread that drives a context: allocations and launches of that thread then land on the right GPU) */
int eigd_ctx_make_current(eigd_ctx* ctx);
int eigd_sync(eigd_ctx* ctx);
int eigd_malloc(eigd_ctx* ctx, size_t bytes, void** dptr);
int eigd_free(eigd_ctx* ctx, void* dptr);
int eigd_memset(eigd_ctx* ctx, void* dptr, int value, size_t bytes);
int eigd_h2d(eigd_ctx* ctx, void* dptr, const void* hsrc, size_t bytes);
int eigd_d2h(eigd_ctx* ctx, void* hdst, const void* dptr, size_t bytes);
int eigd_d2d(eigd_ctx* ctx, void* ddst, const void* dsrc, size_t bytes);
/* page-locked host memory for callers that keep the reference's numpy call surface (solve_adjoint(Phib) -> psi,
 * add_total_derivative(..., psi, ...), 1988-2134 / 2167-2207): eigd_h2d / eigd_d2h run at the direct-DMA rate when the
 * host pointer was obtained from eigd_host_alloc or covered by eigd_host_register */
int eigd_host_alloc(size_t bytes, void** hptr);
int eigd_host_free(void* hptr);
int eigd_host_register(void* hptr, size_t bytes);
int eigd_host_unregister(void* hptr);
int eigd_mem_info(eigd_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
/* HIP-event stopwatch on the ctx stream (bench.py roofline timing) */
int eigd_timer_start(eigd_ctx* ctx);
int eigd_timer_stop_ms(eigd_ctx* ctx, double* ms);

/* ---- CSR SpMV / SpMM ------------------------------------------------------
 * replaces scipy _sparsetools csr_matvec / csr_matvecs behind every `A @ x`,
 * `B @ X` (call sites 255, 260, 263-265, 519, 637, 800, 843-844, 1004-1005,
 * 1173, 1190-1192, 1250-1252, 1500, 1503).  Y = alpha * A X + beta * Y on
 * n x k row-major blocks; k = 1 is the LDS-staged CSR-stream SpMV kernel, whose
 * per-row summation order is scipy's (bit-identical result for alpha=1, beta=0). */
int eigd_csr_upload(eigd_ctx* ctx, int n, int64_t nnz, const int32_t* hindptr, const int32_t* hindices,
                    const double* hdata, eigd_mat** out);
/* rectangular n x ncols matrix (x has ncols rows, y has n): the gather / averaging / filter maps of the design-variable
 * chain (examples/node_filter.py:160-217; element averages and their transposes, buckling.py:209-213, 866-870) */
int eigd_csr_upload_rect(eigd_ctx* ctx, int n, int ncols, int64_t nnz, const int32_t* hindptr, const int32_t* hindices,
                         const double* hdata, eigd_mat** out);
int eigd_csr_update_values(eigd_mat* A, const double* hdata);
/* the same with the values already on the device (e.g. from eigd_assemble) */
int eigd_csr_update_values_dev(eigd_mat* A, const double* dvals);
int eigd_mat_free(eigd_mat* A);
int eigd_spmm(eigd_mat* A, const double* dX, int ldx, double* dY, int ldy, int k, double alpha, double beta);
/* same product enqueued on another context of the same device */
int eigd_spmm_on(eigd_ctx* ctx, eigd_mat* A, const double* dX, int ldx, double* dY, int ldy, int k, double alpha,
                 double beta);

/* ---- sparse shift-invert factorisation ------------------------------------
 * replaces SuperLU behind SpLuOperator (11-23): splu(mat) -> analyse + factor,
 * lu.solve(x) -> eigd_factor_solve.  The matrix must be symmetric; it is given as
 * full CSR (== CSC).  Ordering (nested dissection on the compressed graph),
 * elimination tree, supernodes and all index maps are host C++ (eigd_symbolic);
 * the multifrontal LL^T numeric factorisation and the level-scheduled multi-RHS
 * triangular sweeps are gfx950 kernels.                                          */
int eigd_symbolic_create(int n, const int32_t* hindptr, const int32_t* hindices, int leaf_size, int panel_width,
                         eigd_symbolic** out);
/* same, with dof coordinates (n x dim row-major, dim <= 3) as an ordering hint: geometric nested dissection
 * (straight separators on structured meshes) instead of the purely algebraic level-structure dissection */
int eigd_symbolic_create_geom(int n, const int32_t* hindptr, const int32_t* hindices, int leaf_size, int panel_width,
                              int dim, const double* hcoords, eigd_symbolic** out);
int eigd_symbolic_free(eigd_symbolic* s);
/* sizes: [0]=n [1]=nfronts [2]=nlevels [3]=nnz(L) incl. diagonal blocks [4]=front buffer doubles
 *        [5]=sum of front dimensions [6]=max front dimension [7]=border entries [8]=lower nnz of A
 *        [9]=number of (level, step) launches [10]=flops of the numeric factorisation [11]=max columns in a front */
int eigd_symbolic_sizes(eigd_symbolic* s, int64_t* out, int nout);
/* copy-out of the symbolic arrays (tests emulate the numeric phase in numpy from these) */
int eigd_symbolic_get_i32(eigd_symbolic* s, const char* name, int32_t* out, int64_t cap);
int eigd_symbolic_get_i64(eigd_symbolic* s, const char* name, int64_t* out, int64_t cap);

int eigd_factor_create(eigd_ctx* ctx, eigd_symbolic* s, const double* hdata /* CSR values, full matrix */,
                       eigd_factor** out);
int eigd_factor_refactor(eigd_factor* f, const double* hdata);
/* the same with the CSR values already on the device (e.g. from eigd_assemble): no host round trip */
int eigd_factor_refactor_dev(eigd_factor* f, const double* dvals);
int eigd_factor_free(eigd_factor* f);
/* X (n x k row-major, ld) <- alpha * M^{-1} X ; any k >= 1 (processed in column blocks of <= 32) */
int eigd_factor_solve(eigd_factor* f, double* dX, int ldx, int k, double alpha);
/* out of place: Out <- alpha * M^{-1} In (In is not modified; In == Out is allowed) -- Z[:, kp] = factor(W[:, kp]) (1248) */
int eigd_factor_solve_to(eigd_factor* f, const double* dIn, int ldin, double* dOut, int ldout, int k, double alpha);
/* concurrent sweeps on one factor: a lane owns its own vector workspaces and runs on `ctx`'s stream */
int eigd_factor_lane_create(eigd_factor* f, eigd_ctx* ctx, eigd_lane** out);
int eigd_factor_lane_free(eigd_lane* lane);
int eigd_factor_lane_solve_to(eigd_lane* lane, const double* dIn, int ldin, double* dOut, int ldout, int k, double alpha);
/* stats: [0]=nnz(L) [1]=device bytes held [2]=flops of the numeric factorisation [3]=number of fronts
          [4]=negative pivots of P M P^T = L S L^T (inertia: eigenvalues of the pencil below the shift; 0 = SPD)
          [5]=static pivots [6]=planes of the sweeps' vector workspace (carry planes, + 1 where the right-hand sides of
          levels with thousands of workgroups are pre-assembled) */
int eigd_factor_stats(eigd_factor* f, double* out, int nout);
/* bytes of L streamed by one k-column solve (algorithmic, for the roofline) */
int eigd_factor_solve_bytes(eigd_factor* f, int k, double* bytes);

/* ---- dense panel kernels ---------------------------------------------------
 * replace the BLAS calls numpy makes for the tall-skinny products, projections,
 * Gram-Schmidt sweeps and axpy soups of the path.                               */

/* C (ku x kx, HOST, row-major) = U^T X.  U(r, a) = dU[r*rsu + a*csu] (row-major block: rsu=ld,
 * csu=1; Lanczos basis stored as k=1 stack: rsu=1, csu=slab stride).  X n x kx row-major.
 * sites: V.T @ Phib 502/510/620, Phi.T @ R 810/989/1180, W.T @ R 1283, _project 28      */
int eigd_gemm_tn(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int64_t rsu, int64_t csu, const double* dX,
                 int ldx, double* hC);
/* X (n x kx) = beta * X + alpha * U C, C (ku x kx) given on the HOST.
 * sites: V @ Y0 1648, B @ V @ (...) 519, Z @ y 1028/1277/1301, Vb @ (...) 678, _project 29   */
int eigd_gemm_nn(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int64_t rsu, int64_t csu, const double* hC,
                 double* dX, int ldx, double alpha, double beta);
/* Out = U C for a basis kept as row-major panels of 64 columns (panel p at dU + p * pstride doubles, leading dimension
 * 64; ku <= 192 basis columns, C (ku x kx) on the HOST, row-major): every basis panel is read once per 80 output columns
 * and the result written once.  Out: panels of opw columns, ostride doubles apart, rows of ldo doubles (opw = ldo = 64:
 * the layout of the basis; opw >= kx, ldo >= kx: a plain row-major block).  Out must not overlap the basis.
 * sites: the restart V <- V S of ARPACK's dsaup2/dseupd behind eigsh_mod (arpack.py:41-56, 438-440), V @ (T Cf) 501-521 */
int eigd_panels_times(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int64_t pstride, const double* hC,
                      double* dOut, int64_t ostride, int opw, int ldo);
/* fused oblique projector X <- X - U (V^T X), coefficient matrix stays on the device (26-30); ku <= 128: the N
 * eigenvectors of the caller plus the further converged pairs the adjoint stage deflates */
int eigd_project(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int ldu, const double* dV, int ldv, double* dX,
                 int ldx);
/* the projector followed by the squared column norms of the result (1257 + 1259 of sibk in one pass over X):
 * dOut (device, kx) receives them; a pinned copy is left for eigd_colnorm2_fetch as after eigd_colnorm2_dev.
 * The update pass is MEASURED: the coefficient pass also forms the column norms of X, and when no coefficient exceeds
 * tol (0: 1e-13) times the norm of its column divided by uscale -- the block was built
 * from projected vectors -- X is left as it is and those norms are the result.  uscale = the largest Euclidean column
 * norm of U (0: unknown, 1 is used): the test then compares the size of the update, |u_a| |C[a][b]|, with |x_b| and does
 * not depend on how B scales against the Euclidean norm.  eigd_project_stats: out[0] = projections measured,
 * out[1] = updates applied since the last call (resets both). */
int eigd_project_norm2(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int ldu, const double* dV, int ldv,
                       double* dX, int ldx, double* dOut, double uscale, double tol);
int eigd_project_stats(eigd_ctx* ctx, int* out);
/* One step of block Gram-Schmidt against a panel of the Lanczos basis with the coefficients KEPT ON THE DEVICE (the
 * restarted block eigensolver that stands in for ARPACK's dsaitr reorthogonalisation, eigenvector_derivatives.py:1908-
 * 1986): C = V^T X (ku x kx, ku, kx <= 64) is written to dC (row stride ldc), then X <- X - U C.  tol > 0: the update
 * is MEASURED -- applied only if some |C[a][b]| > tol * sqrt(dNorm2[b]), dNorm2 (device, kx) being the squared B-norms
 * x_b^T B x_b of the columns (eigd_coldot_dev of X and B X): the relative B-orthogonality of the block against the panel,
 * independent of the scale of B (the second pass of "twice is enough"); dNorm2 = NULL: against the Euclidean norms
 * |x_b| as in eigd_project_norm2 -- and dFlag[0] (device double) receives 1.0 or 0.0 accordingly.  No host
 * synchronisation: the caller fetches the coefficient blocks of a whole step at once. */
int eigd_project_to(eigd_ctx* ctx, int n, int ku, int kx, const double* dU, int ldu, const double* dV, int ldv, double* dX,
                    int ldx, double* dC, int ldc, double tol, double* dFlag, const double* dNorm2);
/* One pass of the B-orthonormalisation of a new Lanczos block WITHOUT a host round trip (the role of ARPACK's dsaitr
 * normalisation, eigenvector_derivatives.py:1908-1986, for blocks of p <= 32 vectors): G = X^T BX on the device, its
 * SVQB transform by a one-wave Jacobi eigensolver (G = D U w U^T D; X <- X D^-1 U w^-1/2, BX likewise, both in place),
 * dC (device, p x p, row-major) <- Cq dC with X_in = X_out Cq (first != 0: dC <- Cq), dFlag[0] (device double) = 1.0
 * when a direction of the block was numerically dependent (w <= 1e-28 max w; set to 0.0 by a clean first pass).
 * update_bx = 0 leaves BX alone (the last pass, when the caller forms B X afresh from the finished block). */
int eigd_svqb_step(eigd_ctx* ctx, int n, int p, double* dX, int ldx, double* dBX, int ldbx, double* dC, int first,
                   double* dFlag, int update_bx);
/* column-wise dots  out[c] = sum_r X[r,c] Y[r,c]  (HOST out, length k); inner products / norms
 * of 1157-1158, 1219, 1233, 1259, 1504, 1537 batched over the modes                     */
int eigd_coldot(eigd_ctx* ctx, int n, int k, const double* dX, int ldx, const double* dY, int ldy, double* hout);
/* the same with the result left on the device (dOut, length k; no host synchronisation) */
int eigd_coldot_dev(eigd_ctx* ctx, int n, int k, const double* dX, int ldx, const double* dY, int ldy, double* dOut);
/* the same dots accumulated in twice the working precision (error-free products and sums, errors carried in a second
 * double): hout[c] + hout[k + c] = sum_r X[r,c] Y[r,c] to ~1e-30 of sum |X Y|.  For the entries of G = -Phi^T Phib
 * (345, 1180) of numerically repeated pairs, whose DIFFERENCE is divided by the eigenvalue gap in xi, eta (373-383) */
int eigd_coldot_dd(eigd_ctx* ctx, int n, int k, const double* dX, int ldx, const double* dY, int ldy, double* hout);
/* out[r,c] = sum_t coef[t*k + c] * X_t[r,c], nterms <= 4, out may alias any X_t
 * (the w-vector / residual assembly of 96-134, 807-809, 1189-1192, axpys 851-860)      */
int eigd_lincomb(eigd_ctx* ctx, int n, int k, double* dOut, int ldo, int nterms, const double* const* dXs,
                 const int* ldxs, const double* hcoef);
/* batched Gram-Schmidt against a stack of ns slabs (each n x k, leading dimension lds, element stride `slab`;
 * a column range of a wider stack is addressed by offsetting dS and keeping lds):
 *   stack_dot : hH[j*k + c] = sum_r S_j[r,c] T[r,c]            (1229, 1255, 1013, 1530)
 *   stack_axpy: T[r,c] += alpha * sum_j S_j[r,c] hH[j*k + c]    (1230, 1256, 1014, 1531, 1277); an exactly zero
 *               coefficient skips its slab entry (which may never have been written) */
int eigd_stack_dot(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, const double* dT,
                   int ldt, double* hH);
int eigd_stack_axpy(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, const double* hH,
                    double* dT, int ldt, double alpha);
/* fused middle of the twice-applied Gram-Schmidt: T += alpha * sum_j S_j hH1[j]; then hH2[j] = S_j . T (ns <= 32) */
int eigd_stack_axpy_dev(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, const double* dH,
                        double* dT, int ldt, double alpha);   /* eigd_stack_axpy with the coefficients on the device */
int eigd_stack_axpy_dot(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, const double* hH1,
                        double* dT, int ldt, double alpha, double* hH2);
/* one Gram-Schmidt step of the lock-step Krylov solvers in a single call (1254-1256, 1012-1014): T -= S (S^T T),
 * then the part of T still along S is measured (second pass over the stack) and subtracted only if it exceeds
 * tol * |S^T T| for some column.  hH (ns x k) receives the coefficients applied; *hpasses the passes over the stack.
 * One host synchronisation instead of one per pass. */
int eigd_stack_cgs2(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, double* dT, int ldt,
                    double tol, double* hH, int* hpasses);
/* Two Krylov steps per Gram-Schmidt pass (the loops of 1254-1260 for the vectors of steps j+1 and j+2 together): dT
 * holds the pair [T1 | T2] (n x 2k, T2 = OP T1 before either is orthogonalised), both blocks meet columns 0..k-1 of the
 * slabs in the same two passes over the stack; hH is ns x 2k.  eigd_pair_orthonormalise finishes the pair on the
 * device: W1 = T1/|T1|, T2 <- T2 - (T1.T2/|T1|^2) T1, W2 = T2/|T2|; dNorm2 = the 2k squared column norms that
 * eigd_project_norm2 left on the device; dOut (device, 4k) = [|T1|^2 | T1.T2 | |T2|^2 | W1.T2] with a pinned copy for
 * eigd_colnorm2_fetch(ctx, hout, 4k); columns with hskip[c] != 0 become zero */
int eigd_stack_cgs2_pair(eigd_ctx* ctx, int n, int k, int ns, const double* dS, int64_t slab, int lds, double* dT, int ldt,
                         double tol, double* hH, int* hpasses);
int eigd_pair_orthonormalise(eigd_ctx* ctx, int n, int k, double* dT, int ldt, const double* dNorm2, double* dW1, int ldw1,
                             double* dW2, int ldw2, const unsigned char* hskip, double* dOut);
/* squared column norms into device memory (no host synchronisation) and the normalisation that consumes them:
 * Out[:, c] = X[:, c] / sqrt(dNorm2[c]), zero where hskip[c] != 0 or the norm is zero (1259-1260, 1233-1234) */
int eigd_colnorm2_dev(eigd_ctx* ctx, int n, int k, const double* dX, int ldx, double* dOut);
/* the same numbers on the host: their copy was issued in stream order behind eigd_colnorm2_dev, this call waits for
 * that copy only (not for kernels enqueued since) */
int eigd_colnorm2_fetch(eigd_ctx* ctx, double* hout, int k);
/* issue that copy for k numbers some other kernel left on the device (eigd_cg_update's residual norms): one copy in flight */
int eigd_colnorm2_publish(eigd_ctx* ctx, const double* dNorm2, int k);
int eigd_scale_inv_norm(eigd_ctx* ctx, int n, int k, const double* dX, int ldx, double* dOut, int ldo,
                        const double* dNorm2, const unsigned char* hskip);
/* ---- sibk in short-recurrence form (csrc/krylov.hip) -------------------------------------------------------------
 * For a positive definite shift the Krylov operator P K factor of sibk (eigenvector_derivatives.py:1246-1252) is
 * self-adjoint in the inner product of the factor and I - alpha_i OP is positive definite in the deflated space: the
 * Arnoldi process with its Gram-Schmidt loops (1254-1260) and the least-squares problem (1262-1270) collapse to
 * conjugate gradients in that inner product (three-term form: residual and solution directly), one multi-column step
 * per factor application, no Krylov history.  All per-mode scalars live in dState (device, eigd_cg_state_rows() rows of
 * 64 doubles, one column per mode: r.z, gam, rho of the previous step, done, tol^2, alpha_i = +-(lam_i - sigma) of
 * 1264-1269, steps taken when the mode met 1275, flag (1: some step was taken with rho = 1 because the recurrence's denominator
 * was not positive in finite precision -- a restart from the current iterate; 2: r.z or r.z - alpha z.y not positive, the
 * column stopped moving: not positive definite), gam and rho of the current step); the caller passes
 * dState + first column of the block it works on.
 *   eigd_cg_coefficients  dNorm2 (device, |r_k|^2 per column from eigd_project_norm2, or null) < tol^2 -> done;
 *                         with z = factor(r_k) (1248) and y = K z (1250-1252): gam = r.z / (r.z - alpha z.y) and
 *                         rho = 1 / (1 - (gam/gam')(r.z / r'.z') / rho')  (rho = 1 when first != 0); dLog (device, may
 *                         be null; like dState offset by the block's first column): row 2 (step - 1) receives gam,
 *                         row 2 (step - 1) + 1 rho of this step (rows of 64 doubles; gam = 0: the column did not move)
 *   eigd_cg_update        r_old <- rho (r - gam (r - alpha y)) + (1 - rho) r_old,
 *                         psi_old <- rho (psi + gam z) + (1 - rho) psi_old  (psi of 1277; finished modes: copies) --
 *                         the caller swaps the roles of the two buffers afterwards.  dPsi null: the residual alone --
 *                         psi_m = sum_k s_k z_k with s_m = alpha_m, s_k = alpha_k + beta_k s_{k+1}, alpha_k = rho_k gam_k,
 *                         beta_k = (rho_{k+1} - 1) alpha_k / alpha_{k+1} is then the caller's to form from the z it kept
 *                         and the log (eigd_stack_axpy).  dNorm2 (device, k; may be null):
 *                         the squared column norms of the new residual, formed in the same pass (for the steps whose
 *                         residual is not projected; eigd_colnorm2_publish hands them to the host) */
int eigd_cg_state_rows(void);
int eigd_cg_coefficients(eigd_ctx* ctx, int n, int k, const double* dZ, int ldz, const double* dR, int ldr, const double* dY,
                         int ldy, const double* dNorm2, double* dState, int step, int first, double* dLog);
/* y = A z (the SpMM of 1250-1252) and eigd_cg_coefficients in one call: on a tiled matrix with 5 <= k <= 32 the two inner
 * products come out of the product's own pass (the tiles' shares in a fixed order), y bit-identical to eigd_spmm */
int eigd_spmm_cg(eigd_ctx* ctx, eigd_mat* A, int k, const double* dZ, int ldz, double* dY, int ldy, const double* dR, int ldr,
                 const double* dNorm2, double* dState, int step, int first, double* dLog);
/* dS (device, nsteps x k) = the coefficients s of psi = sum_j s_j z_j from the log eigd_cg_coefficients wrote (see
 * eigd_cg_update with dPsi null); eigd_stack_axpy_dev forms the sum from the kept z without a host round trip */
int eigd_cg_solution_coefficients(eigd_ctx* ctx, int k, const double* dLog, int nsteps, double* dS);
int eigd_cg_update(eigd_ctx* ctx, int n, int k, const double* dR, int ldr, double* dRold, int ldro, const double* dPsi,
                   int ldpsi, double* dPsiOld, int ldpso, const double* dZ, int ldz, const double* dY, int ldy,
                   const double* dState, int first, double* dNorm2);
/* copy an n x k block between buffers with different leading dimensions / column offsets */
int eigd_copy_block(eigd_ctx* ctx, int n, int k, const double* dSrc, int lds, double* dDst, int ldd);
/* gather columns: Dst[r, j] = Src[r, cols[j]] (compaction of the active modes) */
int eigd_gather_cols(eigd_ctx* ctx, int n, int kdst, const double* dSrc, int lds, const int32_t* hcols, double* dDst,
                     int ldd);
int eigd_scatter_cols(eigd_ctx* ctx, int n, int ksrc, const double* dSrc, int lds, const int32_t* hcols, double* dDst,
                      int ldd);

/* ---- element-wise bilinear forms (device-side derivative callbacks) -------------
 * out[e] += alpha * scale[e] * sum_c w_e(:,c)^T M_e v_e(:,c): the d/d(rho_e) contractions the
 * reference's harness callbacks evaluate with numpy einsums (examples/buckling.py:178-218,
 * 283-340; natural_frequency.py:162-203, 238-284; thermal.py:150-190, 216-246) and that
 * add_eig_total_derivative calls (33-182).  d_edofs: nelem x nd dof list (-1 = constrained), nd <= 24 (6-dof shell
 * facets: the CRM-like configuration, examples/crm.py:295-376).  dMe: one shared nd x nd matrix (per_elem = 0),
 * nelem matrices (per_elem = 1), or one matrix per element TYPE selected by d_etype[e] (per_elem = 2). */
int eigd_elem_bilinear(eigd_ctx* ctx, int nelem, int nd, const int32_t* d_edofs, const double* dMe, int per_elem,
                       const int32_t* d_etype, const double* dscale, const double* dW, int ldw, const double* dV, int ldv,
                       int k, double alpha, double* dOut);

/* ---- element assembly on the device ------------------------------------
 * The COO -> CSR assembly loops of the harnesses (examples/buckling.py:152-176, 220-255;
 * natural_frequency.py:134-160, 205-236; thermal.py:126-148, 192-214): K = sum_e scale[e] * P_e^T Me P_e.
 * eigd_assembler_create analyses the dof lists once (host): CSR pattern of the assembled matrix (rows sorted) and,
 * per stored entry, the list of element entries that add into it.  eigd_assemble then forms the CSR VALUES on the
 * device in that fixed order (no atomics; bitwise reproducible); they feed eigd_csr_update_values_dev and
 * eigd_factor_refactor_dev without visiting the host.  elem_dofs: nelem x nd (host), -1 = constrained dof. */
typedef struct eigd_assembler eigd_assembler;
int eigd_assembler_create(eigd_ctx* ctx, int n, int nelem, int nd, const int32_t* elem_dofs, eigd_assembler** out);
int eigd_assembler_free(eigd_assembler* a);
int eigd_assembler_nnz(eigd_assembler* a, int64_t* nnz);
int eigd_assembler_pattern(eigd_assembler* a, int32_t* hindptr /* n + 1 */, int32_t* hindices /* nnz */);
/* dMe: one shared nd x nd matrix (per_elem = 0), nelem of them (1), or one per element type d_etype[e] (2);
 * dscale: nelem or NULL; dvals: nnz (device); nd <= 24 */
int eigd_assemble(eigd_assembler* a, const double* dMe, int per_elem, const int32_t* d_etype, const double* dscale,
                  double* dvals);
/* element matrices linear in the element's dof values, Me[e] = sum_m (L[m] . u_e) Q[m]: the stress (geometric)
 * stiffness of the linear pre-buckling state (examples/buckling.py:220-255) with L = C0 B at the Gauss points and
 * Q = detJ * (dN dN^T terms).  d_edofs: nelem x nd dofs of the FULL vector du; dL: nterms x nd; dQ: nterms x nd x nd;
 * dOut: nelem x nd x nd (feeds eigd_assemble with per_elem = 1). */
int eigd_elem_linear_matrices(eigd_ctx* ctx, int nelem, int nd, const int32_t* d_edofs, const double* du, int nterms,
                              const double* dL, const double* dQ, double* dOut);

/* transpose of the above with respect to u (path adjoint of the fundamental state, examples/buckling.py:283-316,
 * 930-947, 974-979): dOut[e*nd + a] = alpha * scale[e] * sum_m L[m][a] * sum_c w_e(:,c)^T Q[m] v_e(:,c), with w_e, v_e
 * gathered from the n x k blocks dW, dV through the (constrained = -1) dof list d_edofs; nterms <= 16.  The element
 * vectors are summed into dofs by a CSR product with the incidence matrix (fixed order, reproducible). */
int eigd_elem_linear_adjoint(eigd_ctx* ctx, int nelem, int nd, const int32_t* d_edofs, int nterms, const double* dL,
                             const double* dQ, const double* dscale, const double* dW, int ldw, const double* dV,
                             int ldv, int k, double alpha, double* dOut);

/* elementwise maps of the design-variable chain (examples/node_filter.py:175-203 projection and its
 * derivative, examples/buckling.py:157-160, 207-208 SIMP penalisation and its derivative), dout[i] = map(dx[i]):
 *   kind 0: x^p + c0     kind 1: p x^(p-1) (* dg[i] if dg)     kind 4: c0 x + c1
 *   kind 2: tanh projection with beta = p, eta = c0            kind 3: its derivative times dg[i] */
int eigd_design_map(eigd_ctx* ctx, int64_t n, int kind, double p, double c0, double c1, const double* dx,
                    const double* dg, double* dout);

/* ---- multi-GPU: the df/dx reduction over RCCL / xGMI ---------------------
 * The per-mode adjoint solves are sharded over the ranks (one process per GPU); each rank sums the total-derivative
 * contributions of its own modes (the loop over i of 93-134 / the column sum of 135-180) into a device vector, and
 * ONE all-reduce (sum, fp64) of that vector completes df/dx.  eigd_comm_unique_id is called on rank 0, the 128 bytes
 * travel to the other rank processes through the host, every rank then calls eigd_comm_init (collective).  The
 * all-reduce is in place on a device buffer and ordered on the ctx stream; nranks == 1 needs no id and no RCCL. */
#define EIGD_COMM_ID_BYTES 128
typedef struct eigd_comm eigd_comm;
int eigd_comm_unique_id(void* hid128);
int eigd_comm_init(eigd_ctx* ctx, int nranks, int rank, const void* hid128, eigd_comm** out);
int eigd_comm_destroy(eigd_comm* comm);
int eigd_comm_info(eigd_comm* comm, int* nranks, int* rank);
int eigd_allreduce_sum(eigd_comm* comm, double* dbuf, int64_t len);
/* the same with max: the slowest rank's wall time of a timed region (bench.py) */
int eigd_allreduce_max(eigd_comm* comm, double* dbuf, int64_t len);

#ifdef __cplusplus
}
#endif
#endif /* EIGD_HIP_H */
