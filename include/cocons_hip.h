/*
 * cocons_hip.h -- C ABI of the MI355X-native dense hot path of blasif/cocons.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / Rcpp types.
 * Every entry point names the reference interface it replaces (paths relative to
 * the reference tree).  The R-side `.Call` glue a maintainer would add is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - matrices are column-major (R layout); `locs` is n x 2, `X` is n x p.
 *   - `theta` is a 6 x p row-major table in the order of the reference's
 *     dictionary minus "mean" (R/profile.R:5-7): std.dev, scale, aniso, tilt,
 *     smooth, nugget -- i.e. what `theta_list[-1]` carries by name
 *     (src/cocons_full.cpp:47-54).
 *   - return value: 0 ok; k > 0 = leading minor of order k is not positive
 *     (LAPACK dpotrf convention; the glue maps it to the reference's 1e+06
 *     sentinel or stop("Cholesky error"), R/neg2loglikelihood.R:200-206);
 *     < 0 = bad argument or HIP error, text in cocons_last_error().
 *   - nothing throws, aborts or exits; no HIP call happens at load time
 *     (fork-safe: the device context is created lazily per process).
 *   - threads: the reference's callers are single-threaded R interpreters, one per worker process
 *     (R/optim.R:117-121, 234-235), and that is all the drop-in needs.  Beyond it: ONE handle serves one call
 *     at a time (a second thread entering with the same handle waits until the first call has returned -- every
 *     entry point holds the handle's operation lock); DIFFERENT handles may be created, used and destroyed from
 *     different threads concurrently, the stateless entry points likewise; cocons_last_error() is per thread.
 *     A handle must not be destroyed while another thread is inside a call on it, and a stream installed with
 *     cocons_fit_set_stream is the caller's: the library never launches its own probe kernels on it.
 */
#ifndef COCONS_HIP_H
#define COCONS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define COCONS_P_MAX 32          /* max columns of the design matrix */

typedef struct cocons_fit cocons_fit;   /* opaque: device-resident data of one model fit */

/* library ------------------------------------------------------------------ */
const char *cocons_last_error(void);
int cocons_abi_version(void);
/* number of visible HIP devices (>=0), or <0 on error */
int cocons_device_count(void);

/* ---- stateless covariance assembly (host buffers in, host buffer out) --------
 * replaces .Call("_cocons_cov_rns", theta, locs, x_covariates, smooth_limits)
 *   src/RcppExports.cpp:29-40 -> cov_rns, src/cocons_full.cpp:40-321          */
int cocons_cov_rns(int n, int p, const double *theta, const double *locs,
                   const double *X, const double *smooth_limits, double *out_nxn);
/* replaces .Call("_cocons_cov_rns_classic", theta, locs, x_covariates)
 *   src/RcppExports.cpp:59-69 -> cov_rns_classic, src/cocons_full.cpp:480-594  */
int cocons_cov_rns_classic(int n, int p, const double *theta, const double *locs,
                           const double *X, double *out_nxn);
/* replaces .Call("_cocons_cov_rns_pred", theta, locs, locs_pred, x_covariates,
 *                x_covariates_pred, smooth_limits)
 *   src/RcppExports.cpp:43-56 -> cov_rns_pred, src/cocons_full.cpp:334-471
 * out is m x n column-major, row = prediction location.                        */
int cocons_cov_rns_pred(int n, int m, int p, const double *theta, const double *locs,
                        const double *locs_pred, const double *X, const double *X_pred,
                        const double *smooth_limits, double *out_mxn);
/* replaces .Call("_cocons_sumsmoothlone", x, lambda, alpha)
 *   src/RcppExports.cpp:16-26 -> sumsmoothlone, src/cocons_full.cpp:12-30 (host, O(p)) */
double cocons_sumsmoothlone(const double *x, int len, double lambda, double alpha);

/* ---- sparse/taper covariance entries (first slice) ---------------
 * replaces .Call("_cocons_cov_rns_taper", theta, locs, x_covariates, colindices, rowpointers, smooth_limits)
 *   src/RcppExports.cpp:88-101 -> cov_rns_taper, src/cocons_taper.cpp:151-433
 * and .Call("_cocons_cov_rns_taper_pred", theta, locs, locs_pred, x_covariates, x_covariates_pred,
 *           colindices, rowpointers, smooth_limits)
 *   src/RcppExports.cpp:72-87 -> cov_rns_taper_pred, src/cocons_taper.cpp:17-139.
 * colindices (nnz) / rowpointers (rows + 1) are the spam CSR pattern, 1-BASED as spam stores them, and are
 * only read: the reference shifts its (coerced copies of the) index vectors by -1 in place (:73-74, :211-212),
 * which a caller never sees because spam's integer slots are copied on coercion to NumericVector.
 * entries[w] receives the covariance of stored entry w (row-wise order).  The tapering itself and the
 * sparse Cholesky (R/neg2loglikelihood.R:20-108) stay with the caller.                              */
int cocons_cov_rns_taper(int n, int p, const double *theta, const double *locs, const double *X,
                         const double *smooth_limits, int nnz, const int *colindices, const int *rowpointers,
                         double *entries);
int cocons_cov_rns_taper_pred(int n, int m, int p, const double *theta, const double *locs,
                              const double *locs_pred, const double *X, const double *X_pred,
                              const double *smooth_limits, int nnz, const int *colindices,
                              const int *rowpointers, double *entries);

/* ---- taper objective through the dense factorisation (second slice) ----------------
 * replaces the body of GetNeg2loglikelihoodTaper (R/neg2loglikelihood.R:20-53): ref_taper@entries *
 * cov_rns_taper(...), update.spam.chol.NgPeyton, determinant, forwardsolve.  The handle takes what is constant
 * over the optimisation -- locs, x_covariates, z (n x r), smooth.limits and ref_taper's pattern (colindices /
 * rowpointers, 1-based, symmetric, diagonal stored) with its entries -- and cocons_neg2loglik_dense on it
 * returns  sum_k [ n log 2 pi + 2 sum log diag chol(S) + resid_k' S^-1 resid_k ],  S = taper o cov_rns_taper(theta):
 * spam's value, from a DENSE factorisation of S (zeros stored) -- for n^2 doubles within the device's memory.
 * `parts` as for the dense handle (log-det half, quadratic forms), from which the caller forms
 * GetNeg2loglikelihoodTaperProfile (:73-108, with theta$std.dev[1] = 0).  cocons_neg2loglik_batch pipelines its
 * points over clones of such a handle as for a dense one, cocons_predict_taper is its prediction core; every other
 * fit entry point refuses it.  NULL + cocons_last_error() on failure.                                                */
cocons_fit *cocons_fit_create_taper(int n, int p, int r, const double *locs, const double *X, const double *z,
                                    const double *smooth_limits, int device, int nnz, const int *colindices,
                                    const int *rowpointers, const double *taper_entries);

/* Kriging core of the sparse branch of cocoPredict (R/predict.R:216-283) on a taper handle: replaces
 * cov_rns_taper / cov_rns_taper_pred times their tapers, inv_cov <- spam::solve(taper_two, t(pred_taper)) (:244),
 * crossprod(resid, inv_cov) (:252) and rowSums(pred_taper * t(inv_cov)) (:267).  pred_taper's slots go in as they
 * are (m rows, columns = observations, 1-based); theta / mean / z_col as for cocons_predict_dense.
 * stochastic[m], quadform[m] come back; the systematic part and the variance lines (:247-249, :264-272) stay in R. */
int cocons_predict_taper(cocons_fit *fit, const double *theta, const double *mean, int z_col, int m,
                         const double *locs_pred, const double *X_pred, int nnz_pred, const int *colindices_pred,
                         const int *rowpointers_pred, const double *taper_entries_pred,
                         double *stochastic, double *quadform);

/* ---- fit handle: everything that is constant over an optimisation -------------
 * Created once per cocoOptim / getHessian call from the arguments the reference
 * passes unchanged to every GetNeg2loglikelihood* evaluation
 * (R/optim.R:237-259): locs, x_covariates (= mod_DM), z (n x r), optional
 * x_betas (n x q, Profile only), smooth.limits.  After creation only O(p) bytes
 * cross PCIe per evaluation.  `device` < 0 picks (pid-stable) device 0.
 * `n_extra_rows` reserves room for cocons_predict_dense (0 if unused).          */
cocons_fit *cocons_fit_create(int n, int p, int r, int q, const double *locs,
                              const double *X, const double *z, const double *x_betas,
                              const double *smooth_limits, int device);
void cocons_fit_destroy(cocons_fit *fit);

/* Fused -2 log-likelihood core: replaces the chain
 *   cov_rns -> base::chol (dpotrf) -> sum(log(diag)) -> forwardsolve (dtrsm) -> crossprod
 * of GetNeg2loglikelihood, R/neg2loglikelihood.R:195-218.
 * `mean` (length p) is theta_list$mean.  Outputs:
 *   *sum_logliks = sum_k [ n log(2 pi) + 2 logdet + || R^-T (z_k - X mean) ||^2 ]
 *   parts[0] = sum(log(diag(chol))), parts[1..r] = the r quadratic forms (may be NULL)
 * The penalty (.cocons.getPen, R/checkFunctions.R:474-492) is O(p) host work and
 * stays with the caller.                                                         */
int cocons_neg2loglik_dense(cocons_fit *fit, const double *theta, const double *mean,
                            double *sum_logliks, double *parts);

/* Batch of nb independent evaluations of the same fit (the 1 + 2P points of one
 * finite-difference gradient, R/optim.R:237-259 with R/profile.R:11-12; getHessian's
 * 3 P (P+1)/2 points, R/getFunctions.R:979-1016).  thetas: nb x (6 p) row-major tables,
 * means: nb x p, values[nb] = sum_logliks of each, status[nb] = 0 or the failing minor k > 0.
 * Evaluations are pipelined over a few internal slots (COCONS_BATCH_SLOTS, default 2, each on the engine schedule) so the
 * latency-bound panel chain of one overlaps the updates and the assembly of the others.    */
int cocons_neg2loglik_batch(cocons_fit *fit, int nb, const double *thetas, const double *means,
                            double *values, int *status);

/* Profile / REML cores: replace R/neg2loglikelihood.R:132-160 and :254-287.
 * Both avoid chol2inv and the n x n P_mat through
 *   z' P z = ||L^-1 z||^2 - (Y'y)' (Y'Y)^-1 (Y'y),  Y = L^-1 Xb, y = L^-1 z.
 * profile: Xb = x_betas given at fit creation (q columns);
 * reml:    Xb = x_covariates (as the reference does, :273-276); `rank` = qr(X)$rank,
 *          computed by the caller.  parts[0] = sum(log(diag(chol))),
 * parts[1] = sum(log(diag(chol(W)))), parts[2 .. 2+r) = quadratic forms,
 * parts[2+r .. 2+r+nxb) = the GLS coefficients W^-1 Xb' Sigma^-1 rowSums(z)/r that cocoOptim
 * recovers after a pml/reml fit (R/optim.R:329-341); nxb = q (profile) or p (reml).
 * `parts` may be NULL, else must hold 2 + r + nxb doubles.                        */
int cocons_neg2loglik_profile(cocons_fit *fit, const double *theta,
                              double *sum_logliks, double *parts);
int cocons_neg2loglik_reml(cocons_fit *fit, const double *theta, int rank,
                           double *sum_logliks, double *parts);

/* Dense kriging core: replaces R/predict.R:136-183
 *   observed_cov <- cov_rns(...); cov_pred <- cov_rns_pred(...);
 *   inv_cov <- solve(observed_cov, t(cov_pred)); crossprod(resid, inv_cov);
 *   rowSums(cov_pred * t(inv_cov))
 * with one bordered Cholesky (Sigma is SPD) instead of LU.  Uses the first
 * realization column `z_col` of the fit's z.  Outputs (length m):
 *   stochastic[i] = c_i' Sigma^-1 (z - X mean),  quadform[i] = c_i' Sigma^-1 c_i     */
int cocons_predict_dense(cocons_fit *fit, const double *theta, const double *mean,
                         int z_col, int m, const double *locs_pred, const double *X_pred,
                         double *stochastic, double *quadform);

/* Rows of the covariance (cor = 0) or correlation (cor != 0, stats::cov2cor) matrix of the fit's
 * locations without forming the n x n matrix: what getCovMatrix's consumers read of it --
 * plot(type = "correlations") uses tmp_cov[ww, ] only (R/methods.R:161-165, :210-214; cov_rns at
 * R/getFunctions.R:44-52).  idx: nidx 0-based row indices in the caller's (original) observation order;
 * classic != 0 selects cov_rns_classic; out: nidx rows of n doubles (row b at out + b * n).
 * O(n p) bytes go down and nidx * n doubles come back instead of the 8 n^2-byte matrix.               */
int cocons_cov_rows(cocons_fit *fit, const double *theta, int classic, int nidx, const int *idx, int cor,
                    double *out);

/* Marginal simulation core: replaces R/sim.R:147-172
 *   covmat <- cov_rns(...) | cov_rns_classic(...); cholS <- chol(covmat);
 *   t(sweep(t(iiderrors) %*% cholS, 2, x %*% mean, "+"))
 * iiderrors and out are n x nsim column-major (the caller draws the N(0,1) numbers, as the
 * reference does with rnorm); classic != 0 selects cov_rns_classic.                        */
int cocons_sim_dense(cocons_fit *fit, const double *theta, const double *mean, int classic,
                     int nsim, const double *iiderrors, double *out);

/* Conditional simulation core: replaces R/sim.R:84-127 (cov_rns, cov_rns_pred, cov_rns on the new
 * locations, solve, chol of the Schur complement, cocoPredict(type = "mean")) by ONE Cholesky of the
 * joint covariance of (observed, new) locations.  locs_pred = newlocs (cross-covariance),
 * locs_unobs = the coordinates the reference hands to cov_rns for covmat_unobs (the first two
 * columns of newdataset, :96-99); iiderrors / out are m x nsim column-major.               */
int cocons_sim_cond_dense(cocons_fit *fit, const double *theta, const double *mean, int z_col,
                          int m, const double *locs_pred, const double *X_pred,
                          const double *locs_unobs, int nsim, const double *iiderrors, double *out);

/* Dense Cholesky of a caller-supplied SPD matrix (host, n x n column-major, lower
 * triangle read) with nrhs right-hand sides: replaces base::chol + forwardsolve
 * (R/neg2loglikelihood.R:200,214) for callers that already hold Sigma.
 * L (optional, n x n) receives the lower factor (= t(chol(Sigma))), Y (optional,
 * n x nrhs) receives L^-1 rhs, logdet_half = sum(log(diag(L))).                   */
int cocons_chol_solve(int n, const double *A, int nrhs, const double *rhs,
                      double *L, double *Y, double *logdet_half);

/* ---- measurement hooks (bench.py / rocprof): device-resident, no host copies ---
 * Runs `reps` complete evaluations of cocons_neg2loglik_dense back to back on the
 * fit's stream with HIP events around each stage.  ms[0]=assembly, ms[1]=Cholesky
 * (+solve, fused), ms[2]=reductions, ms[3]=whole evaluation, ms[4]=average duration
 * of one trailing-update (MFMA) launch, ms[5]=number of such launches per
 * evaluation, ms[6]=sum of trailing-update launch durations per evaluation,
 * ms[7]=algorithmic flops of those launches (K m (m+1) + 2 K r m each; m = trailing order),
 * ms[8]=duration of the ONE persistent launch that runs the head of the factorisation under the
 * dependency-driven schedule (it is also the first of the launches counted in ms[5..7]; 0 when the
 * classic schedule ran), ms[9]=the update flops inside it.  `ms` must hold 10 doubles.           */
int cocons_fit_profile(cocons_fit *fit, const double *theta, const double *mean,
                       int reps, double *ms);

/* State of the resident diagonal-block engine of a dense handle: out[0] = 1 if the last
 * completed operation ran on the engine schedule (0: plain schedule -- small n, a band-limited taper fit, a batch
 * slot, COCONS_ENGINE=0, or the back-off after a time-out), out[1] = hand-off time-outs in the life of the handle
 * (each was answered by ONE repeat of that operation on the plain schedule, then 2, 4 ... 64 further operations
 * stay on it before the engine is tried again), out[2] = the abort code of the last time-out (0 = none).  No
 * reference counterpart: the observability of a mechanism the reference does not have.                       */
int cocons_fit_engine_state(cocons_fit *fit, int *out3);

/* ---- natively sharded evaluation across the GPUs of one node ----------------------
 * Sigma is ROW-BLOCK partitioned (block b = rows 256 b .. 256 b + 255; blocks dealt in groups, see
 * cocons_shard_block_owner below): a rank assembles, solves and updates ITS rows of every column.  Per 256-column
 * block the owner of the diagonal block factors it and the library broadcasts it (0.56 MB, RCCL over xGMI, issued in
 * front of the bulk exchange; COCONS_SHARD_COMM2=1: on a stream -- and communicator -- of its own), every rank solves its rows of
 * the panel, the owner of the NEXT diagonal block updates and factors it from its own rows at once, the solved rows
 * are all-gathered packed by owner (second communication stream), every rank updates its rows of the trailing
 * matrix; the 1 + r^2 partial sums (+ the failing minor) are all-reduced at the end.  No
 * data-path call leaves the library: once a fit has collectives, cocons_neg2loglik_dense on it IS the sharded
 * evaluation and returns the same value on every rank.
 *
 * (a) one process per GPU (torch.distributed.run, MPI, optimParallel workers ...): rank 0 calls
 *     cocons_comm_unique_id, the 128 bytes reach the other ranks by whatever channel the host has, and
 *     every rank calls cocons_fit_comm_init on its own fit (ncclCommInitRank).
 * (b) one process, several GPUs (what a single R session needs, R/optim.R:117-121's single-caller model):
 *     cocons_multi_create takes the device list (ncclCommInitAll), cocons_multi_neg2loglik_dense drives
 *     all of them from the calling thread.
 * (c) tests / other transports: cocons_fit_set_collectives installs caller-provided broadcast and
 *     all-reduce functions (e.g. gloo when several ranks share one GPU, which RCCL refuses).           */
#define COCONS_UNIQUE_ID_BYTES 128
int cocons_comm_unique_id(void *id_out /* COCONS_UNIQUE_ID_BYTES */);
int cocons_fit_comm_init(cocons_fit *fit, int nranks, int rank, const void *id /* COCONS_UNIQUE_ID_BYTES */);
/* broadcast `bytes` at device pointer dev_ptr from rank `root` to every rank; `stream` (hipStream_t) is the
 * library's communication stream for these broadcasts (ordered behind whatever produced / last read the buffer):
 * the function may enqueue on it or block.  Return 0 on success.        */
typedef int (*cocons_bcast_fn)(void *user, void *dev_ptr, long long bytes, int root, void *stream);
/* in-place all-reduce of `count` HOST doubles, op 0 = sum, 1 = min; blocking.                            */
typedef int (*cocons_allreduce_fn)(void *user, double *host_inout, int count, int op);
int cocons_fit_set_collectives(cocons_fit *fit, int rank, int world, cocons_bcast_fn bcast,
                               cocons_allreduce_fn allreduce, void *user);
/* ... and the all-gather of the sharded evaluation: dev_buf holds `world` slots of bytes_per_rank bytes, slot r is
 * rank r's contribution (already in place on rank r); on return every slot is filled on every rank.  `stream`: the
 * stream the library has ordered the buffer's producer on (synchronise it, or enqueue behind it).  0 = ok.        */
typedef int (*cocons_allgather_fn)(void *user, void *dev_buf, long long bytes_per_rank, void *stream);
int cocons_fit_set_allgather(cocons_fit *fit, cocons_allgather_fn allgather);
/* number of ranks the fit is sharded over (1 = not sharded) */
int cocons_fit_world(cocons_fit *fit);

typedef struct cocons_multi cocons_multi;   /* one fit per device + their communicators */
cocons_multi *cocons_multi_create(int n, int p, int r, const double *locs, const double *X, const double *z,
                                  const double *smooth_limits, int ndev, const int *devices);
void cocons_multi_destroy(cocons_multi *m);
/* same outputs as cocons_neg2loglik_dense */
int cocons_multi_neg2loglik_dense(cocons_multi *m, const double *theta, const double *mean,
                                  double *sum_logliks, double *parts);

/* cocoPredict's dense core (cocons_predict_dense) with the m prediction locations split over the devices of
 * the handle: no exchange, slices concatenated on the host (BASELINE config C5).                         */
int cocons_multi_predict_dense(cocons_multi *m, const double *theta, const double *mean, int z_col,
                               int m_pred, const double *locs_pred, const double *X_pred,
                               double *stochastic, double *quadform);
/* Replica mode inside ONE process: the nb independent parameter points of a finite-difference
 * gradient (R/optim.R:256-259, 1 + 2P points) or of getHessian (R/getFunctions.R:979-1016) dealt over the devices
 * of the handle (point i on device i mod ndev), every device running its share through cocons_neg2loglik_batch on its
 * own fit; no collective, so the handle may list a device more than once.  Arguments as cocons_neg2loglik_batch. */
int cocons_multi_neg2loglik_batch(cocons_multi *m, int nb, const double *thetas, const double *means,
                                  double *values, int *status);
/* What the communicators really span: *ndev = devices of the handle, *rccl_count = ncclCommCount of its first
 * communicator (0: no communicator -- a device is listed twice).                                                  */
int cocons_multi_comm_ranks(cocons_multi *m, int *ndev, int *rccl_count);
/* The same for a fit with collectives of its own (cocons_fit_comm_init / cocons_fit_set_collectives): ncclCommCount,
 * ncclCommUserRank, ncclCommCuDevice (or the caller-provided world / rank and the fit's device).  bench.py prints
 * them for every rank.                                                                                            */
int cocons_fit_comm_info(cocons_fit *fit, int *count, int *user_rank, int *device);

/* ---- how the sharded evaluation deals the matrix (no GPU call) ----------------------------------------
 * Sigma is ROW-BLOCK partitioned: block b = rows 256 b ... 256 b + 255 (the reference's chol walks the upper
 * triangle row block by row block, R/neg2loglikelihood.R:200); blocks are dealt in groups of G consecutive
 * blocks, owner(b) = (b / G) mod world (COCONS_SHARD_GROUP, default 4).  Per block: its owner factors the
 * 256 x 256 diagonal block and broadcasts it (0.5 MB), every rank solves ITS rows of the panel, the solved
 * rows are all-gathered, every rank updates its rows.                              */
int cocons_shard_block_owner(int b, int world);
int cocons_shard_num_blocks(cocons_fit *fit);
/* 1 if the handle holds exactly these data (bitwise comparison with the host copies it keeps), else 0.  Host work only.
 * The R glue's handle cache (glue/cocons_hip_glue.c, _cocons_hip_fit_cached) calls it when its O(1) address check
 * misses, so that GetNeg2loglikelihood(theta, par.pos, locs, x_covariates, smooth.limits, z, n, lambda) -- the
 * reference's signature, R/neg2loglikelihood.R:183-191, no handle argument -- finds its handle without hashing the data. */
int cocons_fit_same_data(cocons_fit *fit, int n, int p, int r, int q, const double *locs, const double *X,
                         const double *z, const double *x_betas, const double *smooth_limits);
/* HIP stream the fit launches on (hipStream_t as void*), so the caller can order
 * collectives against it. */
void *cocons_fit_stream(cocons_fit *fit);
/* launch on a caller-owned stream instead (hipStream_t as void*; NULL = default stream) */
int cocons_fit_set_stream(cocons_fit *fit, void *stream);
int cocons_fit_sync(cocons_fit *fit);

#ifdef __cplusplus
}
#endif
#endif /* COCONS_HIP_H */
