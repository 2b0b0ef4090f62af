# glue/R/cocons_hip.R -- R side of the HIP drop-in: replaces the BODIES of the reference's objective
# closures (R/neg2loglikelihood.R:127-291); names, arguments and return values are unchanged, so
# cocoOptim / getHessian / cocoPredict / cocoSim call them as before.  The thin wrappers of
# R/RcppExports.R (cov_rns, cov_rns_classic, cov_rns_pred, cov_rns_taper*, sumsmoothlone) stay as they
# are: only the native symbols behind them change (glue/cocons_hip_glue.c).
#
# The device-resident data of a fit live in an explicit handle (external pointer).  Callers that hold
# one pass it as `fit = `; otherwise -- cocoOptim calls the closures with the reference's signatures, which have no
# handle -- the native side keeps up to eight handles per process and finds the right one in O(1) by the addresses and
# dimensions of (locs, x_covariates, z, x_betas) and smooth.limits.  An address is a sound key because the cached objects
# are PRESERVED (no other object can get the address) and marked NOT MUTABLE (R code that modifies one must duplicate
# it first: new address, miss, data compared), and a hit is confirmed against the handle's copy of the data; only when
# the addresses miss are the data compared with every cached handle, and only then is a handle created
# (glue/cocons_hip_glue.c, _cocons_hip_fit_cached).  No hash of the data on any path, no package beyond base R.
# cocons_hip_forget() drops the cached handles and releases the objects they key on.

cocons_hip_fit <- function(locs, x_covariates, z, smooth.limits, x_betas = NULL, device = NULL) {
  if (is.null(device)) {
    ndev <- max(1L, .Call(`_cocons_hip_device_count`))
    device <- as.integer(Sys.getenv("COCONS_HIP_DEVICE", Sys.getpid() %% ndev))   # worker -> GPU map
  }
  z <- as.matrix(z)
  storage.mode(locs) <- storage.mode(x_covariates) <- storage.mode(z) <- "double"
  if (!is.null(x_betas)) { x_betas <- as.matrix(x_betas); storage.mode(x_betas) <- "double" }
  .Call(`_cocons_hip_fit_create`, locs, x_covariates, z, x_betas, as.double(smooth.limits), as.integer(device))
}

.cocons.hip.cached <- function(locs, x_covariates, z, smooth.limits, x_betas = NULL) {
  if (!is.matrix(z)) z <- as.matrix(z)                     # (no copy when the caller passes a matrix, as cocoOptim does)
  if (!is.double(locs) || !is.double(x_covariates) || !is.double(z) || (!is.null(x_betas) && !is.double(x_betas)) ||
      !is.double(smooth.limits)) {                           # integer inputs: converted once per call -- the rare path
    storage.mode(locs) <- storage.mode(x_covariates) <- storage.mode(z) <- "double"
    if (!is.null(x_betas)) storage.mode(x_betas) <- "double"
    smooth.limits <- as.double(smooth.limits)
  }
  .Call(`_cocons_hip_fit_cached`, locs, x_covariates, z, x_betas, smooth.limits, -1L)   # -1: worker -> GPU map, natively
}

cocons_hip_forget <- function() invisible(.Call(`_cocons_hip_cache_clear`))

.cocons.hip.result <- function(res, safe) {       # the reference's tryCatch contract, :200-206
  if (res[[1]] > 0L) {
    if (safe) return(NULL) else stop("Cholesky error")
  }
  res[[2]]
}

GetNeg2loglikelihood <- function(theta, par.pos, locs, x_covariates, smooth.limits, z, n, lambda,
                                 safe = TRUE, fit = NULL) {
  theta_list <- cocons::getModelLists(theta = theta, par.pos = par.pos, type = "diff")     # :193
  if (is.null(fit)) fit <- .cocons.hip.cached(locs, x_covariates, z, smooth.limits)
  val <- .cocons.hip.result(.Call(`_cocons_hip_neg2loglik`, fit, theta_list[-1], theta_list$mean), safe)
  if (is.null(val)) return(1e+06)
  val + .cocons.getPen(n * dim(as.matrix(z))[2], lambda, theta_list, smooth.limits)          # :220
}

# the 1 + 2P points of one finite-difference gradient, or getHessian's 3P(P+1)/2 (R/getFunctions.R:979-1016)
GetNeg2loglikelihoodBatch <- function(thetas, par.pos, locs, x_covariates, smooth.limits, z, n, lambda,
                                      safe = TRUE, fit = NULL) {
  tl <- lapply(thetas, function(t) cocons::getModelLists(theta = t, par.pos = par.pos, type = "diff"))
  if (is.null(fit)) fit <- .cocons.hip.cached(locs, x_covariates, z, smooth.limits)
  res <- .Call(`_cocons_hip_neg2loglik_batch`, fit, lapply(tl, function(x) x[-1]), lapply(tl, function(x) x$mean))
  out <- res[[2]]
  for (i in seq_along(tl)) {
    if (res[[1]][i] > 0L) { if (safe) out[i] <- 1e+06 else stop("Cholesky error") }
    else out[i] <- out[i] + .cocons.getPen(n * dim(as.matrix(z))[2], lambda, tl[[i]], smooth.limits)
  }
  out
}

GetNeg2loglikelihoodProfile <- function(theta, par.pos, locs, x_covariates, smooth.limits, z, n, x_betas,
                                        lambda, safe = TRUE, fit = NULL) {
  theta_list <- cocons::getModelLists(theta = theta, par.pos = par.pos, type = "diff")
  if (is.null(fit)) fit <- .cocons.hip.cached(locs, x_covariates, z, smooth.limits, x_betas)
  v <- .cocons.hip.result(.Call(`_cocons_hip_neg2loglik_profile`, fit, theta_list[-1]), safe)
  if (is.null(v)) return(1e+06)
  v[1] + .cocons.getPen(n * dim(as.matrix(z))[2], lambda, theta_list, smooth.limits)
}

GetNeg2loglikelihoodREML <- function(theta, par.pos, locs, x_covariates, x_betas, smooth.limits, z, n,
                                     lambda, safe = TRUE, fit = NULL) {
  theta_list <- cocons::getModelLists(theta = theta, par.pos = par.pos, type = "diff")
  if (is.null(fit)) fit <- .cocons.hip.cached(locs, x_covariates, z, smooth.limits)
  v <- .cocons.hip.result(.Call(`_cocons_hip_neg2loglik_reml`, fit, theta_list[-1],
                                as.integer(qr(x_covariates)$rank)), safe)                  # :270
  if (is.null(v)) return(1e+06)
  v[1] + .cocons.getPen(n * dim(as.matrix(z))[2], lambda, theta_list, smooth.limits)
}

# GLS coefficients after a pml / reml fit (R/optim.R:329-341) without a second chol:
# v = c(sum_logliks, logdet, logdet_W, quad_1..r, beta_1..) as returned by the cores above
.cocons.hip.betas <- function(v, r) v[-seq_len(3 + r)]

# dense kriging core for cocoPredict (R/predict.R:136-183): the four lines cov_rns / cov_rns_pred /
# solve / rowSums become
#   kr <- .cocons.hip.predict(fit, theta_list, newlocs, X_pred_std)
#   stochastic <- kr[, 1];  quadform <- kr[, 2]      # c_i' Sigma^-1 resid,  c_i' Sigma^-1 c_i
.cocons.hip.predict <- function(fit, theta_list, newlocs, X_pred, z_col = 1L) {
  res <- .Call(`_cocons_hip_predict`, fit, theta_list[-1], theta_list$mean, as.integer(z_col), newlocs, X_pred)
  if (res[[1]] > 0L) stop("Cholesky error")
  res[[2]]
}

# rows of cov2cor(cov_rns(...)) for plot(type = "correlations") (R/methods.R:161-165): tmp_cov[ww, ]
.cocons.hip.cor.rows <- function(fit, theta_list, index, classic = FALSE)
  .Call(`_cocons_hip_cov_rows`, fit, theta_list[-1], classic, as.integer(index), TRUE)

# one R process, several GPUs: Sigma row blocks sharded over `devices`, RCCL inside the library
cocons_hip_multi <- function(locs, x_covariates, z, smooth.limits, devices)
  .Call(`_cocons_hip_multi_create`, locs, x_covariates, as.matrix(z), as.double(smooth.limits), as.integer(devices))

# replica mode inside ONE R process: a list of parameter points (the 2p+1 points of a finite-difference gradient,
# R/optim.R:256-259, or getHessian's grid, R/getFunctions.R:979-1016) dealt over the GPUs of a multi handle
cocons_hip_multi_neg2loglik_batch <- function(m, theta_lists) {
  res <- .Call(`_cocons_hip_multi_neg2loglik_batch`, m, lapply(theta_lists, function(th) th[-1]),
               lapply(theta_lists, function(th) th$mean))
  ifelse(res[[1]] > 0L, 1e+06, res[[2]])
}

# c(active, time-outs, last abort code) of the resident diagonal-block engine of a fit handle (diagnostic)
cocons_hip_engine_state <- function(fit) .Call(`_cocons_hip_engine_state`, fit)

# cocoPredict's dense core with the prediction locations split over the GPUs of a multi handle (config C5)
cocons_hip_multi_predict <- function(m, theta_list, newlocs, X_pred, z_col = 1L) {
  res <- .Call(`_cocons_hip_multi_predict`, m, theta_list[-1], theta_list$mean, as.integer(z_col), newlocs, X_pred)
  if (res[[1]] > 0L) stop("Cholesky error")
  res[[2]]
}

# ---- type = "sparse": the taper objective through the dense factorisation on the device --------------------
# handle of one optimisation; ref_taper is the spam object coco() builds (R/cocons.R), n^2 doubles must fit the GPU
cocons_hip_taper_fit <- function(locs, x_covariates, z, smooth.limits, ref_taper, device = -1L)
  .Call(`_cocons_hip_fit_create_taper`, locs, x_covariates, as.matrix(z), as.double(smooth.limits), as.integer(device),
        ref_taper@colindices, ref_taper@rowpointers, as.double(ref_taper@entries))

# body of GetNeg2loglikelihoodTaper (R/neg2loglikelihood.R:20-53); cholS is not used
.cocons.hip.GetNeg2loglikelihoodTaper <- function(theta, par.pos, fit, smooth.limits, z, n, lambda, safe = TRUE) {
  theta_list <- getModelLists(theta = theta, par.pos = par.pos, type = "diff")
  val <- .cocons.hip.result(.Call(`_cocons_hip_neg2loglik`, fit, theta_list[-1], theta_list$mean), safe)
  if (is.null(val)) return(1e+06)
  val + .cocons.getPen(n * dim(z)[2], lambda, theta_list, smooth.limits)
}

# body of GetNeg2loglikelihoodTaperProfile (R/neg2loglikelihood.R:73-108)
.cocons.hip.GetNeg2loglikelihoodTaperProfile <- function(theta, par.pos, fit, smooth.limits, z, n, lambda, safe = TRUE) {
  theta_list <- getModelLists(theta = theta, par.pos = par.pos, type = "diff")
  theta_list$std.dev[1] <- 0
  v <- .cocons.hip.result(.Call(`_cocons_hip_neg2loglik_parts`, fit, theta_list[-1], theta_list$mean), safe)
  if (is.null(v)) return(1e+06)
  r <- dim(z)[2]; logdet <- v[2]; sum_in <- sum(v[-(1:2)])
  r * n * log(2 * pi) + r * n + r * 2 * logdet + r * n * log(sum_in / (r * n)) +
    .cocons.getPen(n * r, lambda, theta_list, smooth.limits)
}

# sparse branch of cocoPredict (R/predict.R:216-283): the lines from cov_rns_taper to rowSums(pred_taper * t(inv_cov)) become
#   kr <- .cocons.hip.predict.taper(fit, theta_list, newlocs, X_pred_std, pred_taper)   # pred_taper: the spam object of :229-231
#   stochastic_part <- kr[, 1];  uncertainty_some <- uncertainty_some - kr[, 2]
.cocons.hip.predict.taper <- function(fit, theta_list, newlocs, X_pred, pred_taper, z_col = 1L) {
  res <- .Call(`_cocons_hip_predict_taper`, fit, theta_list[-1], theta_list$mean, as.integer(z_col), newlocs, X_pred,
               pred_taper@colindices, pred_taper@rowpointers, as.double(pred_taper@entries))
  if (res[[1]] > 0L) stop("Cholesky error")
  res[[2]]
}
