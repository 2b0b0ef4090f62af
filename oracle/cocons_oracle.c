/*
 * oracle/cocons_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, single thread) of the reference's dense
 * covariance assembly, written from a reading of
 *     /root/reference/src/cocons_full.cpp   and   src/cocons_types.h
 * operation-for-operation (same fma chains, same compensated 2x2 products,
 * same 1/exp(-t) link forms, same branch thresholds eps and 706.0).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path (cocons_amd/) never links or imports it.
 *
 * PARITY UNPINNED w.r.t. the reference binary: the reference cannot be built
 * here (needs Rcpp.h, R.h and Boost headers, none installed) and its own test
 * script holds no numeric golden vector for cov_rns* or -2loglik
 * (tests/coco_test.R asserts shapes / eigenvalues>0 / non-NA only).  The oracle
 * is therefore pinned by (a) closed-form Matern identities, (b) 40-digit mpmath
 * evaluations committed under tests/golden/, (c) cross-branch identities
 * between cov_rns / cov_rns_classic / cov_rns_pred.
 *
 * Third-party arithmetic on the path that is NOT under /root/reference:
 *   - boost::math::cyl_bessel_k  (BH headers, version unpinned, DESCRIPTION:25;
 *     call sites src/cocons_full.cpp:294,450,573).  Restated below from the
 *     published algorithm Boost's bessel_ik.hpp documents: Temme's series for
 *     x <= 2, Steed's continued fraction CF2 for x > 2, forward recurrence in
 *     the order from mu = nu - round(nu) to nu.
 *   - LAPACK dpotrf/dtrtrs behind base::chol / forwardsolve
 *     (R/neg2loglikelihood.R:200,214): taken from scipy's LAPACK in
 *     oracle/oracle.py; a long-double unblocked Cholesky is provided here as a
 *     higher-precision truth for small n.
 *
 * Matrix arguments are column-major (R layout).  theta is a 6 x p row-major
 * table in the reference's dictionary order minus "mean":
 *   row 0 std.dev, 1 scale, 2 aniso, 3 tilt, 4 smooth, 5 nugget.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

enum { TH_SD = 0, TH_SCALE = 1, TH_ANISO = 2, TH_TILT = 3, TH_SMOOTH = 4, TH_NUGGET = 5 };

/* ---- src/cocons_types.h:12-18  Pexpfma_new ------------------------------ */
static double pexpfma(const double *b, const double *X, int n, int p, int row)
{
    double t = 0.0;
    for (int i = 0; i < p; ++i)
        t = fma(X[row + (size_t)i * n], b[i], t);
    return 1 / exp(-1 * t);
}

/* ---- src/cocons_types.h:20-28  Pexpfma_new_smoothness ------------------- */
static double pexpfma_smooth(const double *b, const double *X, int n, int p, int row,
                             double min_v, double max_v)
{
    double t = 0.0;
    for (int i = 0; i < p; ++i)
        t = fma(X[row + (size_t)i * n], b[i], t);
    return (max_v - min_v) / (1 + exp(-1 * t)) + min_v;
}

/* ---- src/cocons_types.h:40-47  newinvlogitfma --------------------------- */
static double invlogit_pi(const double *b, const double *X, int n, int p, int row)
{
    double t = 0.0;
    for (int i = 0; i < p; ++i)
        t = fma(X[row + (size_t)i * n], b[i], t);
    return M_PI / (1 + exp(-1 * t));
}

/* ---- src/cocons_types.h:49-54  kahan ------------------------------------ */
static double kahan(double a, double b, double c, double d)
{
    double cd = c * d;
    double err = fma(c, d, -cd);
    double res = fma(a, b, -cd);
    return res - err;
}

/* ---- src/cocons_types.h:56-70 ------------------------------------------- */
static int all_zero_from_second(const double *x, int p)
{
    for (int i = 1; i < p; ++i)
        if (x[i] != 0) return 0;
    return 1;
}

static int map_smooth_value(double v)
{
    if (fabs(v - 0.5) < 1e-6) return 1;
    if (fabs(v - 1.5) < 1e-6) return 2;
    if (fabs(v - 2.5) < 1e-6) return 3;
    return 0;
}

/* ---- Bessel K_nu(x): Temme series / Steed CF2 / forward recurrence ------- */
#include "rgamma_coeffs.h"
static const double rg_even[RG_NTERMS] = RG_EVEN_INIT;
static const double rg_odd[RG_NTERMS] = RG_ODD_INIT;

double oracle_besselk(double nu, double x)
{
    if (!(x > 0)) return (x == 0) ? INFINITY : NAN;
    if (nu < 0) nu = -nu;
    int n = (int)floor(nu + 0.5);
    double v = nu - n;                 /* |v| <= 1/2 */
    double kv, kv1;
    const double tol = DBL_EPSILON;
    if (x <= 2.0) {
        /* Temme's series for K_v and K_{v+1}.  gam1, gam2 from the Taylor series
         * of 1/Gamma(1+z): well conditioned as v -> 0 (orders near an integer). */
        double v2 = v * v, gam1 = 0, gam2 = 0;
        for (int j = RG_NTERMS - 1; j >= 0; --j) {
            gam1 = gam1 * v2 + rg_odd[j];
            gam2 = gam2 * v2 + rg_even[j];
        }
        double gampl = gam2 - v * gam1;       /* 1/Gamma(1+v) */
        double gammi = gam2 + v * gam1;       /* 1/Gamma(1-v) */
        double x2 = 0.5 * x, pimu = M_PI * v;
        double fact = fabs(pimu) < tol ? 1.0 : pimu / sin(pimu);
        double d = -log(x2), e = v * d;
        double fact2 = fabs(e) < tol ? 1.0 : sinh(e) / e;
        double ff = fact * (gam1 * cosh(e) + gam2 * fact2 * d);
        double sum = ff;
        e = exp(e);
        double pp = 0.5 * e / gampl, q = 0.5 / (e * gammi), c = 1.0;
        d = x2 * x2;
        double sum1 = pp;
        for (int i = 1; i < 100000; ++i) {
            ff = (i * ff + pp + q) / (i * (double)i - v2);
            c *= d / i;
            pp /= i - v;
            q /= i + v;
            double del = c * ff;
            sum += del;
            sum1 += c * (pp - i * ff);
            if (fabs(del) < fabs(sum) * tol) break;
        }
        kv = sum;
        kv1 = sum1 * (2.0 / x);
    } else {
        /* Steed's algorithm for the continued fraction CF2 */
        double a = v * v - 0.25;
        double b = 2 * (x + 1), D = 1 / b, f = D, delta = D;
        double prev = 0, cur = 1, C = -a, Q = C, S = 1 + Q * delta;
        for (int k = 2; k < 100000; ++k) {
            a -= 2 * (k - 1);
            b += 2;
            D = 1 / (b + a * D);
            delta *= b * D - 1;
            f += delta;
            double qn = (prev - (b - 2) * cur) / a;
            prev = cur;
            cur = qn;
            C *= -a / k;
            Q += C * qn;
            S += Q * delta;
            if (fabs(Q * delta) < fabs(S) * tol) break;
        }
        kv = sqrt(M_PI / (2 * x)) * exp(-x) / S;
        kv1 = kv * (0.5 + v + x + (v * v - 0.25) * f) / x;
    }
    /* forward recurrence K_{v+k+1} = K_{v+k-1} + 2 (v+k)/x K_{v+k} */
    double prev = kv, cur = kv1;
    for (int k = 1; k <= n; ++k) {
        double next = 2 * (v + k) * cur / x + prev;
        prev = cur;
        cur = next;
    }
    return prev;
}

/* per-location vectors shared by the three entry points ---------------------- */
typedef struct {
    double *tilt, *rd, *an, *dets, *sigma, *nugget, *smooth;
} locvec_t;

static int locvec_alloc(locvec_t *v, int n)
{
    double *blk = (double *)calloc((size_t)7 * (n > 0 ? n : 1), sizeof(double));
    if (!blk) return -1;
    v->tilt = blk; v->rd = blk + n; v->an = blk + 2 * (size_t)n; v->dets = blk + 3 * (size_t)n;
    v->sigma = blk + 4 * (size_t)n; v->nugget = blk + 5 * (size_t)n; v->smooth = blk + 6 * (size_t)n;
    return 0;
}

static void locvec_free(locvec_t *v) { free(v->tilt); }

/* helper vectors derived from theta exactly as the reference forms them:
 *   scale_je = scale with [0]=0 (cocons_full.cpp:49,64), 2*scale_je,
 *   sqrt_vector = 2*scale_je + aniso (:66), 0.5*std.dev (:104) */
typedef struct { double *two_scale_je, *sqrt_vector, *half_sd; } thvec_t;

static int thvec_make(thvec_t *t, const double *theta, int p)
{
    double *blk = (double *)malloc((size_t)3 * p * sizeof(double));
    if (!blk) return -1;
    t->two_scale_je = blk; t->sqrt_vector = blk + p; t->half_sd = blk + 2 * p;
    for (int i = 0; i < p; ++i) {
        double sje = (i == 0) ? 0.0 : theta[TH_SCALE * p + i];
        t->two_scale_je[i] = 2 * sje;
        t->sqrt_vector[i] = 2 * sje + theta[TH_ANISO * p + i];
        t->half_sd[i] = 0.5 * theta[TH_SD * p + i];
    }
    return 0;
}

static void locvec_fill_common(locvec_t *v, const thvec_t *t, const double *theta,
                               const double *X, int n, int p)
{
    /* cocons_full.cpp:98-107 (and :374-383, :517-527) */
    for (int w = 0; w < n; ++w) {
        v->tilt[w] = invlogit_pi(theta + TH_TILT * p, X, n, p, w);
        v->rd[w] = pexpfma(t->two_scale_je, X, n, p, w);
        v->an[w] = pexpfma(theta + TH_ANISO * p, X, n, p, w);
        v->dets[w] = pexpfma(t->sqrt_vector, X, n, p, w);
        v->sigma[w] = pexpfma(t->half_sd, X, n, p, w);
        v->nugget[w] = pexpfma(theta + TH_NUGGET * p, X, n, p, w);
    }
}

/* geometry shared by every branch: returns smooth_s_Q_ij and det_ij.
 * (i = first/"ii" location, j = second/"jj"), cocons_full.cpp:122-141 etc. */
static inline double pair_geometry(const locvec_t *vi, int ii, const locvec_t *vj, int jj,
                                   double dx, double dy, double smtns, double global_range,
                                   double *det_out)
{
    double s11 = (vi->rd[ii] + vj->rd[jj]) * 0.5;
    double s22 = kahan(vi->rd[ii], vi->an[ii] * vi->an[ii],
                       -vj->rd[jj], vj->an[jj] * vj->an[jj]) * 0.5;
    double s12 = kahan(vi->rd[ii] * vi->an[ii], cos(vi->tilt[ii]),
                       -1 * vj->rd[jj] * vj->an[jj], cos(vj->tilt[jj])) * 0.5;
    double det = kahan(s11, s22, s12, s12);
    double u = sqrt(8 * smtns / (global_range * det)) *
               sqrt(fma(kahan(s22, dx * dx, -s11, dy * dy), 1, -2 * s12 * dx * dy));
    *det_out = det;
    return u;
}

static inline double pair_amplitude(const locvec_t *vi, int ii, const locvec_t *vj, int jj)
{
    return sqrt(vi->dets[ii] * sin(vi->tilt[ii]) * vj->dets[jj] * sin(vj->tilt[jj]));
}

/* general-nu value, cocons_full.cpp:291-307 (and :447-463, :570-586) */
static inline double matern_general(double smtns, double u, double si, double sj,
                                    double amp, double det)
{
    if (u < 706.0)
        return pow(2.0, -(smtns - 1)) / tgamma(smtns) * pow(u, smtns) *
               oracle_besselk(smtns, u) * si * sj * amp / sqrt(det);
    return pow(2.0, -(smtns - 1)) / tgamma(smtns) * pow(u, smtns) *
           sqrt(M_PI / (2.0 * u)) * exp(-u) * si * sj * amp / sqrt(det);
}

/* ---- cov_rns: src/cocons_full.cpp:40-321 ----------------------------------- */
int oracle_cov_rns(int n, int p, const double *theta, const double *locs, const double *X,
                   const double *smooth_limits, double *out)
{
    const double epsilon = DBL_EPSILON;
    locvec_t v; thvec_t t;
    if (locvec_alloc(&v, n)) return -1;
    if (thvec_make(&t, theta, p)) { locvec_free(&v); return -1; }
    memset(out, 0, (size_t)n * n * sizeof(double));
    const double *sd = theta + TH_SD * p;
    double global_range = 1 / exp(-2 * theta[TH_SCALE * p + 0]);           /* :62 */

    int q_smooth_fix = all_zero_from_second(theta + TH_SMOOTH * p, p);       /* :77 */
    int smooth_switch = 0;
    double smooth_value = 0.0;
    if (q_smooth_fix && smooth_limits[0] == smooth_limits[1]) {              /* :85-88 */
        smooth_value = smooth_limits[0];
        smooth_switch = map_smooth_value(smooth_value);
        /* quirk: smooth vector stays zero-initialised (:83) */
    } else {
        for (int w = 0; w < n; ++w)                                          /* :92-94 */
            v.smooth[w] = sqrt(pexpfma_smooth(theta + TH_SMOOTH * p, X, n, p, w,
                                              smooth_limits[0], smooth_limits[1]));
    }
    locvec_fill_common(&v, &t, theta, X, n, p);

    for (int ii = 0; ii < n; ++ii)                                           /* :110-112 */
        out[ii + (size_t)ii * n] = pexpfma(sd, X, n, p, ii) + v.nugget[ii];

    for (int ii = 0; ii < n; ++ii) {
        for (int jj = ii + 1; jj < n; ++jj) {
            double dx = locs[ii] - locs[jj];
            double dy = locs[ii + (size_t)n] - locs[jj + (size_t)n];
            double smtns = (smooth_switch != 0) ? smooth_value : v.smooth[ii] * v.smooth[jj];
            double det;
            double u = pair_geometry(&v, ii, &v, jj, dx, dy, smtns, global_range, &det);
            double val;
            if (u <= epsilon) {
                val = pexpfma(sd, X, n, p, ii) + v.nugget[ii];
            } else {
                double amp = pair_amplitude(&v, ii, &v, jj);
                switch (smooth_switch) {
                case 1:   /* :150 */
                    val = exp(-u) * v.sigma[ii] * v.sigma[jj] * amp / sqrt(det);
                    break;
                case 2:   /* :196 */
                    val = (1 + u) * exp(-u) * v.sigma[ii] * v.sigma[jj] * amp / sqrt(det);
                    break;
                case 3:   /* :242 */
                    val = (1 + u + u * u / 3) * exp(-u) * v.sigma[ii] * v.sigma[jj] * amp / sqrt(det);
                    break;
                default:  /* :291-307 */
                    val = matern_general(smtns, u, v.sigma[ii], v.sigma[jj], amp, det);
                }
            }
            out[ii + (size_t)jj * n] = out[jj + (size_t)ii * n] = val;
        }
    }
    free(t.two_scale_je);
    locvec_free(&v);
    return 0;
}

/* ---- cov_rns_classic: src/cocons_full.cpp:480-594 --------------------------- */
int oracle_cov_rns_classic(int n, int p, const double *theta, const double *locs,
                           const double *X, double *out)
{
    const double epsilon = DBL_EPSILON;
    locvec_t v; thvec_t t;
    if (locvec_alloc(&v, n)) return -1;
    if (thvec_make(&t, theta, p)) { locvec_free(&v); return -1; }
    memset(out, 0, (size_t)n * n * sizeof(double));
    const double *sd = theta + TH_SD * p;
    double global_range = 1 / exp(-2 * theta[TH_SCALE * p + 0]);           /* :501 */
    locvec_fill_common(&v, &t, theta, X, n, p);
    for (int w = 0; w < n; ++w)                                              /* :524 */
        v.smooth[w] = pexpfma(theta + TH_SMOOTH * p, X, n, p, w);

    for (int ii = 0; ii < n; ++ii) {
        for (int jj = ii; jj < n; ++jj) {
            if (ii == jj) {                                                  /* :532-536 */
                out[ii + (size_t)ii * n] = pexpfma(sd, X, n, p, ii) + v.nugget[ii];
                continue;
            }
            double dx = locs[ii] - locs[jj];
            double dy = locs[ii + (size_t)n] - locs[jj + (size_t)n];
            double smtns = (v.smooth[ii] + v.smooth[jj]) / 2;                /* :554 */
            double det;
            double u = pair_geometry(&v, ii, &v, jj, dx, dy, smtns, global_range, &det);
            double val;
            if (u <= epsilon)
                val = pexpfma(sd, X, n, p, ii) + v.nugget[ii];
            else
                val = matern_general(smtns, u, v.sigma[ii], v.sigma[jj],
                                     pair_amplitude(&v, ii, &v, jj), det);
            out[ii + (size_t)jj * n] = out[jj + (size_t)ii * n] = val;
        }
    }
    free(t.two_scale_je);
    locvec_free(&v);
    return 0;
}

/* ---- cov_rns_pred: src/cocons_full.cpp:334-471 ------------------------------
 * out is m x n column-major, row = prediction location. */
int oracle_cov_rns_pred(int n, int m, int p, const double *theta, const double *locs,
                        const double *locs_pred, const double *X, const double *X_pred,
                        const double *smooth_limits, double *out)
{
    const double epsilon = DBL_EPSILON;
    locvec_t v, vp; thvec_t t;
    if (locvec_alloc(&v, n)) return -1;
    if (locvec_alloc(&vp, m)) { locvec_free(&v); return -1; }
    if (thvec_make(&t, theta, p)) { locvec_free(&v); locvec_free(&vp); return -1; }
    const double *sd = theta + TH_SD * p;
    double global_range = 1 / exp(-2 * theta[TH_SCALE * p + 0]);           /* :351 */
    locvec_fill_common(&v, &t, theta, X, n, p);
    locvec_fill_common(&vp, &t, theta, X_pred, m, p);
    for (int w = 0; w < n; ++w)                                              /* :381 */
        v.smooth[w] = sqrt(pexpfma_smooth(theta + TH_SMOOTH * p, X, n, p, w,
                                          smooth_limits[0], smooth_limits[1]));
    for (int w = 0; w < m; ++w)                                              /* :401 */
        vp.smooth[w] = sqrt(pexpfma_smooth(theta + TH_SMOOTH * p, X_pred, m, p, w,
                                           smooth_limits[0], smooth_limits[1]));

    for (int ii = 0; ii < m; ++ii) {
        for (int jj = 0; jj < n; ++jj) {
            double val;
            if (locs_pred[ii] == locs[jj] && locs_pred[ii + (size_t)m] == locs[jj + (size_t)n]) {
                val = pexpfma(sd, X_pred, m, p, ii) + vp.nugget[ii];        /* :410-414 */
            } else {
                double dx = locs_pred[ii] - locs[jj];
                double dy = locs_pred[ii + (size_t)m] - locs[jj + (size_t)n];
                double smtns = vp.smooth[ii] * v.smooth[jj];                /* :431 */
                double det;
                double u = pair_geometry(&vp, ii, &v, jj, dx, dy, smtns, global_range, &det);
                if (u <= epsilon)
                    val = pexpfma(sd, X_pred, m, p, ii) + vp.nugget[ii];    /* :440-442 */
                else
                    val = matern_general(smtns, u, vp.sigma[ii], v.sigma[jj],
                                         pair_amplitude(&vp, ii, &v, jj), det);
            }
            out[ii + (size_t)jj * m] = val;
        }
    }
    free(t.two_scale_je);
    locvec_free(&v);
    locvec_free(&vp);
    return 0;
}

/* ---- src/cocons_taper.cpp:151-433  cov_rns_taper -------------------------
 * CSR-indexed isotropic nonstationary Matern (the sparse/taper path's covariance entries).  colindices
 * and rowpointers are 1-based as spam stores them (the reference shifts them by -1 in place, :211-212);
 * entry w of row ii (0-based) pairs ii with jj = colindices[w] - 1.  Per entry, operation for operation:
 *   ii == jj          -> Pexp(std.dev, x_ii) + Pexp(nugget, x_ii)                             (:229-231)
 *   prefactor = (2 * pow(r_ii, 0.5) * pow(r_jj, 0.5)) / (r_ii + r_jj), r = Pexp(2 * scale)   (:236, :207)
 *   global_range = (r_ii + r_jj) / 2                                                          (:238)
 *   u = sqrt(8 nu) * sqrt(pow(dx, 2) + pow(dy, 2)) / sqrt(global_range)                       (:242-243)
 *   u <= eps          -> the diagonal value of ii                                             (:246-249)
 *   value = prefactor * M_nu(u) * sigma_ii * sigma_jj, sigma = Pexp(0.5 * std.dev)            (:253, ...)
 * with the same fixed-smoothness dispatch and quirk as cov_rns (a fixed nu outside {0.5, 1.5, 2.5} leaves
 * smooth_vector at zero, :173, :189-201 -> u = 0 -> every off-diagonal entry = diagonal value of ii).   */
int oracle_cov_rns_taper(int n, int p, const double *theta, const double *locs, const double *X,
                         const double *smooth_limits, int nnz, const int *colindices,
                         const int *rowpointers, double *out)
{
    const double epsilon = DBL_EPSILON;
    double *range_v = (double *)calloc((size_t)n * 3, sizeof(double));
    if (!range_v) return -1;
    double *sigma_v = range_v + n, *smooth_v = range_v + 2 * (size_t)n;
    double two_scale[64], half_sd[64];
    if (p > 64) { free(range_v); return -1; }
    for (int i = 0; i < p; ++i) {
        two_scale[i] = 2 * theta[TH_SCALE * p + i];
        half_sd[i] = 0.5 * theta[TH_SD * p + i];
    }
    const double *smooth = theta + TH_SMOOTH * p, *sd = theta + TH_SD * p, *ng = theta + TH_NUGGET * p;
    int smooth_switch = 0;
    double smooth_value = 0.0;
    if (all_zero_from_second(smooth, p) && smooth_limits[0] == smooth_limits[1]) {
        smooth_value = smooth_limits[0];
        smooth_switch = map_smooth_value(smooth_value);
    } else {
        for (int w = 0; w < n; ++w)
            smooth_v[w] = sqrt(pexpfma_smooth(smooth, X, n, p, w, smooth_limits[0], smooth_limits[1]));
    }
    for (int w = 0; w < n; ++w) {
        range_v[w] = pexpfma(two_scale, X, n, p, w);
        sigma_v[w] = pexpfma(half_sd, X, n, p, w);
    }
    int acc = 0;
    for (int ii = 0; ii < n; ++ii) {
        for (int w = rowpointers[ii] - 1; w < rowpointers[ii + 1] - 1; ++w) {
            const int jj = colindices[w] - 1;
            if (acc >= nnz) { free(range_v); return -2; }
            if (ii == jj) {
                out[acc++] = pexpfma(sd, X, n, p, ii) + pexpfma(ng, X, n, p, ii);
                continue;
            }
            const double smtns = (smooth_switch == 0) ? smooth_v[ii] * smooth_v[jj] : smooth_value;
            const double prefactor = (2 * pow(range_v[ii], 0.5) * pow(range_v[jj], 0.5)) / (range_v[ii] + range_v[jj]);
            const double global_range = (range_v[ii] + range_v[jj]) / 2;
            const double d0 = locs[ii] - locs[jj], d1 = locs[ii + (size_t)n] - locs[jj + (size_t)n];
            const double u = sqrt(8 * smtns) * sqrt(pow(d0, 2) + pow(d1, 2)) / sqrt(global_range);
            if (u <= epsilon) {
                out[acc++] = pexpfma(sd, X, n, p, ii) + pexpfma(ng, X, n, p, ii);
                continue;
            }
            double v;
            switch (smooth_switch) {
            case 1: v = prefactor * exp(-u) * sigma_v[ii] * sigma_v[jj]; break;
            case 2: v = prefactor * (1 + u) * exp(-u) * sigma_v[ii] * sigma_v[jj]; break;
            case 3: v = prefactor * (1 + u + u * u / 3) * exp(-u) * sigma_v[ii] * sigma_v[jj]; break;
            default:
                if (u < 706.0)
                    v = prefactor * pow(2.0, -(smtns - 1)) / tgamma(smtns) * pow(u, smtns) * oracle_besselk(smtns, u) *
                        sigma_v[ii] * sigma_v[jj];
                else
                    v = prefactor * pow(2.0, -(smtns - 1)) / tgamma(smtns) * pow(u, smtns) * sqrt(M_PI / (2.0 * u)) *
                        exp(-u) * sigma_v[ii] * sigma_v[jj];
            }
            out[acc++] = v;
        }
    }
    free(range_v);
    return acc == nnz ? 0 : -3;
}

/* ---- src/cocons_taper.cpp:17-139  cov_rns_taper_pred ----------------------
 * rows = prediction locations (rowpointers has m + 1 entries), columns = observation locations.  Always
 * the Bessel branch; coincident coordinates or u <= eps -> sigma_pred_ii * sigma_pred_ii +
 * Pexp(nugget, x_pred_ii) (:86-88, :104-107).                                                        */
int oracle_cov_rns_taper_pred(int n, int m, int p, const double *theta, const double *locs,
                              const double *locs_pred, const double *X, const double *X_pred,
                              const double *smooth_limits, int nnz, const int *colindices,
                              const int *rowpointers, double *out)
{
    const double epsilon = DBL_EPSILON;
    double *buf = (double *)calloc((size_t)(n + m) * 3, sizeof(double));
    if (!buf || p > 64) { free(buf); return -1; }
    double *scale_v = buf, *sigma_v = buf + n, *smooth_v = buf + 2 * (size_t)n;
    double *scale_p = buf + 3 * (size_t)n, *sigma_p = scale_p + m, *smooth_p = scale_p + 2 * (size_t)m;
    double two_scale[64], half_sd[64];
    for (int i = 0; i < p; ++i) {
        two_scale[i] = 2 * theta[TH_SCALE * p + i];
        half_sd[i] = 0.5 * theta[TH_SD * p + i];
    }
    const double *smooth = theta + TH_SMOOTH * p, *ng = theta + TH_NUGGET * p;
    for (int w = 0; w < n; ++w) {
        scale_v[w] = pexpfma(two_scale, X, n, p, w);
        sigma_v[w] = pexpfma(half_sd, X, n, p, w);
        smooth_v[w] = sqrt(pexpfma_smooth(smooth, X, n, p, w, smooth_limits[0], smooth_limits[1]));
    }
    for (int w = 0; w < m; ++w) {
        scale_p[w] = pexpfma(two_scale, X_pred, m, p, w);
        sigma_p[w] = pexpfma(half_sd, X_pred, m, p, w);
        smooth_p[w] = sqrt(pexpfma_smooth(smooth, X_pred, m, p, w, smooth_limits[0], smooth_limits[1]));
    }
    int acc = 0;
    for (int ii = 0; ii < m; ++ii) {
        for (int w = rowpointers[ii] - 1; w < rowpointers[ii + 1] - 1; ++w) {
            const int jj = colindices[w] - 1;
            if (acc >= nnz) { free(buf); return -2; }
            if (locs_pred[ii] == locs[jj] && locs_pred[ii + (size_t)m] == locs[jj + (size_t)n]) {
                out[acc++] = sigma_p[ii] * sigma_p[ii] + pexpfma(ng, X_pred, m, p, ii);
                continue;
            }
            const double smtns = smooth_p[ii] * smooth_v[jj];
            const double prefactor = (2 * pow(scale_p[ii], 0.5) * pow(scale_v[jj], 0.5)) / (scale_p[ii] + scale_v[jj]);
            const double global_range = (scale_p[ii] + scale_v[jj]) / 2;
            const double d0 = locs_pred[ii] - locs[jj], d1 = locs_pred[ii + (size_t)m] - locs[jj + (size_t)n];
            const double u = sqrt(8 * smtns) * sqrt(pow(d0, 2) + pow(d1, 2)) / sqrt(global_range);
            if (u <= epsilon) {
                out[acc++] = sigma_p[ii] * sigma_p[ii] + pexpfma(ng, X_pred, m, p, ii);
                continue;
            }
            if (u < 706.0)
                out[acc++] = prefactor * pow(2.0, -(smtns - 1)) / tgamma(smtns) * pow(u, smtns) * oracle_besselk(smtns, u) *
                             sigma_p[ii] * sigma_v[jj];
            else
                out[acc++] = prefactor * pow(2.0, -(smtns - 1)) / tgamma(smtns) * pow(u, smtns) * sqrt(M_PI / (2.0 * u)) *
                             exp(-u) * sigma_p[ii] * sigma_v[jj];
        }
    }
    free(buf);
    return acc == nnz ? 0 : -3;
}

/* ---- sumsmoothlone: src/cocons_full.cpp:12-30 ------------------------------- */
double oracle_sumsmoothlone(const double *x, int len, double lambda, double alpha)
{
    double sum = 0;
    for (int w = 0; w < len; ++w) {
        if (fabs(x[w]) > 1e-4)
            sum = sum + fabs(x[w]);
        else
            sum = sum + pow(alpha, -1) * (log(1 + exp(-alpha * x[w])) + log(1 + exp(alpha * x[w])));
    }
    return lambda * sum;
}

/* ---- long-double Cholesky truth for small n ---------------------------------
 * Returns 0, or k>0 if the leading minor of order k is not positive (LAPACK
 * dpotrf convention).  On success: *logdet_half = sum(log(diag(chol))) and
 * quad[c] = || L^{-1} rhs[:,c] ||^2 for the nrhs columns of rhs (n x nrhs,
 * column-major).  A is n x n column-major (only the lower triangle is read). */
int oracle_chol_ld(int n, const double *A, int nrhs, const double *rhs,
                   double *logdet_half, double *quad, double *Y)
{
    long double *L = (long double *)malloc((size_t)n * n * sizeof(long double));
    if (!L) return -1;
    for (int j = 0; j < n; ++j)
        for (int i = j; i < n; ++i)
            L[i + (size_t)j * n] = A[i + (size_t)j * n];
    long double ld = 0;
    for (int j = 0; j < n; ++j) {
        long double d = L[j + (size_t)j * n];
        for (int k = 0; k < j; ++k) d -= L[j + (size_t)k * n] * L[j + (size_t)k * n];
        if (!(d > 0)) { free(L); return j + 1; }
        d = sqrtl(d);
        L[j + (size_t)j * n] = d;
        ld += logl(d);
        for (int i = j + 1; i < n; ++i) {
            long double s = L[i + (size_t)j * n];
            for (int k = 0; k < j; ++k) s -= L[i + (size_t)k * n] * L[j + (size_t)k * n];
            L[i + (size_t)j * n] = s / d;
        }
    }
    *logdet_half = (double)ld;
    long double *y = (long double *)malloc((size_t)n * sizeof(long double));
    for (int c = 0; c < nrhs; ++c) {
        long double q = 0;
        for (int i = 0; i < n; ++i) {
            long double s = rhs[i + (size_t)c * n];
            for (int k = 0; k < i; ++k) s -= L[i + (size_t)k * n] * y[k];
            y[i] = s / L[i + (size_t)i * n];
            q += y[i] * y[i];
            if (Y) Y[i + (size_t)c * n] = (double)y[i];
        }
        quad[c] = (double)q;
    }
    free(y);
    free(L);
    return 0;
}
