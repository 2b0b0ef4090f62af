// matern_device.hpp -- per-pair device arithmetic of the nonstationary Matern kernel.
//
// Follows the operation order of the reference's pair loop
// (src/cocons_full.cpp:122-153 and its copies at :168-199, :214-245, :260-297,
// :418-461, :540-584) so that results agree with the CPU path to a few ulp:
// the compensated 2x2 products of src/cocons_types.h:49-54 are kept as explicit
// fma sequences (this file must be compiled with -ffp-contract=off).
//
// K_nu(x) replaces boost::math::cyl_bessel_k (call sites :294, :450, :573) with
// the same published scheme: Temme's series (x <= 2), Steed's CF2 (x > 2),
// forward recurrence from mu = nu - round(nu).  The reciprocal-gamma pieces come
// from the Taylor table in rgamma_coeffs.h, which also yields 1/Gamma(nu), so no
// tgamma call is needed per pair.
#pragma once
#include <hip/hip_runtime.h>
#include "rgamma_coeffs.h"

namespace cocons {

__constant__ double c_rg_even[RG_NTERMS] = RG_EVEN_INIT;
__constant__ double c_rg_odd[RG_NTERMS] = RG_ODD_INIT;

enum PairMode : int {
    MODE_HALF = 1,       // nu = 0.5 closed form   (cocons_full.cpp:117-160)
    MODE_THREEHALF = 2,  // nu = 1.5               (:163-206)
    MODE_FIVEHALF = 3,   // nu = 2.5               (:209-252)
    MODE_GEOM = 0,       // general, nu_ij = snu_i * snu_j  (:255-315, :407-468)
    MODE_MEAN = 4        // classic, nu_ij = (nu_i + nu_j)/2 (:529-591)
};

// per-location SoA written by loc_params_kernel: field f of location w at base[w + f*stride]
//   0 x, 1 y            coordinates
//   2 rd                Pexp(2*scale_je)                 range_det_vector
//   3 an2               an*an                            aniso_det_vector squared
//   4 ra                rd*an
//   5 ct, 6 st          cos(tilt), sin(tilt)
//   7 dets              Pexp(2*scale_je + aniso)
//   8 ds                dets * st
//   9 sigma             Pexp(0.5*std.dev)
//  10 snu               sqrt(nu_w) (GEOM) or nu_w (MEAN)
//  11 diag              Pexp(std.dev) + nugget
//  12 ng                Pexp(nugget)   (taper prediction variant: sigma^2 + ng, src/cocons_taper.cpp:88)
constexpr int LOCP_FIELDS = 13;

// src/cocons_types.h:49-54
__device__ __forceinline__ double kahan(double a, double b, double c, double d)
{
    double cd = c * d;
    double err = fma(c, d, -cd);
    double res = fma(a, b, -cd);
    return res - err;
}

// 1/x for moderate x: v_rcp_f64 seed (24 bits, measured: tools/diag/seed_precision.hip) + ONE third-order step
// r (1 + e + e^2), e = 1 - x r (truncation e^3 < 2^-72; |rel err| ~ 1e-16; no scaling, the operands below are
// O(1)..O(1e3)) -- three dependent operations where two Newton steps took four
__device__ __forceinline__ double fast_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(fma(e, e, e), r, r);
}

// sqrt(x) for x >= 0 of ordinary magnitude (squared distances over squared ranges, ratios of determinants): v_rsq_f64 (24
// bits) + one third-order step + one Heron correction, ~1 ulp in ten instructions; zero stays zero.  The library square root
// spends as many again on scaling for operands near the ends of the exponent range, which do not occur here.
// A NEGATIVE or NaN argument comes out as NaN, like sqrt's (v_rsq_f64 gives NaN for both): a quadratic form that degenerate
// parameters turned into NaN -- overflowing link functions, inf - inf in the averaged kernel matrix -- must poison the entry
// as it does in the reference (src/cocons_full.cpp:286-305: NaN fails both branch tests and reaches Sigma; the Cholesky then
// fails and GetNeg2loglikelihood's tryCatch contract fires, R/neg2loglikelihood.R:200-206).  Until round 4 this returned 0
// for anything not > 0, which sent such a pair down the `u <= epsilon` branch with the DIAGONAL value.  (+inf gives NaN
// here where sqrt gives +inf: either way the entry is not a number the reference would have produced a finite value from.)
__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y0 = __builtin_amdgcn_rsq(x);
    const double s0 = x * y0;
    const double t = fma(-s0, y0, 1.0);
    const double y = fma(y0 * t, fma(t, 0.375, 0.5), y0);
    double s_ = x * y;
    s_ = fma(fma(-s_, s_, x), 0.5 * y, s_);
    return x == 0.0 ? 0.0 : s_;
}

// One Horner step p t + C with the coefficient in a SCALAR register pair.  Written out because the compiler's own choice for
// fma(p, t, literal) in straight-line code is v_fmac_f64 with the constant moved into a fresh vector register pair first:
// two v_mov_b32 and the fused multiply-add, three vector instructions per coefficient in a kernel that is bound by vector
// issue (the scalar moves this form needs instead issue beside them).  Same operation, same bits.
__device__ __forceinline__ double hfma(double p, double t, double C)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(t), "s"(C));
    return r;
}

// log2(u) for a normal positive u (the branches below call it with 2 <= u < 1100): u = 2^e m with m in [sqrt(1/2), sqrt 2),
// log m = 2 s (1 + s^2/3 + s^4/5 + ...) in s = (m - 1)/(m + 1), |s| <= 0.172 (the term behind s^22/23 is 2e-20).  Absolute error
// ~2e-16, which the exponent nu log2(u) + 1 - nu it goes into turns into a relative 5e-16 of the result; the library routine
// pays seventy instructions of double-double arithmetic for its last half ulp.
__device__ __forceinline__ double log2_pos(double u)
{
    double m = __builtin_amdgcn_frexp_mant(u);            // [1/2, 1)
    int e = __builtin_amdgcn_frexp_exp(u);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double s_ = (m - 1.0) * fast_rcp(m + 1.0);
    const double s2 = s_ * s_;
    double p = 1.0 / 23.0;
    p = hfma(p, s2, 1.0 / 21.0);
    p = hfma(p, s2, 1.0 / 19.0);
    p = hfma(p, s2, 1.0 / 17.0);
    p = hfma(p, s2, 1.0 / 15.0);
    p = hfma(p, s2, 1.0 / 13.0);
    p = hfma(p, s2, 1.0 / 11.0);
    p = hfma(p, s2, 1.0 / 9.0);
    p = hfma(p, s2, 1.0 / 7.0);
    p = hfma(p, s2, 1.0 / 5.0);
    p = hfma(p, s2, 1.0 / 3.0);
    p = hfma(p, s2, 1.0);
    return fma(p * s_, 2.0 * 1.4426950408889634074, (double)e);
}

// 2^a e^-u for |a| < 1000, 0 <= u < 1100, in ONE exponential: u = n_u ln 2 + r (Cody-Waite, r exact to ~1e-17), a = n_a + f_a
// (exact), so 2^a e^-u = 2^(n_a - n_u + k) 2^g with g = f_a - r log2(e) - k in [-1/2, 1/2]; 2^g = e^(g ln 2) by its Taylor series
// to degree 14 (|g ln 2| <= 0.347: truncation 1e-19).  Relative error ~4e-16; replaces exp2(a) * exp(-u) (two library
// calls, ~60 instructions) by ~30.
__device__ __forceinline__ double pow2a_expmu(double a, double u)
{
    const double L2E = 1.4426950408889634074, LN2 = 0.6931471805599453094;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double nu_ = rint(u * L2E);
    double r = fma(-nu_, LN2_HI, u);
    r = fma(-nu_, LN2_LO, r);
    const double na = rint(a);
    const double f = fma(-r, L2E, a - na);
    const double k = rint(f);
    const double t = (f - k) * LN2;
    double p = 1.0 / 87178291200.0;                   // 1/14!
    p = hfma(p, t, 1.0 / 6227020800.0);
    p = hfma(p, t, 1.0 / 479001600.0);
    p = hfma(p, t, 1.0 / 39916800.0);
    p = hfma(p, t, 1.0 / 3628800.0);
    p = hfma(p, t, 1.0 / 362880.0);
    p = hfma(p, t, 1.0 / 40320.0);
    p = hfma(p, t, 1.0 / 5040.0);
    p = hfma(p, t, 1.0 / 720.0);
    p = hfma(p, t, 1.0 / 120.0);
    p = hfma(p, t, 1.0 / 24.0);
    p = hfma(p, t, 1.0 / 6.0);
    p = hfma(p, t, 0.5);
    p = hfma(p, t, 1.0);
    p = hfma(p, t, 1.0);
    return ldexp(p, (int)(na - nu_ + k));
}

__constant__ double c_inv_k[64] = {
    0.0, 1.0, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6, 1.0 / 7, 1.0 / 8, 1.0 / 9, 1.0 / 10, 1.0 / 11, 1.0 / 12,
    1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16, 1.0 / 17, 1.0 / 18, 1.0 / 19, 1.0 / 20, 1.0 / 21, 1.0 / 22, 1.0 / 23,
    1.0 / 24, 1.0 / 25, 1.0 / 26, 1.0 / 27, 1.0 / 28, 1.0 / 29, 1.0 / 30, 1.0 / 31, 1.0 / 32, 1.0 / 33, 1.0 / 34,
    1.0 / 35, 1.0 / 36, 1.0 / 37, 1.0 / 38, 1.0 / 39, 1.0 / 40, 1.0 / 41, 1.0 / 42, 1.0 / 43, 1.0 / 44, 1.0 / 45,
    1.0 / 46, 1.0 / 47, 1.0 / 48, 1.0 / 49, 1.0 / 50, 1.0 / 51, 1.0 / 52, 1.0 / 53, 1.0 / 54, 1.0 / 55, 1.0 / 56,
    1.0 / 57, 1.0 / 58, 1.0 / 59, 1.0 / 60, 1.0 / 61, 1.0 / 62, 1.0 / 63};

constexpr double HANKEL_U0 = 20.0;
constexpr int HANKEL_TERMS = 20;

// 2^(1-nu)/Gamma(nu) * u^nu * K_nu(u) for 0 < u < 706; for u >= 706 the reference's asymptotic stand-in (see below)
__device__ __noinline__ double matern_bessel(double nu, double u)
{
    const double tol = 2.220446049250313e-16;
    const double pi = 3.14159265358979323846;
    int n = (int)floor(nu + 0.5);
    double mu = nu - n;
    double mu2 = mu * mu;
    double gam1 = 0.0, gam2 = 0.0;
#pragma unroll
    for (int j = RG_NTERMS - 1; j >= 0; --j) {
        gam1 = hfma(gam1, mu2, c_rg_odd[j]);
        gam2 = hfma(gam2, mu2, c_rg_even[j]);
    }
    double gampl = gam2 - mu * gam1;   // 1/Gamma(1+mu)
    double gammi = gam2 + mu * gam1;   // 1/Gamma(1-mu)
    // (nu <= 3.5: what the 20 terms are validated for -- at u = 20 the truncation error grows to 1e-14 at nu = 8 and
    // 1e-12 at nu = 15; smooth_limits come from the user, so larger orders keep the continued fraction)
    // u >= 706: the reference switches to the LEADING term of this very series, sqrt(pi / 2u) e^-u, for every nu
    // (src/cocons_full.cpp:301-305; values below 1e-300 that underflow to zero near u = 745) -- the same code with S = 1
    // instead of a pow / tgamma / exp chain of library calls (range 0.02 at n = 10^4: 2.0 -> 1.3 ms assembly)
    if (u >= HANKEL_U0 && (nu <= 3.5 || u >= 706.0)) {
        // Large arguments: Hankel's asymptotic series  K_nu(u) ~ sqrt(pi / 2u) e^-u sum_k a_k(nu) / u^k,
        // a_k = prod_{j<=k} (4 nu^2 - (2j-1)^2) / (8 j), directly at order nu (no recurrence from mu).  For
        // u >= 20 and nu <= 3.5 twenty terms leave a truncation error below 1.5e-16 (checked against mpmath over
        // nu in [0.25, 3.5], tests/test_gpu_parity.py); the work is the same for every lane, where CF2 takes
        // 8..12 data-dependent steps of ~25 instructions in this range.  1/Gamma(nu) from the same table.
        const double w = fast_rcp(u), p4 = 4.0 * nu * nu * w;
        double t = 1.0, S = 1.0;
#pragma unroll
        for (int k = 1; k <= HANKEL_TERMS / 2; ++k) {
            // t_k = t_{k-1} (4 nu^2 - (2k-1)^2) / (8 k u); the two constants fold at compile time
            const double Ak = 1.0 / (8.0 * k), Bk = (double)((2 * k - 1) * (2 * k - 1)) / (8.0 * k);
            t *= fma(p4, Ak, -Bk * w);
            S += t;
        }
        // (far out -- u >= 45 or so, most pairs of a short correlation range -- the tenth term is already below half an ulp of
        // the sum and every later one is smaller still in this range of k: the second half is skipped by the waves that can)
        if (fabs(t) > 1.0e-17 * S) {
#pragma unroll
            for (int k = HANKEL_TERMS / 2 + 1; k <= HANKEL_TERMS; ++k) {
                const double Ak = 1.0 / (8.0 * k), Bk = (double)((2 * k - 1) * (2 * k - 1)) / (8.0 * k);
                t *= fma(p4, Ak, -Bk * w);
                S += t;
            }
        }
        if (u >= 706.0) S = 1.0;
        double rg = (n == 0) ? mu * gampl : gampl;
        double prod = 1.0;
        for (int k = 1; k < n; ++k) prod *= (mu + k);
        rg = rg * fast_rcp(prod);
        // 2^(1-nu) u^nu sqrt(pi / 2u) e^-u = sqrt(pi/2) 2^((nu - 1/2) log2 u + 1 - nu) e^-u: one exponential
        return 1.2533141373155002512 * pow2a_expmu(fma(nu - 0.5, log2_pos(u), 1.0 - nu), u) * rg * S;
    }
#ifndef COCONS_TRAP_ULO
#define COCONS_TRAP_ULO 2.0
#endif
#ifndef COCONS_TRAP_UHI
#define COCONS_TRAP_UHI 20.0
#endif
    if (u >= COCONS_TRAP_ULO && u < COCONS_TRAP_UHI && nu <= 3.5) {
        // The middle band (round 4): the integral  e^u K_nu(u) = int_0^inf exp(-u (cosh t - 1)) cosh(nu t) dt  by the trapezoid
        // rule, which converges geometrically for an analytic even integrand: with  1/h^2 = 1/0.24^2 + u/0.78^2  the
        // discretisation error is below 4e-15 for nu <= 3.5 and 0.5 <= u <= 30 (mpmath, tests/test_gpu_parity.py), and the
        // terms fall below one ulp of the sum after 13 (u = 8 .. 20) to 22 (u = 0.75) nodes.  Per node one exponential and
        // two three-term recurrences -- cosh(j h) - 1 and cosh(nu j h) -- directly at order nu: no series set-up (Temme: a
        // logarithm, an exponential, two reciprocals before the first term), no order recurrence, and a cost that hardly
        // depends on u, where Steed's CF2 takes 6 steps at u = 19 and 50 at u = 2.1 (the band that made the assembly 2.5 ms
        // at correlation range 1.0 against 1.2 ms at 0.05, n = 10^4).  Every term is positive: no cancellation anywhere.
        const double L2E = 1.4426950408889634074;
        const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
        double hh = fma(u, 1.0 / (0.78 * 0.78), 1.0 / (0.24 * 0.24));      // 1 / h^2
        double h = __builtin_amdgcn_rsq(hh);
        h = h * fma(fma(-hh * h, h, 1.0), 0.5, 1.0);                       // one Newton step: 48 bits, h is a free parameter
        const double h2 = h * h;
        // kappa = cosh(h) - 1 = h^2/2 (1 + h^2/12 (1 + h^2/30 (1 + h^2/56 (1 + h^2/90 (1 + h^2/132 (1 + h^2/182))))))
        double kap = fma(h2, 1.0 / 182.0, 1.0);
        kap = fma(kap * h2, 1.0 / 132.0, 1.0);
        kap = fma(kap * h2, 1.0 / 90.0, 1.0);
        kap = fma(kap * h2, 1.0 / 56.0, 1.0);
        kap = fma(kap * h2, 1.0 / 30.0, 1.0);
        kap = fma(kap * h2, 1.0 / 12.0, 1.0);
        kap = kap * (0.5 * h2);
        // cosh(nu h): |nu h| <= 0.77, Taylor polynomial in (nu h)^2 to degree 11 (0.77^24 / 24! = 3e-27)
        const double xx = nu * h, x2 = xx * xx;
        double chn = 1.0 / 51090942171709440000.0 * (1.0 / 22.0);         // 1/22!
        chn = hfma(chn, x2, 1.0 / 2432902008176640000.0);                   // 1/20!
        chn = hfma(chn, x2, 1.0 / 6402373705728000.0);                      // 1/18!
        chn = hfma(chn, x2, 1.0 / 20922789888000.0);                        // 1/16!
        chn = hfma(chn, x2, 1.0 / 87178291200.0);                           // 1/14!
        chn = hfma(chn, x2, 1.0 / 479001600.0);                             // 1/12!
        chn = hfma(chn, x2, 1.0 / 3628800.0);                               // 1/10!
        chn = hfma(chn, x2, 1.0 / 40320.0);                                 // 1/8!
        chn = hfma(chn, x2, 1.0 / 720.0);                                   // 1/6!
        chn = hfma(chn, x2, 1.0 / 24.0);                                    // 1/4!
        chn = hfma(chn, x2, 0.5);
        chn = hfma(chn, x2, 1.0);
        const double tk = 2.0 * kap, tc = 2.0 * chn;
        double c_prev = 0.0, c = kap;        // cosh(j h) - 1 for j - 1, j
        double w_prev = 1.0, w = chn;        // cosh(nu j h)
        double S = 0.5;
        // exp(-u c): -u c = k ln 2 + r, e^r by its Taylor polynomial to degree 13 (|r| <= 0.347: 4e-18)
        auto node_exp = [&](double cc) {
            const double x = -u * cc;
            const double k = rint(x * L2E);
            double r = fma(-k, LN2_HI, x);
            r = fma(-k, LN2_LO, r);
            double pe = 1.0 / 6227020800.0;
            pe = fma(pe, r, 1.0 / 479001600.0);
            pe = fma(pe, r, 1.0 / 39916800.0);
            pe = fma(pe, r, 1.0 / 3628800.0);
            pe = fma(pe, r, 1.0 / 362880.0);
            pe = fma(pe, r, 1.0 / 40320.0);
            pe = fma(pe, r, 1.0 / 5040.0);
            pe = fma(pe, r, 1.0 / 720.0);
            pe = fma(pe, r, 1.0 / 120.0);
            pe = fma(pe, r, 1.0 / 24.0);
            pe = fma(pe, r, 1.0 / 6.0);
            pe = fma(pe, r, 0.5);
            pe = fma(pe, r, 1.0);
            pe = fma(pe, r, 1.0);
            return ldexp(pe, (int)k);
        };
        // two nodes per round, one test (a node too many costs less than asking after every one)
        for (int j = 1; j < 64; j += 2) {
            const double t1 = node_exp(c) * w;
            const double c1 = fma(tk, c + 1.0, fma(2.0, c, -c_prev));
            const double w1 = fma(tc, w, -w_prev);
            const double t2 = node_exp(c1) * w1;
            S += t1;
            S += t2;
            if (t2 < S * (4.0 * tol)) break;       // (what is left behind the last node is a fraction of it)
            c_prev = c1; c = fma(tk, c1 + 1.0, fma(2.0, c1, -c));
            w_prev = w1; w = fma(tc, w1, -w);
        }
        double rg = (n == 0) ? mu * gampl : gampl;
        double prod = 1.0;
        for (int k = 1; k < n; ++k) prod *= (mu + k);
        rg = rg * fast_rcp(prod);
        return pow2a_expmu(fma(nu, log2_pos(u), 1.0 - nu), u) * rg * (S * h);
    }
    double kmu, kmu1;                  // K_mu, K_{mu+1}, both WITHOUT the factor exp(-u) when u > 2
    double escale;                     // the factor still to be applied: exp(-u) (CF2) or 1 (Temme)
    double l2u;                        // log2(u) for the final power of two
    if (u <= 2.0) {
        // Temme's series.  Its set-up used to be five library calls (sin, log, sinh, cosh, exp: ~250 instructions, more
        // than the series itself); now ONE log and ONE exp: sin(pi mu) on |pi mu| <= pi/2 by its Taylor polynomial
        // (degree 21: 1e-18), cosh(e) and sinh(e)/e from E = exp(e) and 1/E (sinh(e)/e by its series where |e| < 1/4: the
        // difference E - 1/E cancels there), and log2(u) = 1 - d log2(e) from the logarithm already taken.
        double x2 = 0.5 * u, pimu = pi * mu;
        const double y = pimu * pimu;
        double sp = -1.0 / 51090942171709440000.0;          // -1/21!
        sp = hfma(sp, y, 1.0 / 121645100408832000.0);         //  1/19!
        sp = hfma(sp, y, -1.0 / 355687428096000.0);           // -1/17!
        sp = hfma(sp, y, 1.0 / 1307674368000.0);              //  1/15!
        sp = hfma(sp, y, -1.0 / 6227020800.0);                // -1/13!
        sp = hfma(sp, y, 1.0 / 39916800.0);                   //  1/11!
        sp = hfma(sp, y, -1.0 / 362880.0);                    // -1/9!
        sp = hfma(sp, y, 1.0 / 5040.0);                       //  1/7!
        sp = hfma(sp, y, -1.0 / 120.0);                       // -1/5!
        sp = hfma(sp, y, 1.0 / 6.0);                          //  1/3!
        sp = fma(-sp, y, 1.0);                               // sin(x)/x = 1 - x^2 (1/3! - x^2 (1/5! - ...))
        double fact = fast_rcp(sp);                          // pi mu / sin(pi mu)
        // (d = -ln(u/2) = ln 2 (1 - log2 u) from the kernel's own log2; E = e^(mu d) = 2^(mu (1 - log2 u)) through the one
        // exponential routine: the library's log and exp cost 110 instructions between them)
        l2u = log2_pos(u);
        const double oml = 1.0 - l2u;
        double d = oml * 0.6931471805599453094, e = mu * d;
        const double E = pow2a_expmu(mu * oml, 0.0), Ei = fast_rcp(E);
        const double e2 = e * e;
        double sh = 1.0 / 6227020800.0;                       // sinh(e)/e = sum e^(2k) / (2k+1)!, |e| < 1/4: 1e-19 after e^12
        sh = hfma(sh, e2, 1.0 / 39916800.0);
        sh = hfma(sh, e2, 1.0 / 362880.0);
        sh = hfma(sh, e2, 1.0 / 5040.0);
        sh = hfma(sh, e2, 1.0 / 120.0);
        sh = hfma(sh, e2, 1.0 / 6.0);
        sh = hfma(sh, e2, 1.0);
        double fact2 = fabs(e) < 0.25 ? sh : 0.5 * (E - Ei) * fast_rcp(e);
        double ff = fact * (gam1 * (0.5 * (E + Ei)) + gam2 * fact2 * d);
        double sum = ff;
        e = E;
        double pp = 0.5 * e / gampl, q = 0.5 / (e * gammi), c = 1.0;
        d = x2 * x2;
        double sum1 = pp;
        for (int i = 1; i < 500; ++i) {
            double di = (double)i;
            double inv = fast_rcp((di - mu) * (di + mu));
            ff = (di * ff + pp + q) * inv;
            c *= d * ((i < 64) ? c_inv_k[i] : 1.0 / di);
            pp *= inv * (di + mu);
            q *= inv * (di - mu);
            double del = c * ff;
            sum += del;
            sum1 += c * (pp - di * ff);
            if (fabs(del) < fabs(sum) * tol) break;
        }
        kmu = sum;
        kmu1 = sum1 * (2.0 / u);
        escale = 1.0;
    } else {
        // Steed's CF2.  With A_k = C_k q_k and B_k = C_k q_{k-1} the recurrences
        //   q_k = (q_{k-2} - (b_k - 2) q_{k-1}) / a_k ,  C_k = -C_{k-1} a_k / k
        // become  A_k = -(B_{k-1} - (b_k - 2) A_{k-1}) / k ,  B_k = -(a_k / k) A_{k-1}
        // (a_k cancels), leaving one reciprocal per step.
        l2u = log2_pos(u);
        double a = mu2 - 0.25;
        double b = 2.0 * (u + 1.0), D = fast_rcp(b), f = D, delta = D;
        double Ak = -a;          // C_1 q_1 = -a * 1
        double Bk = 0.0;         // C_1 q_0 = 0
        double Q = Ak, S = 1.0 + Q * delta;
        for (int k = 2; k < 500; ++k) {
            a -= 2 * (k - 1);
            b += 2.0;
            D = fast_rcp(fma(a, D, b));
            delta *= fma(b, D, -1.0);
            f += delta;
            double ik = (k < 64) ? c_inv_k[k] : 1.0 / (double)k;
            double An = -(Bk - (b - 2.0) * Ak) * ik;
            Bk = -(a * ik) * Ak;
            Ak = An;
            Q += Ak;
            double qd = Q * delta;
            S += qd;
            if (fabs(qd) < fabs(S) * tol) break;
        }
        kmu = sqrt(pi / (2.0 * u)) / S;
        kmu1 = kmu * (0.5 + mu + u + (mu2 - 0.25) * f) / u;
        escale = 0.0;                  // marks: exp(-u) still to be applied (folded into the final power of two)
    }
    // forward recurrence to order nu, and 1/Gamma(nu) = gampl / prod_{k=1}^{n-1} (mu+k)
    double rg = (n == 0) ? mu * gampl : gampl;
    double pk = kmu, ck = kmu1;
    double two_over_u = 2.0 / u;
    double prod = 1.0;
    for (int k = 1; k <= n; ++k) {
        double next = fma((mu + k) * two_over_u, ck, pk);
        pk = ck;
        ck = next;
        if (k < n) prod *= (mu + k);
    }
    rg = rg * fast_rcp(prod);
    // 2^(1-nu) u^nu = 2^(nu log2 u + 1 - nu): the exponent stays O(25), so its rounding is harmless; times e^-u on the CF2 side
    const double ex = fma(nu, l2u, 1.0 - nu);
    return (escale == 0.0 ? pow2a_expmu(ex, u) : exp2(ex)) * rg * pk;
}

// Value of one covariance entry from the per-location SoA; ia = first ("ii") location,
// ib = second ("jj").  `base_a`/`base_b` are the SoAs of the two sides.  The fields are
// loaded in two stages -- geometry before the Bessel call, amplitude after it -- and the
// lane-varying index is laundered through an empty asm so that the loads are NOT hoisted
// out of the caller's column loop: nothing but (det, m) stays live across the call, which
// is what lets the kernel run at 4 waves/SIMD without scratch.
// gr = global_range = 1/exp(-2*scale[0]); nu_fixed is used by the closed-form modes.
template <int MODE>
__device__ __forceinline__ double pair_value_idx(const double *base_a, size_t sa, int ia,
                                                 const double *base_b, size_t sb, int ib,
                                                 double gr, double nu_fixed, bool coincident_check)
{
    const double epsilon = 2.220446049250313e-16;
    const double rgr = 1.0 / gr;                        // (wave-uniform: once per kernel)
    // field f of location i: a wave-uniform base (base + f * stride: scalar registers) plus ONE 32-bit byte offset per
    // side -- per-lane 64-bit pointer arithmetic for every field was a twentieth of the kernel's instructions
#define FA(f) (*(const double *)((const char *)(base_a + (size_t)(f) * sa) + oa))
#define FB(f) (*(const double *)((const char *)(base_b + (size_t)(f) * sb) + ob))
    unsigned oa = 8u * (unsigned)ia, ob = 8u * (unsigned)ib;
    double ax = FA(0), ay = FA(1), bx = FB(0), by = FB(1);
    if (coincident_check && ax == bx && ay == by) return FA(11);    // cocons_full.cpp:410-414
    double ard = FA(2), aan2 = FA(3), ara = FA(4), act = FA(5);
    double brd = FB(2), ban2 = FB(3), bra = FB(4), bct = FB(5);
    double s11 = (ard + brd) * 0.5;
    double s22 = kahan(ard, aan2, -brd, ban2) * 0.5;
    double s12 = kahan(ara, act, -bra, bct) * 0.5;
    double det = kahan(s11, s22, s12, s12);
    double dx = ax - bx, dy = ay - by;
    double smtns;
    if (MODE == MODE_GEOM) smtns = FA(10) * FB(10);
    else if (MODE == MODE_MEAN) smtns = (FA(10) + FB(10)) / 2;
    else smtns = nu_fixed;
    // u = sqrt(8 nu / (gr det)) sqrt(q) as ONE square root over ONE reciprocal of det (the reference takes two roots and a
    // division, :140-141; the same reciprocal serves the normalisation below): u moves by a few ulp, M(u) by u times that --
    // below 3e-13 at u = 700, inside the entrywise tolerance of 2e-12
    const double rdet = fast_rcp(det);
    double u = sqrt_pos((8 * smtns) * (rgr * rdet) *
                        fma(kahan(s22, dx * dx, -s11, dy * dy), 1.0, -2 * s12 * dx * dy));
    if (u <= epsilon) return FA(11);
    if (u != u) return u;       // NaN geometry (degenerate parameters): the entry is NaN, as in the reference (sqrt_pos)
    double m;
    if (MODE == MODE_HALF) m = exp(-u);
    else if (MODE == MODE_THREEHALF) m = (1 + u) * exp(-u);
    else if (MODE == MODE_FIVEHALF) m = (1 + u + u * u / 3) * exp(-u);
    else m = matern_bessel(smtns, u);
    // stage 2: amplitude (fields reloaded, see above)
    asm volatile("" : "+v"(oa), "+v"(ob));
    // (1 / det is formed again rather than kept across the Bessel call: one more live value there spills)
    double amp = sqrt_pos(FA(8) * FB(8) * fast_rcp(det));      // sqrt(dets_i sin t_i dets_j sin t_j / det), field 8 = dets sin t
    return m * FA(9) * FB(9) * amp;
#undef FA
#undef FB
}

// One entry of the sparse/taper covariance (src/cocons_taper.cpp:229-262 and its copies, :86-129):
// isotropic, local range r = Pexp(2 scale) (field 2 -- the caller passes the FULL scale vector here),
// sigma = Pexp(0.5 std.dev) (field 9), sqrt(nu) (field 10).  ia = row ("ii") location, ib = column.
// PRED: exact coordinate match or u <= eps -> sigma_ii * sigma_ii + Pexp(nugget)_ii (:88, :106),
// otherwise (cov_rns_taper) u <= eps -> the diagonal value of ii (:248); ii == jj is the caller's case.
template <int MODE, bool PRED>
__device__ __forceinline__ double taper_value_idx(const double *base_a, size_t sa, int ia,
                                                  const double *base_b, size_t sb, int ib, double nu_fixed)
{
    const double epsilon = 2.220446049250313e-16;
    const double *pa = base_a + ia, *pb = base_b + ib;
    const double ax = pa[0], ay = pa[sa], bx = pb[0], by = pb[sb];
    const double si = pa[9 * sa];
    const double own = PRED ? si * si + pa[12 * sa] : pa[11 * sa];
    if (PRED && ax == bx && ay == by) return own;
    const double ri = pa[2 * sa], rj = pb[2 * sb];
    const double smtns = (MODE == MODE_GEOM) ? pa[10 * sa] * pb[10 * sb] : nu_fixed;
    const double prefactor = (2 * sqrt(ri) * sqrt(rj)) / (ri + rj);
    const double global_range = (ri + rj) / 2;
    const double dx = ax - bx, dy = ay - by;
    const double u = sqrt(8 * smtns) * sqrt(dx * dx + dy * dy) / sqrt(global_range);
    if (u <= epsilon) return own;
    double m;
    if (MODE == MODE_HALF) m = exp(-u);
    else if (MODE == MODE_THREEHALF) m = (1 + u) * exp(-u);
    else if (MODE == MODE_FIVEHALF) m = (1 + u + u * u / 3) * exp(-u);
    else m = matern_bessel(smtns, u);
    return prefactor * m * si * pb[9 * sb];
}

}  // namespace cocons
