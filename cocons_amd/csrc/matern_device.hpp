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

// per-location quantities, one struct per side of a pair
struct LocP {
    double x, y;     // coordinates
    double rd;       // Pexp(2*scale_je)           range_det_vector
    double an2;      // an*an                       aniso_det_vector squared
    double ra;       // rd*an
    double ct, st;   // cos(tilt), sin(tilt)
    double dets;     // Pexp(2*scale_je + aniso)
    double ds;       // dets * st
    double sigma;    // Pexp(0.5*std.dev)
    double snu;      // sqrt(nu_w) (GEOM) or nu_w (MEAN)
    double diag;     // Pexp(std.dev) + nugget
};
constexpr int LOCP_FIELDS = 12;

// src/cocons_types.h:49-54
__device__ __forceinline__ double kahan(double a, double b, double c, double d)
{
    double cd = c * d;
    double err = fma(c, d, -cd);
    double res = fma(a, b, -cd);
    return res - err;
}

// 2^(1-nu)/Gamma(nu) * u^nu * K_nu(u), 0 < u < 706
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
        gam1 = fma(gam1, mu2, c_rg_odd[j]);
        gam2 = fma(gam2, mu2, c_rg_even[j]);
    }
    double gampl = gam2 - mu * gam1;   // 1/Gamma(1+mu)
    double gammi = gam2 + mu * gam1;   // 1/Gamma(1-mu)
    double kmu, kmu1;
    if (u <= 2.0) {
        double x2 = 0.5 * u, pimu = pi * mu;
        double fact = fabs(pimu) < tol ? 1.0 : pimu / sin(pimu);
        double d = -log(x2), e = mu * d;
        double fact2 = fabs(e) < tol ? 1.0 : sinh(e) / e;
        double ff = fact * (gam1 * cosh(e) + gam2 * fact2 * d);
        double sum = ff;
        e = exp(e);
        double pp = 0.5 * e / gampl, q = 0.5 / (e * gammi), c = 1.0;
        d = x2 * x2;
        double sum1 = pp;
        for (int i = 1; i < 500; ++i) {
            double di = (double)i;
            double inv = 1.0 / ((di - mu) * (di + mu));
            ff = (di * ff + pp + q) * inv;
            c *= d / di;
            pp *= inv * (di + mu);
            q *= inv * (di - mu);
            double del = c * ff;
            sum += del;
            sum1 += c * (pp - di * ff);
            if (fabs(del) < fabs(sum) * tol) break;
        }
        kmu = sum;
        kmu1 = sum1 * (2.0 / u);
    } else {
        double a = mu2 - 0.25;
        double b = 2.0 * (u + 1.0), D = 1.0 / b, f = D, delta = D;
        double prev = 0.0, cur = 1.0, C = -a, Q = C, S = 1.0 + Q * delta;
        for (int k = 2; k < 500; ++k) {
            a -= 2 * (k - 1);
            b += 2.0;
            D = 1.0 / (b + a * D);
            delta *= b * D - 1.0;
            f += delta;
            double ra = 1.0 / a;
            double qn = (prev - (b - 2.0) * cur) * ra;
            prev = cur;
            cur = qn;
            C *= -a / k;
            Q += C * qn;
            S += Q * delta;
            if (fabs(Q * delta) < fabs(S) * tol) break;
        }
        kmu = sqrt(pi / (2.0 * u)) * exp(-u) / S;
        kmu1 = kmu * (0.5 + mu + u + (mu2 - 0.25) * f) / u;
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
    rg = rg / prod;
    return exp2(-(nu - 1.0)) * rg * pow(u, nu) * pk;
}

// value of one covariance entry; A = first ("ii") side, B = second ("jj") side.
// gr = global_range = 1/exp(-2*scale[0]); nu_fixed used by the closed-form modes.
template <int MODE>
__device__ __forceinline__ double pair_value(const LocP &A, const LocP &B, double gr, double nu_fixed)
{
    const double epsilon = 2.220446049250313e-16;
    double s11 = (A.rd + B.rd) * 0.5;
    double s22 = kahan(A.rd, A.an2, -B.rd, B.an2) * 0.5;
    double s12 = kahan(A.ra, A.ct, -B.ra, B.ct) * 0.5;
    double det = kahan(s11, s22, s12, s12);
    double dx = A.x - B.x, dy = A.y - B.y;
    double smtns;
    if (MODE == MODE_GEOM) smtns = A.snu * B.snu;
    else if (MODE == MODE_MEAN) smtns = (A.snu + B.snu) / 2;
    else smtns = nu_fixed;
    double u = sqrt(8 * smtns / (gr * det)) *
               sqrt(fma(kahan(s22, dx * dx, -s11, dy * dy), 1.0, -2 * s12 * dx * dy));
    if (u <= epsilon) return A.diag;
    double amp = sqrt(A.ds * B.dets * B.st);
    double sdet = sqrt(det);
    if (MODE == MODE_HALF)
        return exp(-u) * A.sigma * B.sigma * amp / sdet;
    if (MODE == MODE_THREEHALF)
        return (1 + u) * exp(-u) * A.sigma * B.sigma * amp / sdet;
    if (MODE == MODE_FIVEHALF)
        return (1 + u + u * u / 3) * exp(-u) * A.sigma * B.sigma * amp / sdet;
    double m;
    if (u < 706.0) {
        m = matern_bessel(smtns, u);
    } else {   // :301-305
        m = pow(2.0, -(smtns - 1)) / tgamma(smtns) * pow(u, smtns) *
            sqrt(3.14159265358979323846 / (2.0 * u)) * exp(-u);
    }
    return m * A.sigma * B.sigma * amp / sdet;
}

}  // namespace cocons
