// assemble.hip -- covariance assembly kernels (gfx950).
//
//   loc_params_kernel : per-location link functions, replaces the loops at
//                       src/cocons_full.cpp:92-107 / :374-405 / :517-527 and the
//                       inline functions of src/cocons_types.h:12-47.
//   pair_sym_kernel   : n x n symmetric assembly over 64x64 tiles of the lower
//                       triangle (optionally mirrored), replaces the pair loops of
//                       cov_rns (:117-315) and cov_rns_classic (:529-591).
//   pair_rect_kernel  : m x n cross-covariance, replaces cov_rns_pred :407-468.
//
// Layout: per-location SoA, LOCP_FIELDS arrays of length `stride` (coalesced:
// lane = location).  Output is column-major; lanes run along rows so every wave
// store is 512 contiguous bytes.  The kernels are fp64-VALU bound in the general
// (Bessel) modes and HBM-write bound in the closed-form modes.
//
// Compile with -ffp-contract=off (see matern_device.hpp).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "kernels.h"
#include "matern_device.hpp"

namespace cocons {

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
loc_params_kernel(LocArgs a)
{
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.n) return;
    const int p = a.p;
    double t_tilt = 0, t_rd = 0, t_an = 0, t_dets = 0, t_sig = 0, t_ng = 0, t_sm = 0, t_sd = 0;
    for (int i = 0; i < p; ++i) {           // fma chains in column order, types.h:14-16
        double x = a.X[w + (size_t)i * a.ldx];
        t_tilt = fma(x, a.th.tilt[i], t_tilt);
        t_rd = fma(x, a.th.two_scale_je[i], t_rd);
        t_an = fma(x, a.th.aniso[i], t_an);
        t_dets = fma(x, a.th.sqrt_vector[i], t_dets);
        t_sig = fma(x, a.th.half_sd[i], t_sig);
        t_ng = fma(x, a.th.nugget[i], t_ng);
        t_sm = fma(x, a.th.smooth[i], t_sm);
        t_sd = fma(x, a.th.sd[i], t_sd);
    }
    const double pi = 3.14159265358979323846;
    double tilt = pi / (1 + exp(-1 * t_tilt));          // types.h:46
    double rd = 1 / exp(-1 * t_rd);                     // types.h:17
    double an = 1 / exp(-1 * t_an);
    double dets = 1 / exp(-1 * t_dets);
    double sigma = 1 / exp(-1 * t_sig);
    double ng = 1 / exp(-1 * t_ng);
    double snu;
    if (a.smooth_kind == SMOOTH_LOGISTIC_SQRT)          // cocons_full.cpp:93, :381, :401
        snu = sqrt((a.smooth_max - a.smooth_min) / (1 + exp(-1 * t_sm)) + a.smooth_min);
    else if (a.smooth_kind == SMOOTH_EXP)               // :524
        snu = 1 / exp(-1 * t_sm);
    else                                                // fixed-nu branch leaves zeros, :83-88
        snu = 0.0;
    double st = sin(tilt), ct = cos(tilt);
    double *o = a.out;
    size_t s = a.stride;
    o[w + 0 * s] = a.locs[w];
    o[w + 1 * s] = a.locs[w + (size_t)a.ldl];
    o[w + 2 * s] = rd;
    o[w + 3 * s] = an * an;
    o[w + 4 * s] = rd * an;
    o[w + 5 * s] = ct;
    o[w + 6 * s] = st;
    o[w + 7 * s] = dets;
    o[w + 8 * s] = dets * st;
    o[w + 9 * s] = sigma;
    o[w + 10 * s] = snu;
    o[w + 11 * s] = (1 / exp(-1 * t_sd)) + ng;          // :111
    o[w + 12 * s] = ng;
}

constexpr int TS = 64;   // pair tile edge
#ifndef PAIR_MIN_WAVES
#define PAIR_MIN_WAVES 4   // measured on MI355X: 2 -> 3.69 ms, 3 -> 2.93, 4 -> 2.60, 5 -> 2.71, 8 -> 3.65 (n = 10^4 assembly)
#endif

// Symmetric assembly.  One workgroup (256 threads) per 64x64 tile (bi >= bj) of the
// lower triangle, tile columns from a.bj0 up to ncols_out; lane = row, each wave sweeps 16 columns whose parameters are
// wave-uniform (scalar loads).  Entry (r,c), r>c, is evaluated with ii=c, jj=r --
// the reference's (ii<jj) orientation.  npad rows/cols beyond n are written as the
// identity (unit diagonal, zero elsewhere) for the padded factorisation buffer.
template <int MODE, bool MIRROR>
__global__ void __launch_bounds__(256, PAIR_MIN_WAVES)
pair_sym_kernel(PairArgs a)
{
    __shared__ double tile[MIRROR ? TS * (TS + 1) : 1];
    // 1-D grid over the 64x64 tiles (bi >= bj) of the trapezoid rows bj0.., columns bj0..:
    // column j (from bj0) holds H - j tiles and starts at j H - j (j-1)/2 -- no empty workgroups
    int bi, bj;
    {
        const long long L = blockIdx.x;
        const double hh = 2.0 * a.H + 1.0;
        int j = (int)((hh - sqrt(hh * hh - 8.0 * (double)L)) * 0.5);
        while (j > 0 && (long long)j * a.H - (long long)j * (j - 1) / 2 > L) --j;
        while ((long long)(j + 1) * a.H - (long long)(j + 1) * j / 2 <= L) ++j;
        const long long c0 = (long long)j * a.H - (long long)j * (j - 1) / 2;
        bj = a.bj0 + j;
        bi = a.bj0 + j + (int)(L - c0);
    }
    // sharded evaluation: a rank assembles the tile rows of its own 256-row blocks only
    if (a.own_world > 1 && (((bi * TS) / (2 * TILE) / a.own_group) % a.own_world) != a.own_rank) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = a.n;
    const bool offdiag = (bi != bj);           // wave-uniform: then r > c for every entry
    // Lane mapping inside the wave's 64 x 16 share of the tile.  Plain: lane = row, one column per
    // step (column side wave-uniform -> scalar loads).  Blocked (Bessel modes): each step covers an
    // 8 x 8 patch of pairs, so that for spatially ordered inputs the 64 lanes see similar distances
    // and the data-dependent K_nu iteration counts diverge less (the step costs its slowest lane).
    const bool blocked = a.blocked != 0;
    for (int cc = 0; cc < TS / 4; ++cc) {
        int rl, cl;
        if (blocked) {
            rl = 8 * (cc & 7) + (lane & 7);
            cl = wave * (TS / 4) + 8 * (cc >> 3) + (lane >> 3);
        } else {
            rl = lane;
            cl = wave * (TS / 4) + cc;
        }
        const int r = bi * TS + rl;
        const int c = bj * TS + cl;
        double v = 0.0;
        if (c < a.ncols_out) {
            if (r >= n || c >= n) {
                v = (r == c) ? (a.pad_diag != 0.0 ? a.pad_diag : 1.0) : 0.0;
            } else if (r == c) {
                v = a.rows[r + 11 * a.stride];
            } else {
                int rr = r;
                asm volatile("" : "+v"(rr));         // keep the row-side loads inside the loop
                if (offdiag || r > c)                // ii = c, jj = r
                    v = pair_value_idx<MODE>(a.rows, a.stride, c, a.rows, a.stride, rr, a.gr, a.nu_fixed, false);
                else
                    v = pair_value_idx<MODE>(a.rows, a.stride, rr, a.rows, a.stride, c, a.gr, a.nu_fixed, false);
            }
            if (r < a.nrows_out) a.out[(size_t)r + (size_t)c * a.ld] = v;
        }
        if (MIRROR) tile[cl * (TS + 1) + rl] = v;
    }
    if (MIRROR && bi != bj) {
        __syncthreads();
        // transposed write: out[c_global, r_global] for the mirrored tile, coalesced along c
        int c = bj * TS + lane;
        for (int rr = 0; rr < TS / 4; ++rr) {
            int rl = wave * (TS / 4) + rr;
            int rg = bi * TS + rl;
            if (c < a.nrows_out && rg < a.ncols_out)
                a.out[(size_t)c + (size_t)rg * a.ld] = tile[lane * (TS + 1) + rl];
        }
    }
}

// Rectangular cross-covariance: rows = prediction locations (first / "ii" side),
// columns = observation locations.  Exact coordinate match -> prediction-side
// diagonal value (cocons_full.cpp:410-414).
template <int MODE>
__global__ void __launch_bounds__(256, PAIR_MIN_WAVES)
pair_rect_kernel(PairArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m = a.m;
    const bool blocked = a.blocked != 0;           // 8 x 8 pair patches, see pair_sym_kernel
    for (int cc = 0; cc < TS / 4; ++cc) {
        int rl, cl;
        if (blocked) {
            rl = 8 * (cc & 7) + (lane & 7);
            cl = wave * (TS / 4) + 8 * (cc >> 3) + (lane >> 3);
        } else {
            rl = lane;
            cl = wave * (TS / 4) + cc;
        }
        const int r = blockIdx.x * TS + rl;        // prediction location
        const int c = blockIdx.y * TS + cl;
        if (c >= a.ncols_out) continue;
        double v = 0.0;
        if (r < m && c < a.n) {
            int rr = r;
            asm volatile("" : "+v"(rr));
            v = pair_value_idx<MODE>(a.rows, a.stride_rows, rr, a.cols, a.stride, c, a.gr, a.nu_fixed, true);
        }
        if (r < a.nrows_out) a.out[(size_t)r + (size_t)c * a.ld] = v;
    }
}

// rows of right-hand sides under the matrix: out[row0 + k, c] = src[c + k*lds] - trend[c]
// (trend = X %*% mean, R/neg2loglikelihood.R:210,213), zero beyond n.
__global__ void __launch_bounds__(256)
rhs_rows_kernel(RhsArgs a)
{
    int c = a.col0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.ncols_out) return;
    double trend = 0.0;
    if (a.use_trend && c < a.n) {
        // R's %*% is a plain dot product; order i ascending
        for (int i = 0; i < a.p; ++i) trend += a.X[c + (size_t)i * a.ldx] * a.mean[i];
    }
    for (int k = 0; k < a.nrows; ++k) {
        double v = 0.0;
        if (c < a.n) v = a.src[c + (size_t)k * a.lds] - trend;
        a.out[band_index(a.row0 + k, c, a.ld, a.skew, a.npad)] = v;
    }
    for (int k = a.nrows; k < a.nrows + a.nrows_zero; ++k)
        a.out[band_index(a.row0 + k, c, a.ld, a.skew, a.npad)] = 0.0;
}

// Sparse/taper covariance entries (src/cocons_taper.cpp): one thread per stored entry of the CSR pattern
// (colindices / rowpointers 1-based as spam stores them, read-only here -- the reference shifts them
// in place, :73-74, :211-212).  The row of entry w is found by bisection in rowpointers.
struct TaperArgs {
    int nrows, nnz;
    const int *ci, *rp;
    const double *rows; size_t stride_rows;   // SoA of the row side (prediction locations for PRED)
    const double *cols; size_t stride;        // SoA of the column side (observations)
    double nu_fixed;
    double *out;
    // dense target (taper handles): A != nullptr => the entry times its taper value goes straight into the
    // factorisation buffer instead of out[] -- the lower triangle A(ii, jj), jj <= ii, of the symmetric pattern (the
    // upper entries are not even evaluated), or, PRED, row row0 + ii of the rows under the matrix
    const double *tapv; double *A; size_t lda; int row0;
    int skew, npad;          // packed band target (band_index); 0 = dense
};

template <int MODE, bool PRED>
__global__ void __launch_bounds__(256, PAIR_MIN_WAVES)
taper_kernel(TaperArgs a)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.nnz) return;
    int lo = 0, hi = a.nrows - 1;             // largest ii with rp[ii] - 1 <= w
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.rp[mid] - 1 <= w) lo = mid; else hi = mid - 1;
    }
    const int ii = lo, jj = a.ci[w] - 1;
    if (a.A && !PRED && jj > ii) return;
    double v;
    if (!PRED && ii == jj) v = a.rows[ii + 11 * a.stride_rows];
    else v = taper_value_idx<MODE, PRED>(a.rows, a.stride_rows, ii, a.cols, a.stride, jj, a.nu_fixed);
    if (a.A) a.A[band_index((PRED ? a.row0 : 0) + ii, jj, a.lda, a.skew, a.npad)] = a.tapv[w] * v;
    else a.out[w] = v;
}

void launch_taper(int mode, bool pred, int nrows, int nnz, const int *ci, const int *rp, const double *rows,
                  size_t stride_rows, const double *cols, size_t stride, double nu_fixed, double *out, hipStream_t s,
                  const double *tapv, double *A, size_t lda, int row0, int skew, int npad)
{
    if (nnz <= 0) return;
    TaperArgs a;
    a.skew = skew; a.npad = npad;
    a.nrows = nrows; a.nnz = nnz; a.ci = ci; a.rp = rp; a.rows = rows; a.stride_rows = stride_rows;
    a.cols = cols; a.stride = stride; a.nu_fixed = nu_fixed; a.out = out;
    a.tapv = tapv; a.A = A; a.lda = lda; a.row0 = row0;
    dim3 g((nnz + 255) / 256), b(256);
    if (pred) { hipLaunchKernelGGL((taper_kernel<MODE_GEOM, true>), g, b, 0, s, a); return; }
    switch (mode) {
    case MODE_HALF: hipLaunchKernelGGL((taper_kernel<MODE_HALF, false>), g, b, 0, s, a); break;
    case MODE_THREEHALF: hipLaunchKernelGGL((taper_kernel<MODE_THREEHALF, false>), g, b, 0, s, a); break;
    case MODE_FIVEHALF: hipLaunchKernelGGL((taper_kernel<MODE_FIVEHALF, false>), g, b, 0, s, a); break;
    default: hipLaunchKernelGGL((taper_kernel<MODE_GEOM, false>), g, b, 0, s, a); break;
    }
}

// Zero the tiles of a band-limited factorisation buffer: tile column c, tile rows c .. hi[c]-1 (whole 128 x 128 tiles,
// the upper part of the diagonal tile included: the tile factorisation loads full 16 x 16 diagonal blocks).
__global__ void band_zero_kernel(double *A, size_t lda, const int *hi, int nt, int skew)
{
    const int c = blockIdx.x, rt = c + blockIdx.y;
    if (rt >= hi[c]) return;
    double *p = A + (size_t)(skew ? rt - c : rt) * 128 + (size_t)c * 128 * lda;
    for (int e = threadIdx.x; e < 128 * 64; e += blockDim.x) {       // 2 doubles per step
        const int col = e >> 6, r2 = (e & 63) * 2;
        *(double2 *)(p + r2 + (size_t)col * lda) = make_double2(0.0, 0.0);
    }
}

void launch_band_zero(double *A, size_t lda, const int *d_hi, int nt, int max_band, hipStream_t s, int skew)
{
    if (nt <= 0 || max_band <= 0) return;
    hipLaunchKernelGGL(band_zero_kernel, dim3(nt, max_band), dim3(256), 0, s, A, lda, d_hi, nt, skew);
}

// identity on the padding diagonal n .. npad-1 of a taper handle's buffer (its tiles were zeroed)
__global__ void pad_identity_kernel(double *A, size_t lda, int n, int npad, int skew)
{
    const int i = n + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) A[band_index(i, i, lda, skew, npad)] = 1.0;
}

// Columns [0, pad0) of the buffer become unit vectors: 1 on the diagonal, 0 in every row below it, down to row `rows`
// (the rows under the matrix included).  Dense handles keep their identity padding in FRONT of the observations (api.hip,
// fit_create_impl): behind them it would sit in the trailing matrix of every block step -- 3.4 % of the trailing updates'
// arithmetic at n = 10 000 --, in front it is gone after the first panel.
__global__ void __launch_bounds__(256)
front_identity_kernel(double *A, size_t lda, int pad0, int rows)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (r < c || r >= rows || c >= pad0) return;
    A[(size_t)r + (size_t)c * lda] = (r == c) ? 1.0 : 0.0;
}

void launch_front_identity(double *A, size_t lda, int pad0, int rows, hipStream_t s)
{
    if (pad0 > 0 && rows > 0)
        hipLaunchKernelGGL(front_identity_kernel, dim3((rows + 255) / 256, pad0), dim3(256), 0, s, A, lda, pad0, rows);
}

void launch_pad_identity(double *A, size_t lda, int n, int npad, hipStream_t s, int skew)
{
    if (npad > n)
        hipLaunchKernelGGL(pad_identity_kernel, dim3((npad - n + 255) / 256), dim3(256), 0, s, A, lda, n, npad, skew);
}

// Selected rows of the dense covariance (or of cov2cor of it) without ever forming the n x n matrix: what
// plot(type = "correlations") uses of cov_rns (R/methods.R:161-165: tmp_cov[ww, ]).  Entry (i, j) with the
// reference's orientation (ii = the smaller index, src/cocons_full.cpp:119-120); cov2cor as stats::cov2cor:
// (Is[i] * V[i,j]) * Is[j] with Is = 1 / sqrt(diag), diagonal set to exactly 1.
struct RowsArgs {
    int n, nidx;
    const int *idx;
    const double *loc; size_t stride;
    double gr, nu_fixed;
    int cor;
    double *out;            // nidx rows of n, row b at out + b * n
};

template <int MODE>
__global__ void __launch_bounds__(256, PAIR_MIN_WAVES)
cov_rows_kernel(RowsArgs a)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.n) return;
    const int i = a.idx[blockIdx.y];
    double v;
    if (i == j) v = a.loc[i + 11 * a.stride];
    else if (i < j) v = pair_value_idx<MODE>(a.loc, a.stride, i, a.loc, a.stride, j, a.gr, a.nu_fixed, false);
    else v = pair_value_idx<MODE>(a.loc, a.stride, j, a.loc, a.stride, i, a.gr, a.nu_fixed, false);
    if (a.cor) {
        const double isi = 1 / sqrt(a.loc[i + 11 * a.stride]), isj = 1 / sqrt(a.loc[j + 11 * a.stride]);
        v = (i == j) ? 1.0 : (isi * v) * isj;
    }
    a.out[(size_t)blockIdx.y * a.n + j] = v;
}

void launch_cov_rows(int mode, int n, int nidx, const int *idx, const double *loc, size_t stride, double gr,
                     double nu_fixed, int cor, double *out, hipStream_t s)
{
    if (n <= 0 || nidx <= 0) return;
    RowsArgs a;
    a.n = n; a.nidx = nidx; a.idx = idx; a.loc = loc; a.stride = stride; a.gr = gr; a.nu_fixed = nu_fixed;
    a.cor = cor; a.out = out;
    dim3 g((n + 255) / 256, nidx), b(256);
    switch (mode) {
    case MODE_HALF: hipLaunchKernelGGL((cov_rows_kernel<MODE_HALF>), g, b, 0, s, a); break;
    case MODE_THREEHALF: hipLaunchKernelGGL((cov_rows_kernel<MODE_THREEHALF>), g, b, 0, s, a); break;
    case MODE_FIVEHALF: hipLaunchKernelGGL((cov_rows_kernel<MODE_FIVEHALF>), g, b, 0, s, a); break;
    case MODE_MEAN: hipLaunchKernelGGL((cov_rows_kernel<MODE_MEAN>), g, b, 0, s, a); break;
    default: hipLaunchKernelGGL((cov_rows_kernel<MODE_GEOM>), g, b, 0, s, a); break;
    }
}

// Diagnostic: the device Matern/Bessel routine evaluated pointwise, so that it can be pinned
// directly against the mpmath grid (tests/golden/besselk_grid.json) instead of only through Sigma.
__global__ void __launch_bounds__(64)
matern_points_kernel(int n, const double *nu, const double *x, double *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double u = x[i], v = nu[i];
    out[i] = matern_bessel(v, u);
}

void launch_matern_points(int n, const double *nu, const double *x, double *out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(matern_points_kernel, dim3((n + 63) / 64), dim3(64), 0, s, n, nu, x, out);
}

// ---------------------------------------------------------------------------
void launch_loc_params(const LocArgs &a, hipStream_t s)
{
    if (a.n <= 0) return;
    hipLaunchKernelGGL(loc_params_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
}

template <bool MIRROR>
static void launch_sym_mode(int mode, const PairArgs &a, dim3 g, hipStream_t s)
{
    dim3 b(256);
    switch (mode) {
    case MODE_HALF: hipLaunchKernelGGL((pair_sym_kernel<MODE_HALF, MIRROR>), g, b, 0, s, a); break;
    case MODE_THREEHALF: hipLaunchKernelGGL((pair_sym_kernel<MODE_THREEHALF, MIRROR>), g, b, 0, s, a); break;
    case MODE_FIVEHALF: hipLaunchKernelGGL((pair_sym_kernel<MODE_FIVEHALF, MIRROR>), g, b, 0, s, a); break;
    case MODE_MEAN: hipLaunchKernelGGL((pair_sym_kernel<MODE_MEAN, MIRROR>), g, b, 0, s, a); break;
    default: hipLaunchKernelGGL((pair_sym_kernel<MODE_GEOM, MIRROR>), g, b, 0, s, a); break;
    }
}

void launch_pair_sym(int mode, bool mirror, const PairArgs &a, hipStream_t s)
{
    int ext = a.nrows_out > a.ncols_out ? a.nrows_out : a.ncols_out;
    if (ext <= 0) return;
    int T = (ext + TS - 1) / TS;
    int tj1 = (a.ncols_out + TS - 1) / TS;
    if (tj1 <= a.bj0) return;
    PairArgs b = a;
    {
        static int blk = -1;
        if (blk < 0) { const char *e = getenv("COCONS_PAIR_BLOCKED"); blk = e ? atoi(e) : 1; }
        b.blocked = (blk && (mode == MODE_GEOM || mode == MODE_MEAN)) ? 1 : 0;
    }
    const long long H = T - a.bj0, W = tj1 - a.bj0;
    b.H = (int)H;
    dim3 g((unsigned)(W * H - W * (W - 1) / 2));
    if (mirror) launch_sym_mode<true>(mode, b, g, s);
    else launch_sym_mode<false>(mode, b, g, s);
}

void launch_pair_rect(int mode, const PairArgs &a, hipStream_t s)
{
    if (a.nrows_out <= 0 || a.ncols_out <= 0) return;
    dim3 g((a.nrows_out + TS - 1) / TS, (a.ncols_out + TS - 1) / TS), b(256);
    PairArgs pa = a;
    {
        static int blk = -1;
        if (blk < 0) { const char *e = getenv("COCONS_PAIR_BLOCKED"); blk = e ? atoi(e) : 1; }
        pa.blocked = blk ? 1 : 0;
    }
    if (mode == MODE_MEAN) hipLaunchKernelGGL((pair_rect_kernel<MODE_MEAN>), g, b, 0, s, pa);
    else hipLaunchKernelGGL((pair_rect_kernel<MODE_GEOM>), g, b, 0, s, pa);
}

void launch_rhs_rows(const RhsArgs &a, hipStream_t s)
{
    int nc = a.ncols_out - a.col0;
    if (nc <= 0 || a.nrows + a.nrows_zero <= 0) return;
    hipLaunchKernelGGL(rhs_rows_kernel, dim3((nc + 255) / 256), dim3(256), 0, s, a);
}

}  // namespace cocons
