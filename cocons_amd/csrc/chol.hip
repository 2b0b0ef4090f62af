// chol.hip -- blocked right-looking Cholesky of the bordered matrix (gfx950, fp64 MFMA).
//
// Replaces LAPACK dpotrf behind base::chol (R/neg2loglikelihood.R:136,200,259) and the
// dtrsm behind forwardsolve (:214-217): the right-hand sides ride along as extra ROWS
// under the matrix, so L^-1 z falls out of the panel solves and trailing updates.
//
// Storage: column-major, lower triangle, leading dimension lda, everything padded to
// multiples of TILE = 128 (padding rows/cols carry the identity).  We keep
// L = t(chol(Sigma)); the reference's upper factor R satisfies R = L^T, so
// sum(log(diag)) and ||R^-T z|| are identical.
//
// All matrix products run on v_mfma_f64_16x16x4_f64.  A 16x16 block lives in four
// f64 registers per lane in "blk layout":
//      reg r of lane l  <->  element (row = l & 15, col = 4 r + (l >> 4)).
// With the matrix ROW on the lane (contiguous in memory) this layout is at once
//   * the C/D accumulator layout of D[m][n] with n <-> row, m <-> col, and
//   * the A- or B-operand layout for k-step r,
// so a product's result feeds the next product without any data movement:
//      blk_mma(acc, P, Q):  acc(i,j) += sum_k P(i,k) Q(j,k).
#include <hip/hip_runtime.h>
#include <limits.h>
#include "kernels.h"

namespace cocons {

typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void blk_mma(d4 &acc, const d4 &P, const d4 &Q)
{
    acc = MFMA64(Q[0], P[0], acc);
    acc = MFMA64(Q[1], P[1], acc);
    acc = MFMA64(Q[2], P[2], acc);
    acc = MFMA64(Q[3], P[3], acc);
}

// block-packed LDS image: 16x16 blocks of 256 doubles, element (i,k) at k*16 + i
__device__ __forceinline__ d4 lds_blk(const double *blk, int lane)
{
    d4 v;
    int o = (lane >> 4) * 16 + (lane & 15);
    v[0] = blk[o];
    v[1] = blk[o + 64];
    v[2] = blk[o + 128];
    v[3] = blk[o + 192];
    return v;
}

__device__ __forceinline__ void lds_blk_store(double *blk, int lane, const d4 &v)
{
    int o = (lane >> 4) * 16 + (lane & 15);
    blk[o] = v[0];
    blk[o + 64] = v[1];
    blk[o + 128] = v[2];
    blk[o + 192] = v[3];
}

__device__ __forceinline__ d4 glb_blk(const double *A, size_t lda, int row0, int col0, int lane)
{
    const double *p = A + (size_t)(row0 + (lane & 15)) + (size_t)(col0 + (lane >> 4)) * lda;
    d4 v;
    v[0] = p[0];
    v[1] = p[4 * lda];
    v[2] = p[8 * lda];
    v[3] = p[12 * lda];
    return v;
}

__device__ __forceinline__ void glb_blk_store(double *A, size_t lda, int row0, int col0, int lane, const d4 &v)
{
    double *p = A + (size_t)(row0 + (lane & 15)) + (size_t)(col0 + (lane >> 4)) * lda;
    p[0] = v[0];
    p[4 * lda] = v[1];
    p[8 * lda] = v[2];
    p[12 * lda] = v[3];
}

// ---------------------------------------------------------------------------
// 16x16 Cholesky + inverse of the factor, executed by ONE wave (lanes 0..15 hold
// row l of the block; lanes >= 16 mirror lane l&15 so that shuffles stay uniform).
// blk / inv are block-packed LDS images.  Returns the 1-based failing column or 0.
__device__ __forceinline__ int potrf16_wave(double *blk, double *inv, int lane)
{
    const int l = lane & 15;
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = blk[k * 16 + l];
    int fail = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double ajj = __shfl(a[j], j, 64);
        if (!(ajj > 0.0) && fail == 0) fail = j + 1;
        double d = sqrt(ajj);
        double lj = (l == j) ? d : a[j] / d;
        a[j] = lj;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) {
            double lkj = __shfl(lj, k, 64);      // L(k,j)
            a[k] = fma(-lj, lkj, a[k]);          // valid for rows l >= k
        }
    }
    // inverse: lane c owns column c of X = L^-1;  X(i,c) = (delta_ic - sum_{k<i} L(i,k) X(k,c)) / L(i,i)
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double s = (i == l) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) {
            double lik = __shfl(a[k], i, 64);    // L(i,k)
            s = fma(-lik, x[k], s);
        }
        double lii = __shfl(a[i], i, 64);
        x[i] = (i < l) ? 0.0 : s / lii;
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            blk[k * 16 + l] = (k <= l) ? a[k] : 0.0;   // lower factor, zero above the diagonal
            inv[l * 16 + k] = x[k];                     // X(k, c=l) at c*16 + k ... stored as (row k, col l)
        }
    }
    return fail;
}

// ---------------------------------------------------------------------------
// Diagonal tile: 128x128 Cholesky in LDS (block-packed, 64 blocks of 16x16), one
// workgroup of 4 waves.  Steps per 16-column block jb: potrf16 (wave 0) | solve the
// blocks below against inv(L_jj)^T (MFMA) | symmetric update of the rest (MFMA).
__global__ void __launch_bounds__(256)
potrf_tile_kernel(double *A, size_t lda, int c0, double *dinv_out, int *info)
{
    extern __shared__ double smem[];
    double *S = smem;                 // 64 blocks * 256: block (ib,kb) at (ib*8+kb)*256
    double *DI = smem + 64 * 256;     // 8 inverse blocks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // load lower blocks (ib >= kb): thread t of each 256-group moves one element per block
    for (int b = 0; b < 64; ++b) {
        int ib = b >> 3, kb = b & 7;
        if (ib < kb) continue;
        int i = tid & 15, k = tid >> 4;
        S[b * 256 + k * 16 + i] = A[(size_t)(c0 + 16 * ib + i) + (size_t)(c0 + 16 * kb + k) * lda];
    }
    __syncthreads();

    for (int jb = 0; jb < 8; ++jb) {
        if (wave == 0) {
            int f = potrf16_wave(S + (jb * 8 + jb) * 256, DI + jb * 256, lane);
            if (f && lane == 0) atomicMin(info, c0 + 16 * jb + f);
        }
        __syncthreads();
        // X(ib,jb) = A(ib,jb) * inv(L_jj)^T
        d4 Q = lds_blk(DI + jb * 256, lane);
        for (int ib = jb + 1 + wave; ib < 8; ib += 4) {
            double *blk = S + (ib * 8 + jb) * 256;
            d4 P = lds_blk(blk, lane);
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            blk_mma(acc, P, Q);
            lds_blk_store(blk, lane, acc);
        }
        __syncthreads();
        // A(ib,kb) -= X(ib,jb) X(kb,jb)^T for ib >= kb > jb
        int cnt = 0;
        for (int ib = jb + 1; ib < 8; ++ib) {
            for (int kb = jb + 1; kb <= ib; ++kb, ++cnt) {
                if ((cnt & 3) != wave) continue;
                d4 P = lds_blk(S + (ib * 8 + jb) * 256, lane);
                d4 Qk = lds_blk(S + (kb * 8 + jb) * 256, lane);
                double *blk = S + (ib * 8 + kb) * 256;
                d4 acc = lds_blk(blk, lane);
                P = -P;
                blk_mma(acc, P, Qk);
                lds_blk_store(blk, lane, acc);
            }
        }
        __syncthreads();
    }
    for (int b = 0; b < 64; ++b) {
        int ib = b >> 3, kb = b & 7;
        if (ib < kb) continue;
        int i = tid & 15, k = tid >> 4;
        A[(size_t)(c0 + 16 * ib + i) + (size_t)(c0 + 16 * kb + k) * lda] = S[b * 256 + k * 16 + i];
    }
    for (int e = tid; e < 8 * 256; e += 256) dinv_out[e] = DI[e];
}

// ---------------------------------------------------------------------------
// Panel solve: rows [r0, r1) of block column c0:  X <- X * L(c0)^-T.
// One workgroup = 64 rows; each wave owns a 16 x 128 strip held in registers (8 blocks).
// LDS holds the strictly-lower 16x16 blocks of L (28) and the 8 inverse diagonal blocks.
__global__ void __launch_bounds__(256)
trsm_tile_kernel(double *A, size_t lda, int c0, int r0, const double *dinv)
{
    __shared__ double SL[28 * 256];
    __shared__ double DI[8 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        int i = tid & 15, k = tid >> 4, b = 0;
        for (int ib = 1; ib < 8; ++ib)
            for (int kb = 0; kb < ib; ++kb, ++b)
                SL[b * 256 + k * 16 + i] = A[(size_t)(c0 + 16 * ib + i) + (size_t)(c0 + 16 * kb + k) * lda];
        for (int e = tid; e < 8 * 256; e += 256) DI[e] = dinv[e];
    }
    const int rs = r0 + 64 * blockIdx.x + 16 * wave;
    d4 B[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) B[j] = glb_blk(A, lda, rs, c0 + 16 * j, lane);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        d4 X = {0.0, 0.0, 0.0, 0.0};
        d4 Q = lds_blk(DI + j * 256, lane);
        blk_mma(X, B[j], Q);
        B[j] = X;
        d4 NX = -X;
#pragma unroll
        for (int jj = j + 1; jj < 8; ++jj) {
            d4 Lb = lds_blk(SL + (jj * (jj - 1) / 2 + j) * 256, lane);
            blk_mma(B[jj], NX, Lb);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) glb_blk_store(A, lda, rs, c0 + 16 * j, lane, B[j]);
}

// ---------------------------------------------------------------------------
// Trailing update: C(ti,tj) -= P(ti,:) P(tj,:)^T over K panel columns, 128x128 tiles,
// 4 waves x (64x64 = 4x4 MFMA blocks).  Operand tiles stream through LDS in chunks of
// KC=16 panel columns, register-staged double buffering, one barrier per chunk.
constexpr int KC = 16;
constexpr int LDT = 144;    // 128 + 16: lanes l and l+16 land 128 B apart mod 256 -> conflict-free b64 reads

struct UpdArgs {
    double *C; size_t ldc;
    const double *P; size_t ldp;   // panel: element (global row, k) at P[row + k*ldp]
    int K;
    int ti0, tj0, lower_only;
    int ptiles, world, rank;       // sharded path: only tile columns whose panel (tj / ptiles) is owned
};

__global__ void __launch_bounds__(256, 2)
update_kernel(UpdArgs a)
{
    const int ti = a.ti0 + blockIdx.x, tj = a.tj0 + blockIdx.y;
    if (a.lower_only && tj > ti) return;
    if (a.world > 1 && ((tj / a.ptiles) % a.world) != a.rank) return;
    __shared__ double sI[2][KC * LDT];
    __shared__ double sJ[2][KC * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;

    // staging map: thread -> (panel column kc, 8 consecutive rows)
    const int kc = tid >> 4, rg = (tid & 15) * 8;
    const double *gI = a.P + (size_t)(ti * TILE + rg) + (size_t)kc * a.ldp;
    const double *gJ = a.P + (size_t)(tj * TILE + rg) + (size_t)kc * a.ldp;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 stI[4], stJ[4];

    d4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};

    const int nch = a.K / KC;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        stI[v] = *(const d2 *)(gI + 2 * v);
        stJ[v] = *(const d2 *)(gJ + 2 * v);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        *(d2 *)(&sI[0][kc * LDT + rg + 2 * v]) = stI[v];
        *(d2 *)(&sJ[0][kc * LDT + rg + 2 * v]) = stJ[v];
    }
    __syncthreads();

    const int ro = (lane >> 4) * LDT + (lane & 15);
    for (int ch = 0; ch < nch; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nch) {
            const double *pI = gI + (size_t)(ch + 1) * KC * a.ldp;
            const double *pJ = gJ + (size_t)(ch + 1) * KC * a.ldp;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                stI[v] = *(const d2 *)(pI + 2 * v);
                stJ[v] = *(const d2 *)(pJ + 2 * v);
            }
        }
        const double *bI = &sI[cur][ro + 64 * wi];
        const double *bJ = &sJ[cur][ro + 64 * wj];
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            double pi_[4], pj_[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                pi_[x] = bI[s * 4 * LDT + 16 * x];
                pj_[x] = bJ[s * 4 * LDT + 16 * x];
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = MFMA64(pj_[y], pi_[x], acc[x][y]);
        }
        if (ch + 1 < nch) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                *(d2 *)(&sI[cur ^ 1][kc * LDT + rg + 2 * v]) = stI[v];
                *(d2 *)(&sJ[cur ^ 1][kc * LDT + rg + 2 * v]) = stJ[v];
            }
        }
        __syncthreads();
    }
    // C -= acc
    double *Cb = a.C + (size_t)(ti * TILE + 64 * wi + (lane & 15)) +
                 (size_t)(tj * TILE + 64 * wj + (lane >> 4)) * a.ldc;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double *p = Cb + 16 * x + (size_t)(16 * y + 4 * r) * a.ldc;
                *p -= acc[x][y][r];
            }
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    __syncthreads();
    return t;    // valid on thread 0
}

// block b < nr*nr: Gram entry (b / nr, b % nr) over columns [c0,c1) and < n;
// block nr*nr: sum of log of the diagonal over the same columns.
__global__ void __launch_bounds__(256)
finalize_kernel(const double *A, size_t lda, int c0, int c1, int n, int row0, int nr, double *out)
{
    __shared__ double red[4];
    const int b = blockIdx.x;
    const int hi = c1 < n ? c1 : n;
    double s = 0.0;
    if (b == nr * nr) {
        for (int c = c0 + threadIdx.x; c < hi; c += blockDim.x) s += log(A[(size_t)c + (size_t)c * lda]);
    } else {
        const int ra = row0 + b / nr, rb = row0 + b % nr;
        for (int c = c0 + threadIdx.x; c < hi; c += blockDim.x)
            s += A[(size_t)ra + (size_t)c * lda] * A[(size_t)rb + (size_t)c * lda];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[b == nr * nr ? 0 : 1 + b] = s;
}

__global__ void __launch_bounds__(256)
row_reduce_kernel(const double *A, size_t lda, int n, int rowy, int row0, int m,
                  double *stoch, double *quad, int cchunk)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int cb = blockIdx.y * cchunk;
    const int ce = (cb + cchunk < n) ? cb + cchunk : n;
    if (i >= m) return;
    double s = 0.0, q = 0.0;
    for (int c = cb; c < ce; ++c) {
        double v = A[(size_t)(row0 + i) + (size_t)c * lda];
        double y = A[(size_t)rowy + (size_t)c * lda];
        s = fma(v, y, s);
        q = fma(v, v, q);
    }
    atomicAdd(&stoch[i], s);
    atomicAdd(&quad[i], q);
}

// ---------------------------------------------------------------------------
void launch_potrf_tile(double *A, size_t lda, int c0, double *dinv, int *info, hipStream_t s)
{
    static bool attr_set = false;
    const size_t shm = (64 + 8) * 256 * sizeof(double);   // 147,456 B
    if (!attr_set) {
        hipFuncSetAttribute((const void *)potrf_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set = true;
    }
    hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(256), shm, s, A, lda, c0, dinv, info);
}

void launch_trsm_tile(double *A, size_t lda, int c0, int r0, int r1, const double *dinv, hipStream_t s)
{
    int nb = (r1 - r0) / 64;
    if (nb <= 0) return;
    hipLaunchKernelGGL(trsm_tile_kernel, dim3(nb), dim3(256), 0, s, A, lda, c0, r0, dinv);
}

void launch_update_from(double *A, size_t lda, const double *P, size_t ldp, int K,
                        int ti0, int ti1, int tj0, int tj1, bool lower_only, hipStream_t s,
                        int ptiles, int world, int rank)
{
    if (ti1 <= ti0 || tj1 <= tj0 || K <= 0) return;
    UpdArgs a;
    a.C = A; a.ldc = lda; a.P = P; a.ldp = ldp; a.K = K;
    a.ti0 = ti0; a.tj0 = tj0; a.lower_only = lower_only ? 1 : 0;
    a.ptiles = ptiles; a.world = world; a.rank = rank;
    hipLaunchKernelGGL(update_kernel, dim3(ti1 - ti0, tj1 - tj0), dim3(256), 0, s, a);
}

void launch_update(double *A, size_t lda, int k0, int K, int ti0, int ti1, int tj0, int tj1,
                   bool lower_only, hipStream_t s)
{
    launch_update_from(A, lda, A + (size_t)k0 * lda, lda, K, ti0, ti1, tj0, tj1, lower_only, s, 1, 1, 0);
}

void launch_finalize_cols(const double *A, size_t lda, int c0, int c1, int n, int row0, int nr,
                          double *out, hipStream_t s)
{
    hipLaunchKernelGGL(finalize_kernel, dim3(nr * nr + 1), dim3(256), 0, s, A, lda, c0, c1, n, row0, nr, out);
}

void launch_finalize(const double *A, size_t lda, int n, int row0, int nr, double *out, hipStream_t s)
{
    launch_finalize_cols(A, lda, 0, n, n, row0, nr, out, s);
}

void launch_row_reduce(const double *A, size_t lda, int n, int rowy, int row0, int m,
                       double *stoch, double *quad, hipStream_t s)
{
    if (m <= 0) return;
    const int cchunk = 256;
    hipMemsetAsync(stoch, 0, (size_t)m * sizeof(double), s);
    hipMemsetAsync(quad, 0, (size_t)m * sizeof(double), s);
    hipLaunchKernelGGL(row_reduce_kernel, dim3((m + 255) / 256, (n + cchunk - 1) / cchunk), dim3(256), 0, s,
                       A, lda, n, rowy, row0, m, stoch, quad, cchunk);
}

}  // namespace cocons
