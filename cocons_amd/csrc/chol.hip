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
#include <stdlib.h>
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
// 16x16 diagonal block on ONE wave, all in registers (blk layout), MFMA-based:
//   for each 4-column group s: broadcast the 4x4 diagonal sub-block (v_readlane), factor
//   and invert it redundantly on every lane (10 + 10 values), then
//     one MFMA  : columns 4s..4s+3  <-  D(:, group s) * inv(L4)^T        (K = 4)
//     one MFMA  : rank-4 update of the whole 16x16 block
// Outputs: the factor L (blk layout) and Q[s] = per-lane MFMA A-operand of inv(L4_s)
// (row m = lane&15, k = lane>>4; zero outside rows 4s..4s+3), which trsm16() reuses.
__device__ __forceinline__ double rdlane(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// l = sqrt(a), r = 1/sqrt(a) for a normal positive a (pivots of an SPD matrix); both to
// about 1 ulp: v_rsq_f64 seed, two Newton steps on r, one correction of l.
__device__ __forceinline__ void rsqrt_pivot(double a, double &l, double &r)
{
    double y = __builtin_amdgcn_rsq(a);
    double h = 0.5 * a;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    double s = a * y;
    s = fma(fma(-s, s, a), 0.5 * y, s);      // s + (a - s^2) / (2 s)
    y = fma(fma(-s, y, 1.0), y, y);          // y + y (1 - s y)
    l = (a > 0.0) ? s : __builtin_nan("");   // non-positive pivot: propagate NaN like sqrt would
    r = (a > 0.0) ? y : __builtin_nan("");
}

// select among the lower-triangular 4x4 values by (c = row in group, k = column)
__device__ __forceinline__ double sel_lower4(int c, int k, double v00, double v10, double v11, double v20,
                                             double v21, double v22, double v30, double v31, double v32, double v33)
{
    double r0 = v00;                                   // c == 0 (k == 0)
    double r1 = (k == 0) ? v10 : v11;                  // c == 1
    double r2 = (k == 0) ? v20 : ((k == 1) ? v21 : v22);
    double r3 = (k == 0) ? v30 : ((k == 1) ? v31 : ((k == 2) ? v32 : v33));
    double r = (c == 0) ? r0 : ((c == 1) ? r1 : ((c == 2) ? r2 : r3));
    return (k <= c) ? r : 0.0;
}

template <int S>
__device__ __forceinline__ void potrf16_step(d4 &D, double (&Q)[4], int lane, int &fail)
{
    const int m = lane & 15, k = lane >> 4;
    const double ds = D[S];
    // element (4S+a, 4S+b) sits on lane (4S+a) + 16 b of register S
    double a00 = rdlane(ds, 4 * S + 0), a10 = rdlane(ds, 4 * S + 1), a20 = rdlane(ds, 4 * S + 2),
           a30 = rdlane(ds, 4 * S + 3);
    double a11 = rdlane(ds, 4 * S + 1 + 16), a21 = rdlane(ds, 4 * S + 2 + 16), a31 = rdlane(ds, 4 * S + 3 + 16);
    double a22 = rdlane(ds, 4 * S + 2 + 32), a32 = rdlane(ds, 4 * S + 3 + 32);
    double a33 = rdlane(ds, 4 * S + 3 + 48);
    // 4x4 Cholesky (dpotf2 order) -- identical on every lane.  Pivots through
    // rsqrt_pivot(): l = sqrt(a) and r = 1/l from one v_rsq_f64 seed (short dependent chain;
    // this loop is pure latency).
    if (!(a00 > 0.0) && fail == 0) fail = 4 * S + 1;
    double l00, r0;
    rsqrt_pivot(a00, l00, r0);
    double l10 = a10 * r0, l20 = a20 * r0, l30 = a30 * r0;
    double t11 = fma(-l10, l10, a11);
    if (!(t11 > 0.0) && fail == 0) fail = 4 * S + 2;
    double l11, r1;
    rsqrt_pivot(t11, l11, r1);
    double l21 = fma(-l20, l10, a21) * r1, l31 = fma(-l30, l10, a31) * r1;
    double t22 = fma(-l21, l21, fma(-l20, l20, a22));
    if (!(t22 > 0.0) && fail == 0) fail = 4 * S + 3;
    double l22, r2;
    rsqrt_pivot(t22, l22, r2);
    double l32 = fma(-l31, l21, fma(-l30, l20, a32)) * r2;
    double t33 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a33)));
    if (!(t33 > 0.0) && fail == 0) fail = 4 * S + 4;
    double l33, r3;
    rsqrt_pivot(t33, l33, r3);
    // inverse of the 4x4 factor
    double m00 = r0, m11 = r1, m22 = r2, m33 = r3;
    double m10 = -(l10 * m00) * r1;
    double m21 = -(l21 * m11) * r2;
    double m32 = -(l32 * m22) * r3;
    double m20 = -fma(l21, m10, l20 * m00) * r2;
    double m31 = -fma(l32, m21, l31 * m11) * r3;
    double m30 = -fma(l32, m20, fma(l31, m10, l30 * m00)) * r3;
    const int c = m & 3;
    const bool ingrp = (m >> 2) == S;
    double q = sel_lower4(c, k, m00, m10, m11, m20, m21, m22, m30, m31, m32, m33);
    q = ingrp ? q : 0.0;
    Q[S] = q;
    // columns of group S:  X = D(:, group S) * inv(L4)^T ; exact values inside the diagonal sub-block
    d4 z = {0.0, 0.0, 0.0, 0.0};
    d4 X = MFMA64(q, ds, z);
    double lex = sel_lower4(c, k, l00, l10, l11, l20, l21, l22, l30, l31, l32, l33);
    double xs = ingrp ? lex : ((m < 4 * S) ? 0.0 : X[S]);
    // rank-4 update of the remaining columns (registers r > S)
    if (S < 3) {
        d4 U = MFMA64(xs, -xs, D);
#pragma unroll
        for (int r = S + 1; r < 4; ++r) D[r] = U[r];
    }
    D[S] = xs;
}

__device__ __forceinline__ int potrf16_regs(d4 &D, double (&Q)[4], int lane)
{
    int fail = 0;
    potrf16_step<0>(D, Q, lane, fail);
    potrf16_step<1>(D, Q, lane, fail);
    potrf16_step<2>(D, Q, lane, fail);
    potrf16_step<3>(D, Q, lane, fail);
    return fail;
}

// X = B * L^-T for a 16x16 lower block L (blk layout) with the 4x4 inverse operands Q:
// block forward substitution over the four column groups, 7 MFMAs.
__device__ __forceinline__ void trsm16(d4 &B, const d4 &L, const double (&Q)[4])
{
    const d4 z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        d4 T = MFMA64(Q[s], B[s], z);
        B[s] = T[s];
        if (s < 3) {
            d4 U = MFMA64(L[s], -B[s], B);
#pragma unroll
            for (int r = s + 1; r < 4; ++r) B[r] = U[r];
        }
    }
}

// row index of the b-th block of the packed lower triangle (b = ib (ib+1)/2 + kb)
__constant__ int c_tri_ib[36] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5,
                                 6, 6, 6, 6, 6, 6, 6, 7, 7, 7, 7, 7, 7, 7, 7};

// ---------------------------------------------------------------------------
// Diagonal tile: 128x128 Cholesky in LDS (block-packed 16x16 blocks), ONE workgroup of
// 8 waves -- a pure latency kernel on the critical path of the factorisation.
// Per 16-column block jb, two barriers:
//   T: the (7-jb) blocks below the diagonal block are solved with trsm16, one per wave
//   S: symmetric update of the remaining blocks; wave 0 takes block (jb+1,jb+1) first and
//      factors it in registers (potrf16_regs) while waves 1..7 finish the other updates.
// Also exports, per diagonal block, the Q operands (4 x 64 lanes) the panel solve needs.
__global__ void __launch_bounds__(512)
potrf_tile_kernel(double *A, size_t lda, int c0, double *q_out, int *info)
{
    extern __shared__ double smem[];
    // lower-packed image: block (ib,kb), ib >= kb, at (ib (ib+1)/2 + kb) * 256  (72 KB), plus the
    // Q operands of the CURRENT diagonal block (2 KB): 74 KB in all, so the kernel fits on a CU
    // beside four resident update workgroups (look-ahead without reserving CUs)
    double *S = smem;
    double *QS = smem + 36 * 256;
#define SB(ib, kb) (S + ((ib) * ((ib) + 1) / 2 + (kb)) * 256)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    {   // 36 lower blocks, 18 per half-workgroup: every global load is issued before the
        // first LDS store (one round trip instead of 36; matters when the chip is busy)
        const int i = tid & 15, k = (tid >> 4) & 15;
        const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
        const double *src = A + (size_t)(c0 + i) + (size_t)(c0 + k) * lda;
        double v[18];
#pragma unroll
        for (int t = 0; t < 18; ++t) {
            const int bb = 2 * t + half;
            const int ib = c_tri_ib[bb], kb = bb - ib * (ib + 1) / 2;
            v[t] = src[(size_t)(16 * ib) + (size_t)(16 * kb) * lda];
        }
#pragma unroll
        for (int t = 0; t < 18; ++t) {
            const int bb = 2 * t + half;
            const int ib = c_tri_ib[bb], kb = bb - ib * (ib + 1) / 2;
            SB(ib, kb)[k * 16 + i] = v[t];
        }
    }
    __syncthreads();
    if (wave == 0) {
        d4 D = lds_blk(S, lane);
        double Q[4];
        int f = potrf16_regs(D, Q, lane);
        if (f && lane == 0) atomicMin(info, c0 + f);
        lds_blk_store(S, lane, D);
#pragma unroll
        for (int s = 0; s < 4; ++s) { QS[s * 64 + lane] = Q[s]; q_out[s * 64 + lane] = Q[s]; }
    }
    __syncthreads();

    for (int jb = 0; jb < 7; ++jb) {
        double *dblk = SB(jb, jb);
        double *qs = QS;
        // T: one block per wave
        if (jb + 1 + wave < 8) {
            const int ib = jb + 1 + wave;
            d4 L = lds_blk(dblk, lane);
            double Q[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) Q[s] = qs[s * 64 + lane];
            double *blk = SB(ib, jb);
            d4 B = lds_blk(blk, lane);
            trsm16(B, L, Q);
            lds_blk_store(blk, lane, B);
        }
        __syncthreads();
        // S: wave 0 -> next diagonal block, then its factorisation; others share the rest
        if (wave == 0) {
            const int nb = jb + 1;
            d4 P = lds_blk(SB(nb, jb), lane);
            double *blk = SB(nb, nb);
            d4 acc = lds_blk(blk, lane);
            d4 NP = -P;
            blk_mma(acc, NP, P);
            double Q[4];
            int f = potrf16_regs(acc, Q, lane);
            if (f && lane == 0) atomicMin(info, c0 + 16 * nb + f);
            lds_blk_store(blk, lane, acc);
            // the single Q buffer is still being read by the T phase of this jb?  No: T ended at
            // the barrier above; the next reader is the T phase after the barrier below.
#pragma unroll
            for (int s = 0; s < 4; ++s) { QS[s * 64 + lane] = Q[s]; q_out[nb * 256 + s * 64 + lane] = Q[s]; }
        } else {
            int cnt = 0;
            for (int ib = jb + 1; ib < 8; ++ib) {
                for (int kb = jb + 1; kb <= ib; ++kb) {
                    if (ib == jb + 1) continue;            // (jb+1,jb+1) belongs to wave 0
                    if ((cnt++ % 7) + 1 != wave) continue;
                    d4 P = lds_blk(SB(ib, jb), lane);
                    d4 Qk = lds_blk(SB(kb, jb), lane);
                    double *blk = SB(ib, kb);
                    d4 acc = lds_blk(blk, lane);
                    P = -P;
                    blk_mma(acc, P, Qk);
                    lds_blk_store(blk, lane, acc);
                }
            }
        }
        __syncthreads();
    }
    {
        const int i = tid & 15, k = (tid >> 4) & 15, half = tid >> 8;
        double *dst = A + (size_t)(c0 + i) + (size_t)(c0 + k) * lda;
        int b = 0;
        for (int ib = 0; ib < 8; ++ib)
            for (int kb = 0; kb <= ib; ++kb, ++b)
                if ((b & 1) == half)
                    dst[(size_t)(16 * ib) + (size_t)(16 * kb) * lda] = SB(ib, kb)[k * 16 + i];
    }
#undef SB
}

// ---------------------------------------------------------------------------
// Panel solve: rows [r0, r1) of block column c0:  X <- X * L(c0)^-T.
// One workgroup = 64 rows; each wave owns a 16 x 128 strip held in registers (8 blocks).
// LDS holds the 36 lower 16x16 blocks of L and the 8 x 4 Q operands.
__global__ void __launch_bounds__(256)
trsm_tile_kernel(double *A, size_t lda, int c0, int r0, const double *qin)
{
    __shared__ double SL[36 * 256];
    __shared__ double QS[8 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        int i = tid & 15, k = tid >> 4, b = 0;
        for (int ib = 0; ib < 8; ++ib)
            for (int kb = 0; kb <= ib; ++kb, ++b)
                SL[b * 256 + k * 16 + i] = A[(size_t)(c0 + 16 * ib + i) + (size_t)(c0 + 16 * kb + k) * lda];
        for (int e = tid; e < 8 * 256; e += 256) QS[e] = qin[e];
    }
    const int rs = r0 + 64 * blockIdx.x + 16 * wave;
    d4 B[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) B[j] = glb_blk(A, lda, rs, c0 + 16 * j, lane);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        d4 L = lds_blk(SL + (j * (j + 1) / 2 + j) * 256, lane);
        double Q[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) Q[s] = QS[j * 256 + s * 64 + lane];
        trsm16(B[j], L, Q);
        d4 NX = -B[j];
#pragma unroll
        for (int jj = j + 1; jj < 8; ++jj) {
            d4 Lb = lds_blk(SL + (jj * (jj + 1) / 2 + j) * 256, lane);
            blk_mma(B[jj], NX, Lb);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) glb_blk_store(A, lda, rs, c0 + 16 * j, lane, B[j]);
}

// ---------------------------------------------------------------------------
// Panel solve without LDS: the factor's 16x16 blocks and the Q operands are read straight
// from global memory (L2-resident: every workgroup reads the same 72 KB) in blk layout.
// Needs only registers, so its waves can slot in beside resident update workgroups -- this
// is the variant the look-ahead schedule runs on the panel stream.
__global__ void __launch_bounds__(256, 2)
trsm_tile_l2_kernel(double *A, size_t lda, int c0, int r0, const double *qin)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rs = r0 + 64 * blockIdx.x + 16 * wave;
    d4 B[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) B[j] = glb_blk(A, lda, rs, c0 + 16 * j, lane);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        d4 L = glb_blk(A, lda, c0 + 16 * j, c0 + 16 * j, lane);
        // the strictly-upper part of a diagonal block is not stored as zero in global memory
        // only the lower part is meaningful: trsm16 reads L[s] rows >= columns via MFMA, and
        // rows above the diagonal must contribute nothing -> mask them
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((lane & 15) < 4 * r + (lane >> 4)) L[r] = 0.0;
        double Q[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) Q[s] = qin[j * 256 + s * 64 + lane];
        trsm16(B[j], L, Q);
        d4 NX = -B[j];
#pragma unroll
        for (int jj = j + 1; jj < 8; ++jj) {
            d4 Lb = glb_blk(A, lda, c0 + 16 * jj, c0 + 16 * j, lane);
            blk_mma(B[jj], NX, Lb);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) glb_blk_store(A, lda, rs, c0 + 16 * j, lane, B[j]);
}

// ---------------------------------------------------------------------------
// Trailing update: C(ti,tj) -= P(ti,:) P(tj,:)^T over K panel columns.  TM x TM tiles
// (TM = 128: 4 waves x 64x64 = 4x4 MFMA blocks each, the throughput shape; TM = 64:
// 4 waves x 32x32, four times as many workgroups -- used where the grid is too small to
// fill 256 CUs: the narrow in-panel update, the look-ahead update and the late steps).
// Operand tiles stream through LDS in chunks of KC=16 panel columns, register-staged
// double buffering, one barrier per chunk.

struct UpdArgs {
    double *C; size_t ldc;
    const double *P; size_t ldp;   // panel: element (global row, k) at P[row + k*ldp]
    int K;
    int ti0, tj0, lower_only;      // tile indices in units of TM
    int H;                         // lower_only: rows of the trapezoid (ti1 - tj0), 1-D grid over its tiles
    int xcd_swizzle;
    int ptiles, world, rank;       // sharded path: only 128-tile columns whose panel (tj128 / ptiles) is owned
};

// ROLE only names the instantiation (0 = trailing update, 1 = in-panel / sharded / look-ahead
// update) so that profiler summaries keep the dominant trailing launches apart from the narrow ones
template <int TM, int KC, int ROLE>
__global__ void __launch_bounds__(256, (TM == 128 ? 2 : 4))
update_kernel(UpdArgs a)
{
    constexpr int LDT = TM + 16;   // lanes l and l+16 land 128 B apart mod 256 -> conflict-free b64 reads
    constexpr int NB = TM / 32;    // 16x16 blocks per wave and dimension
    constexpr int TPC = 256 / KC;  // threads per panel column
    constexpr int RPT = TM / TPC;  // rows staged per thread and side
    int ti, tj;
    if (a.lower_only) {
        // 1-D grid over the tiles (ti >= tj) of the trapezoid, column by column: column j (0-based
        // from tj0) holds H - j tiles and starts at j H - j (j-1)/2.  No empty workgroups.
        long long L = blockIdx.x;
        if (a.xcd_swizzle) {
            // workgroups b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous
            // run of the tile order so that neighbouring tiles (same operand panels) share an L2
            const long long G = gridDim.x, per = G / 8, rem = G % 8;
            const long long x = L % 8, q = L / 8;
            // XCD x owns [start_x, start_x + per + (x < rem)) of the tile order
            L = x * per + (x < rem ? x : rem) + q;
        }
        const double hh = 2.0 * a.H + 1.0;
        int j = (int)((hh - sqrt(hh * hh - 8.0 * (double)L)) * 0.5);
        while (j > 0 && (long long)j * a.H - (long long)j * (j - 1) / 2 > L) --j;
        while ((long long)(j + 1) * a.H - (long long)(j + 1) * j / 2 <= L) ++j;
        const long long c0 = (long long)j * a.H - (long long)j * (j - 1) / 2;
        tj = a.tj0 + j;
        ti = a.tj0 + j + (int)(L - c0);
    } else {
        ti = a.ti0 + blockIdx.x;
        tj = a.tj0 + blockIdx.y;
    }
    if (a.world > 1 && ((tj * TM / TILE / a.ptiles) % a.world) != a.rank) return;
    __shared__ double sI[2][KC * LDT];
    __shared__ double sJ[2][KC * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;

    // staging map: thread -> (panel column kc, RPT consecutive rows)
    const int kc = tid / TPC, rg = (tid % TPC) * RPT;
    const double *gI = a.P + (size_t)(ti * TM + rg) + (size_t)kc * a.ldp;
    const double *gJ = a.P + (size_t)(tj * TM + rg) + (size_t)kc * a.ldp;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 stI[RPT / 2], stJ[RPT / 2];

    d4 acc[NB][NB];
#pragma unroll
    for (int x = 0; x < NB; ++x)
#pragma unroll
        for (int y = 0; y < NB; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};

    const int nch = a.K / KC;
#pragma unroll
    for (int v = 0; v < RPT / 2; ++v) {
        stI[v] = *(const d2 *)(gI + 2 * v);
        stJ[v] = *(const d2 *)(gJ + 2 * v);
    }
#pragma unroll
    for (int v = 0; v < RPT / 2; ++v) {
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
            for (int v = 0; v < RPT / 2; ++v) {
                stI[v] = *(const d2 *)(pI + 2 * v);
                stJ[v] = *(const d2 *)(pJ + 2 * v);
            }
        }
        const double *bI = &sI[cur][ro + (TM / 2) * wi];
        const double *bJ = &sJ[cur][ro + (TM / 2) * wj];
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            double pi_[NB], pj_[NB];
#pragma unroll
            for (int x = 0; x < NB; ++x) {
                pi_[x] = bI[s * 4 * LDT + 16 * x];
                pj_[x] = bJ[s * 4 * LDT + 16 * x];
            }
#pragma unroll
            for (int x = 0; x < NB; ++x)
#pragma unroll
                for (int y = 0; y < NB; ++y) acc[x][y] = MFMA64(pj_[y], pi_[x], acc[x][y]);
        }
        if (ch + 1 < nch) {
#pragma unroll
            for (int v = 0; v < RPT / 2; ++v) {
                *(d2 *)(&sI[cur ^ 1][kc * LDT + rg + 2 * v]) = stI[v];
                *(d2 *)(&sJ[cur ^ 1][kc * LDT + rg + 2 * v]) = stJ[v];
            }
        }
        __syncthreads();
    }
    // C -= acc
    double *Cb = a.C + (size_t)(ti * TM + (TM / 2) * wi + (lane & 15)) +
                 (size_t)(tj * TM + (TM / 2) * wj + (lane >> 4)) * a.ldc;
#pragma unroll
    for (int x = 0; x < NB; ++x)
#pragma unroll
        for (int y = 0; y < NB; ++y)
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

// Per-row reductions for predict, in two deterministic stages (no floating-point atomics, so the
// kriging outputs are bit-reproducible run to run like the reference's crossprod / rowSums):
// stage 1: partial sums over chunks of `cchunk` columns -> scratch[(chunk * 2 + {0,1}) * m + i]
// stage 2: the chunks of one row summed in ascending order.
__global__ void __launch_bounds__(256)
row_reduce_kernel(const double *A, size_t lda, int n, int rowy, int row0, int m,
                  double *scratch, int cchunk)
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
    scratch[((size_t)blockIdx.y * 2 + 0) * m + i] = s;
    scratch[((size_t)blockIdx.y * 2 + 1) * m + i] = q;
}

__global__ void __launch_bounds__(256)
row_reduce_final_kernel(const double *scratch, int m, int nchunks, double *stoch, double *quad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double s = 0.0, q = 0.0;
    for (int c = 0; c < nchunks; ++c) {
        s += scratch[((size_t)c * 2 + 0) * m + i];
        q += scratch[((size_t)c * 2 + 1) * m + i];
    }
    stoch[i] = s;
    quad[i] = q;
}

// ---------------------------------------------------------------------------
// Y = L E + trend for the lower factor L (marginal simulation: cocoSim's t(iiderrors) %*% cholS,
// R/sim.R:172, is (L E)^T).  One workgroup per 64-row block, lanes along rows (coalesced reads of
// L's columns), the four waves split the k-range and are summed through LDS; E(k, s) is
// wave-uniform.  HBM-bound: the lower triangle of L is read once per group of 8 columns of E.
__global__ void __launch_bounds__(256)
trmm_lower_kernel(const double *A, size_t lda, int n, const double *E, int lde, int nsim,
                  const double *trend, double *Y, int ldy)
{
    __shared__ double red[4][8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = gridDim.x - 1 - blockIdx.x;          // longest rows first
    const int i = rb * 64 + lane;
    const int kend = (rb + 1) * 64 < n ? (rb + 1) * 64 : n;
    for (int s0 = 0; s0 < nsim; s0 += 8) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = wave; k < kend; k += 4) {
            double l = (i < n && k <= i) ? A[(size_t)i + (size_t)k * lda] : 0.0;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s0 + s < nsim) acc[s] = fma(l, E[(size_t)k + (size_t)(s0 + s) * lde], acc[s]);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) red[wave][s][lane] = acc[s];
        __syncthreads();
        if (wave == 0 && i < n) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s0 + s < nsim)
                    Y[(size_t)i + (size_t)(s0 + s) * ldy] =
                        ((red[0][s][lane] + red[1][s][lane]) + (red[2][s][lane] + red[3][s][lane])) + trend[i];
        }
        __syncthreads();
    }
}

void launch_trmm_lower(const double *A, size_t lda, int n, const double *E, int lde, int nsim,
                       const double *trend, double *Y, int ldy, hipStream_t s)
{
    if (n <= 0 || nsim <= 0) return;
    hipLaunchKernelGGL(trmm_lower_kernel, dim3((n + 63) / 64), dim3(256), 0, s, A, lda, n, E, lde, nsim, trend, Y, ldy);
}

// ---------------------------------------------------------------------------
void launch_potrf_tile(double *A, size_t lda, int c0, double *dinv, int *info, hipStream_t s)
{
    const size_t shm = (36 + 1) * 256 * sizeof(double);   // 75,776 B of dynamic LDS (> the 64 KB default)
    // per device and cheap: set on every launch rather than caching a process-wide flag
    (void)hipFuncSetAttribute((const void *)potrf_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(512), shm, s, A, lda, c0, dinv, info);
}

void launch_trsm_tile(double *A, size_t lda, int c0, int r0, int r1, const double *dinv, hipStream_t s,
                      bool no_lds)
{
    int nb = (r1 - r0) / 64;
    if (nb <= 0) return;
    if (no_lds) hipLaunchKernelGGL(trsm_tile_l2_kernel, dim3(nb), dim3(256), 0, s, A, lda, c0, r0, dinv);
    else hipLaunchKernelGGL(trsm_tile_kernel, dim3(nb), dim3(256), 0, s, A, lda, c0, r0, dinv);
}

static int upd64_max_tiles()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("COCONS_UPD64_MAX_TILES");
        v = e ? atoi(e) : 1000000;   // measured: the 64-tile shape wins at every step (45.7 vs 31 TF)
    }
    return v;
}

// K-chunks of 8 (20 KB of LDS, up to 8 workgroups = 8 waves/SIMD per CU) measured 48.1 TFLOP/s
// against 46.8 for chunks of 16 (40 KB, 4 workgroups/CU): default on; COCONS_UPD_KC8=0 to compare
static bool upd_small_lds()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("COCONS_UPD_KC8");
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}

void launch_update_from(double *A, size_t lda, const double *P, size_t ldp, int K,
                        int ti0, int ti1, int tj0, int tj1, bool lower_only, hipStream_t s,
                        int ptiles, int world, int rank)
{
    if (ti1 <= ti0 || tj1 <= tj0 || K <= 0) return;
    UpdArgs a;
    a.C = A; a.ldc = lda; a.P = P; a.ldp = ldp; a.K = K;
    a.lower_only = lower_only ? 1 : 0;
    a.ptiles = ptiles; a.world = world; a.rank = rank;
    // number of 128-tiles that do work
    long nti = ti1 - ti0, ntj = tj1 - tj0;
    long tiles = nti * ntj;
    if (lower_only) {
        tiles = 0;
        for (int tj = tj0; tj < tj1; ++tj) tiles += (ti1 - (tj > ti0 ? tj : ti0));
    }
    if (world > 1) tiles = tiles / world + 1;
    static int swz = -1;
    if (swz < 0) { const char *e = getenv("COCONS_XCD_SWIZZLE"); swz = e ? atoi(e) : 0; }
    a.xcd_swizzle = swz;
    a.H = 0;
    const bool small = tiles <= upd64_max_tiles();
    const int f = small ? 2 : 1;                 // tile indices in units of TM
    a.ti0 = f * ti0; a.tj0 = f * tj0;
    dim3 grid(f * (ti1 - ti0), f * (tj1 - tj0));
    if (lower_only) {
        // requires ti0 >= tj0 == first column: the trapezoid rows tj0..ti1, columns tj0..tj1
        if (ti0 != tj0) { a.lower_only = 0; }    // strictly-below rectangle: every tile does work
        else {
            const long long H = (long long)f * (ti1 - tj0), W = (long long)f * (tj1 - tj0);
            a.H = (int)H;
            grid = dim3((unsigned)(W * H - W * (W - 1) / 2), 1);
        }
    }
    const bool trailing = (K >= 2 * TILE) && world == 1;
    if (small) {
        if (upd_small_lds()) {
            if (trailing) hipLaunchKernelGGL((update_kernel<64, 8, 0>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((update_kernel<64, 8, 1>), grid, dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL((update_kernel<64, 16, 0>), grid, dim3(256), 0, s, a);
        }
    } else {
        hipLaunchKernelGGL((update_kernel<128, 16, 0>), grid, dim3(256), 0, s, a);
    }
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

size_t row_reduce_scratch_doubles(int n, int m)
{
    return (size_t)2 * (size_t)m * (size_t)((n + 255) / 256);
}

void launch_row_reduce(const double *A, size_t lda, int n, int rowy, int row0, int m,
                       double *stoch, double *quad, double *scratch, hipStream_t s)
{
    if (m <= 0) return;
    const int cchunk = 256, nchunks = (n + cchunk - 1) / cchunk;
    hipLaunchKernelGGL(row_reduce_kernel, dim3((m + 255) / 256, nchunks), dim3(256), 0, s,
                       A, lda, n, rowy, row0, m, scratch, cchunk);
    hipLaunchKernelGGL(row_reduce_final_kernel, dim3((m + 255) / 256), dim3(256), 0, s,
                       scratch, m, nchunks, stoch, quad);
}

}  // namespace cocons

// ---------------------------------------------------------------------------
// Measurement probe: back-to-back v_mfma_f64_16x16x4_f64 on every SIMD (one wave per
// SIMD, 4 independent accumulators, operands in registers).  Gives the fp64 matrix rate
// this chip actually sustains, the ceiling the update kernel is priced against.
namespace cocons {
__global__ void __launch_bounds__(256)
mfma_f64_probe_kernel(double *out, int iters, double seed)
{
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) {
        acc0 = MFMA64(a, b, acc0);
        acc1 = MFMA64(b, a, acc1);
        acc2 = MFMA64(a, a, acc2);
        acc3 = MFMA64(b, b, acc3);
    }
    d4 s = acc0 + acc1 + acc2 + acc3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// companion probe: independent v_fma_f64 chains (16 accumulators per lane), the fp64 VECTOR rate
__global__ void __launch_bounds__(256)
vfma_f64_probe_kernel(double *out, int iters, double seed)
{
    double a = seed + threadIdx.x * 1e-9, b = 1.0 - 1e-9;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fma(x[i], b, a);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

double run_vfma_f64_probe(hipStream_t s, int blocks, int iters, double *dbuf)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(vfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, s, dbuf, iters / 10, 1.0);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(vfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, s, dbuf, iters, 1.0);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    double flops = (double)blocks * 256.0 * (double)iters * 16.0 * 2.0;
    return flops / (ms * 1e-3) / 1e12;
}

// both probes at once on two streams: do the matrix and the vector fp64 pipes run concurrently?
// out[0], out[1] = TFLOP/s of the MFMA / FMA kernel while the other one is running
void run_corun_probe(int blocks_mfma, int blocks_vfma, int iters_mfma, int iters_vfma, double *dbuf, double *out)
{
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t a0, a1, b0, b1;
    hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
    double *d0 = dbuf, *d1 = dbuf + (size_t)blocks_mfma * 256;
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks_mfma), dim3(256), 0, s0, d0, 100, 1.0);
    hipLaunchKernelGGL(vfma_f64_probe_kernel, dim3(blocks_vfma), dim3(256), 0, s1, d1, 100, 1.0);
    hipStreamSynchronize(s0); hipStreamSynchronize(s1);
    hipEventRecord(a0, s0);
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks_mfma), dim3(256), 0, s0, d0, iters_mfma, 1.0);
    hipEventRecord(a1, s0);
    hipEventRecord(b0, s1);
    hipLaunchKernelGGL(vfma_f64_probe_kernel, dim3(blocks_vfma), dim3(256), 0, s1, d1, iters_vfma, 1.0);
    hipEventRecord(b1, s1);
    hipStreamSynchronize(s0); hipStreamSynchronize(s1);
    float ma = 0, mb = 0;
    hipEventElapsedTime(&ma, a0, a1);
    hipEventElapsedTime(&mb, b0, b1);
    out[0] = (double)blocks_mfma * 4 * (double)iters_mfma * 4 * 2048.0 / (ma * 1e-3) / 1e12;
    out[1] = (double)blocks_vfma * 256.0 * (double)iters_vfma * 16.0 * 2.0 / (mb * 1e-3) / 1e12;
    out[2] = ma; out[3] = mb;
    hipEventDestroy(a0); hipEventDestroy(a1); hipEventDestroy(b0); hipEventDestroy(b1);
    hipStreamDestroy(s0); hipStreamDestroy(s1);
}

double run_mfma_f64_probe(hipStream_t s, int blocks, int iters, double *dbuf)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, s, dbuf, iters / 10, 1.0);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, s, dbuf, iters, 1.0);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    double flops = (double)blocks * 4 /*waves*/ * (double)iters * 4 /*mfma*/ * 2048.0;
    return flops / (ms * 1e-3) / 1e12;
}
}  // namespace cocons
