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
#include <string.h>
#include <algorithm>
#include <atomic>
#include <vector>
#include "kernels.h"

namespace cocons {

typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
// the same with the first operand negated: for the fp64 forms the instruction's blgp field holds NEG bits
// (neg:[a,b,c]; bit 0 = first source), so D = C - A B costs no extra instruction
#define MFMA64_NEGA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 1)

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

// Write-through (sc1) stores for data handed to another workgroup INSIDE a launch: the bytes leave the
// XCD's L2 at once, so publishing them needs no release fence (buffer_wbl2 would write back every dirty
// line of that L2 -- with a trailing update in flight that is megabytes, and it measurably slowed the
// update chip-wide).  One 8-byte agent-scope relaxed atomic store per element = global_store_dwordx2 sc1.
__device__ __forceinline__ void store_wt(double *p, double v)
{
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// L1-bypassing (sc1) load of data another workgroup published with store_wt: with EVERY load of the
// handed-off bytes of this form the consumer needs no acquire fence (buffer_inv)
__device__ __forceinline__ double load_wt(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ d4 glb_blk_wt(const double *A, size_t lda, int row0, int col0, int lane)
{
    const double *p = A + (size_t)(row0 + (lane & 15)) + (size_t)(col0 + (lane >> 4)) * lda;
    d4 v;
    v[0] = load_wt(p);
    v[1] = load_wt(p + 4 * lda);
    v[2] = load_wt(p + 8 * lda);
    v[3] = load_wt(p + 12 * lda);
    return v;
}

__device__ __forceinline__ void glb_blk_store_wt(double *A, size_t lda, int row0, int col0, int lane, const d4 &v)
{
    double *p = A + (size_t)(row0 + (lane & 15)) + (size_t)(col0 + (lane >> 4)) * lda;
    store_wt(p, v[0]);
    store_wt(p + 4 * lda, v[1]);
    store_wt(p + 8 * lda, v[2]);
    store_wt(p + 12 * lda, v[3]);
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

// l = sqrt(a), r = 1/sqrt(a) for a normal positive a (pivots of an SPD matrix); both to about 1 ulp.  v_rsq_f64 delivers
// 24 bits (measured: tools/diag/seed_precision.hip, max relative error 2^-24.2), so ONE third-order step
//     y = y0 (1 + t/2 + 3 t^2/8),   t = 1 - a y0^2        (truncation 5/16 t^3 < 2^-70)
// gives r after four dependent operations -- r is what the pivot chain of potrf16_step waits for -- and l = a y with one
// Heron correction follows off the chain.  (Rounds 1-2 ran two Newton steps and two corrections: r came out last, after
// eleven dependent operations, sixteen times per 16 x 16 block on the one wave every diagonal tile waits for.)
__device__ __forceinline__ void rsqrt_pivot(double a, double &l, double &r)
{
    const double y0 = __builtin_amdgcn_rsq(a);
    const double s0 = a * y0;
    const double t = fma(-s0, y0, 1.0);
    const double y = fma(y0 * t, fma(t, 0.375, 0.5), y0);
    double s = a * y;
    s = fma(fma(-s, s, a), 0.5 * y, s);      // s + (a - s^2) / (2 s)
    // A pivot that is not a positive finite number comes out as NaN in both results WITHOUT being asked (like sqrt would):
    // v_rsq_f64 gives NaN for a < 0, +-inf for +-0 (then a y0 = 0 inf = NaN) and 0 for +inf (inf 0 = NaN).  Rounds 1-3
    // selected NaN explicitly: eight v_cndmask per pivot on the one wave every diagonal tile waits for.
    l = s;
    r = y;
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
    double l00, r0;
    rsqrt_pivot(a00, l00, r0);
    double l10 = a10 * r0, l20 = a20 * r0, l30 = a30 * r0;
    double t11 = fma(-l10, l10, a11);
    double l11, r1;
    rsqrt_pivot(t11, l11, r1);
    double l21 = fma(-l20, l10, a21) * r1, l31 = fma(-l30, l10, a31) * r1;
    double t22 = fma(-l21, l21, fma(-l20, l20, a22));
    double l22, r2;
    rsqrt_pivot(t22, l22, r2);
    double l32 = fma(-l31, l21, fma(-l30, l20, a32)) * r2;
    double t33 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a33)));
    double l33, r3;
    rsqrt_pivot(t33, l33, r3);
    // which pivot failed first is asked ONCE per group, behind the chain, and only looked into when the last pivot is not
    // a positive number -- a bad pivot makes every later one NaN (the values are the same on every lane: a scalar branch)
    if (__builtin_amdgcn_ballot_w64(!(t33 > 0.0)) != 0ull && fail == 0)
        fail = 4 * S + (!(a00 > 0.0) ? 1 : (!(t11 > 0.0) ? 2 : (!(t22 > 0.0) ? 3 : 4)));
    // (An outer-product form with reciprocals on the dependent chain and the square roots refined beside it -- 3 x 6 + 12
    // dependent operations instead of 4 x 14 -- was measured in round 3: SLOWER, 4.54 -> 4.72 ms on the taper path, whose
    // time is half tile factorisations: one wave issues in order, and the variant has a quarter more instructions.)
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
    // columns of group S:  X = D(:, group S) * inv(L4)^T -- the rows of the diagonal sub-block too (D4 inv(L4)^T = L4: the
    // block arrives SYMMETRIC, potrf_tile_body mirrors the diagonal blocks when it loads the tile and every update keeps
    // them so), with exact zeros above the diagonal.  (Until round 4 those sixteen entries were selected from the scalar
    // factor: a second ten-way select, twenty v_cndmask per group on the wave every diagonal tile waits for.)
    d4 z = {0.0, 0.0, 0.0, 0.0};
    d4 X = MFMA64(q, ds, z);
    (void)l11; (void)l22; (void)l33; (void)l00;
    double xs = (m < 4 * S + k) ? 0.0 : X[S];
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
// qall (LDS, may be null): receives the Q operands of all eight diagonal blocks (8 x 256 doubles)
// WT: the factor and the Q operands are published with write-through stores (engine)
// INV (engine of the dependency-driven schedule, round 5): W = L^-1 of the tile is formed ALONGSIDE the factorisation and
// written (write-through, column-major, leading dimension 128, exact zeros above the diagonal) to winv.  W^T = I L^-T is the
// strip solve applied to the identity: strip w (rows 16 w ..) is X(w, j), j >= w, with
//     X(w, j) = ( I(w, j) - sum_{w <= k < j} X(w, k) L(j, k)^T ) L(j, j)^-T,
// and column block j of it needs nothing but row block j of the factor, which is final once the diagonal block (j, j) is
// factored -- so X(., j) is formed in the S phase of step j by the waves 1 .. 7 (strip w on wave w + 1), in the shadow of wave 0's
// register factorisation of the next diagonal block (1.9 us during which they had ~1 us of work), and only the last column
// block is left for a phase of its own behind the loop.  Until round 4 the inverse was a pass of its own behind the
// factorisation (engine_tile_inverse: 9.6 us per tile ON the chain from one diagonal block to the next, twice per block; the
// engine's stamps, tools/chain_trace.py).  The strips X(w, j), j <= 6, stay in LDS (28 blocks at xi) for the later columns.
#ifndef POTRF_S_WAVES
#define POTRF_S_WAVES 7
#endif
// FROM_LDS (the pair partner of the engine, round 5): the tile's image is in LDS already -- lower blocks in place, diagonal
// blocks mirrored, a barrier passed -- and nothing is fetched.
// mbox (may be null): MAILBOX of the tile for a consumer that FOLLOWS the factorisation column block by column block (the pair
// partner): 44 blocks of 256 doubles, column block j -- blocks (j .. 7, j), then its Q operands -- contiguous at MBOX_OFF(j), in
// the LDS block layout; the caller has filled it with the bit pattern ~0 (a NaN no arithmetic produces), and every finished block
// is stored there too, the moment it is stored to the matrix.  The consumer reads the mailbox until no word of a column block
// is the fill pattern any more: the data is its own flag (8-byte write-through stores are seen whole), which costs ONE round
// trip per column block where a drained flag behind the stores cost the drain, the flag's trip and then the data's.
// late (may be null): a word to be raised (+1) for stores the CALLER issued before the call -- drained in the first S phase (every
// wave, behind ~1 us of other work), so that the caller need not wait for them (the pair partner's copy of X: 64 eight-byte
// write-through stores per lane, which nothing on the chain needs).
#define MBOX_OFF(j) (256 * (9 * (j) - (j) * ((j) - 1) / 2))
__device__ __forceinline__ void mbox_store(double *mb, int lane, const d4 &v)
{
    store_wt(mb + lane, v[0]);
    store_wt(mb + lane + 64, v[1]);
    store_wt(mb + lane + 128, v[2]);
    store_wt(mb + lane + 192, v[3]);
}
template <bool WT, bool FROM_LDS = false>
__device__ __forceinline__ void potrf_tile_body(double *A, size_t lda, int c0, double *q_out, int *info, double *smem,
                                                double *qall, double *winv = nullptr, double *xi = nullptr, double *mbox = nullptr,
                                                unsigned *late = nullptr)
{
    const bool INV = winv != nullptr;          // (wave-uniform; needs qall and xi)
    // lower-packed image: block (ib,kb), ib >= kb, at (ib (ib+1)/2 + kb) * 256  (72 KB), plus the
    // Q operands of the CURRENT diagonal block (2 KB): 74 KB in all
    double *S = smem;
    double *QS = smem + 36 * 256;
#define SB(ib, kb) (S + ((ib) * ((ib) + 1) / 2 + (kb)) * 256)
    // (the thread index is taken afresh: the engine kernel holds several inlined copies of this body, and what the compiler
    // derives from the index -- LDS addresses, the identity's lanes -- would otherwise be formed once at the kernel's entry
    // for all of them and live, or spill, through everything else)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // (round 5) Wave 0 fetches the first diagonal block straight into registers -- blk layout, mirrored like below -- and
    // factors it WHILE the cooperative fetch of the tile is in flight (its own loads are issued first and return first); and
    // every finished block of the factor goes to memory as soon as it is final, from the registers of the wave that formed
    // it, instead of in a pass of its own at the end: the tile's fetch and store (4.5 of its 26 us) now run beside the first
    // and behind the last register factorisation.
    d4 D0 = {0.0, 0.0, 0.0, 0.0};
    double Q0[4] = {0.0, 0.0, 0.0, 0.0};
    if (FROM_LDS) {
        if (wave == 0) {
            D0 = lds_blk(S, lane);
            int f = potrf16_regs(D0, Q0, lane);
            if (f && lane == 0) atomicMin(info, c0 + f);
        }
    } else {
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = lane & 15, c = 4 * r + (lane >> 4);
            const double *sp = i < c ? A + (size_t)(c0 + c) + (size_t)(c0 + i) * lda : A + (size_t)(c0 + i) + (size_t)(c0 + c) * lda;
            D0[r] = WT ? load_wt(sp) : *sp;
        }
    }
    {   // 36 lower blocks, 18 per half-workgroup: every global load is issued before the
        // first LDS store (one round trip instead of 36; matters when the chip is busy)
        const int i = tid & 15, k = (tid >> 4) & 15;
        const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
        const double *src = A + (size_t)(c0 + i) + (size_t)(c0 + k) * lda;
        // (the eight DIAGONAL blocks are loaded symmetric -- their upper half mirrored from the lower, the only half of the
        // matrix that holds data: potrf16_step takes the factor's diagonal sub-blocks out of a product that reads both)
        const double *srcd = (i < k) ? A + (size_t)(c0 + k) + (size_t)(c0 + i) * lda : src;
        double v[18];
#pragma unroll
        for (int t = 0; t < 18; ++t) {
            const int bb = 2 * t + half;
            const int ib = c_tri_ib[bb], kb = bb - ib * (ib + 1) / 2;
            const double *sp = (ib == kb ? srcd : src) + (size_t)(16 * ib) + (size_t)(16 * kb) * lda;
            v[t] = WT ? load_wt(sp) : *sp;
        }
        if (wave == 0) {         // ... while those are in flight
            int f = potrf16_regs(D0, Q0, lane);
            if (f && lane == 0) atomicMin(info, c0 + f);
        }
#pragma unroll
        for (int t = 0; t < 18; ++t) {
            const int bb = 2 * t + half;
            const int ib = c_tri_ib[bb], kb = bb - ib * (ib + 1) / 2;
            SB(ib, kb)[k * 16 + i] = v[t];
        }
    }
    }
    __syncthreads();
    if (wave == 0) {             // (over the unfactored copy the cooperative stores left there)
        lds_blk_store(S, lane, D0);
        if (WT) glb_blk_store_wt(A, lda, c0, c0, lane, D0); else glb_blk_store(A, lda, c0, c0, lane, D0);
        if (mbox) mbox_store(mbox, lane, D0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            QS[s * 64 + lane] = Q0[s];
            if (WT) store_wt(q_out + s * 64 + lane, Q0[s]); else q_out[s * 64 + lane] = Q0[s];
            if (qall) qall[s * 64 + lane] = Q0[s];
            if (mbox) store_wt(mbox + 8 * 256 + s * 64 + lane, Q0[s]);
        }
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
            if (WT) glb_blk_store_wt(A, lda, c0 + 16 * ib, c0 + 16 * jb, lane, B);      // final: to memory now
            else glb_blk_store(A, lda, c0 + 16 * ib, c0 + 16 * jb, lane, B);
            if (mbox) mbox_store(mbox + MBOX_OFF(jb) + (ib - jb) * 256, lane, B);
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
            if (late && jb == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the caller's stores: issued > 2 us ago)
            lds_blk_store(blk, lane, acc);
            if (WT) glb_blk_store_wt(A, lda, c0 + 16 * nb, c0 + 16 * nb, lane, acc);      // final: to memory now
            else glb_blk_store(A, lda, c0 + 16 * nb, c0 + 16 * nb, lane, acc);
            if (mbox) mbox_store(mbox + MBOX_OFF(nb), lane, acc);
            // the single Q buffer is still being read by the T phase of this jb?  No: T ended at
            // the barrier above; the next reader is the T phase after the barrier below.
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                QS[s * 64 + lane] = Q[s];
                if (WT) store_wt(q_out + nb * 256 + s * 64 + lane, Q[s]); else q_out[nb * 256 + s * 64 + lane] = Q[s];
                if (qall) qall[nb * 256 + s * 64 + lane] = Q[s];
                if (mbox) store_wt(mbox + MBOX_OFF(nb) + (8 - nb) * 256 + s * 64 + lane, Q[s]);
            }
        } else {
            int cnt = 0;
            for (int ib = jb + 1; ib < 8; ++ib) {
                for (int kb = jb + 1; kb <= ib; ++kb) {
                    if (ib == jb + 1) continue;            // (jb+1,jb+1) belongs to wave 0
#if POTRF_S_WAVES == 6
                    { const int ix = cnt++ % 6; if ((ix < 3 ? ix + 1 : ix + 2) != wave) continue; }   // (not wave 4: wave 0's SIMD)
#else
                    if ((cnt++ % 7) + 1 != wave) continue;
#endif
                    d4 P = lds_blk(SB(ib, jb), lane);
                    d4 Qk = lds_blk(SB(kb, jb), lane);
                    double *blk = SB(ib, kb);
                    d4 acc = lds_blk(blk, lane);
                    P = -P;
                    blk_mma(acc, P, Qk);
                    lds_blk_store(blk, lane, acc);
                }
            }
            if (late && jb == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the caller's stores, see the header)
            // column block j = jb of the inverse's strips (see the header comment).  Strip w lives on wave INV_WAVE[w]: NOT on
            // wave 4, which shares its SIMD -- and the SIMD's double-precision unit -- with wave 0, whose pivot chain is what the
            // whole tile waits for (fp64 vector chains run up to three times slower beside fp64 MFMAs: DESIGN.md section 4a);
            // strip 6, a single step long, rides with strip 5.  L(j, j) and its Q operands were published by wave 0 in the
            // previous S phase (qall: written once per block -- the single QS buffer is being overwritten by wave 0 right now)
            for (int w = (wave == 7 ? 5 : (wave > 4 ? wave - 2 : wave - 1)); INV && wave != 4 && w <= jb && w <= 6;
                 w = (wave == 7 && w == 5) ? 6 : 8) {
                const int j = jb;
                d4 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = (w == j && (lane & 15) == 4 * r + (lane >> 4)) ? 1.0 : 0.0;
                for (int k = w; k < j; ++k) {
                    d4 Xk = lds_blk(xi + (k * (k + 1) / 2 + w) * 256, lane);
                    d4 Lb = lds_blk(SB(j, k), lane);
                    Xk = -Xk;
                    blk_mma(acc, Xk, Lb);
                }
                d4 Ld = lds_blk(SB(j, j), lane);
                double Qj[4];
#pragma unroll
                for (int sq = 0; sq < 4; ++sq) Qj[sq] = qall[j * 256 + sq * 64 + lane];
                trsm16(acc, Ld, Qj);
                lds_blk_store(xi + (j * (j + 1) / 2 + w) * 256, lane, acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int wr = 16 * j + 4 * r + (lane >> 4), wc = 16 * w + (lane & 15);      // W(wr, wc) = X(w, j)^T
                    store_wt(winv + wr + (size_t)wc * TILE, wr >= wc ? acc[r] : 0.0);
                }
            }
        }
        __syncthreads();
        if (late && jb == 0 && tid == 448) __hip_atomic_fetch_add(late, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (INV) {
        // the last column block (j = 7): its diagonal block was factored in the loop's last S phase; one strip per wave
        const int w = wave, j = 7;
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (w == j && (lane & 15) == 4 * r + (lane >> 4)) ? 1.0 : 0.0;
        for (int k = w; k < j; ++k) {
            d4 Xk = lds_blk(xi + (k * (k + 1) / 2 + w) * 256, lane);
            d4 Lb = lds_blk(SB(j, k), lane);
            Xk = -Xk;
            blk_mma(acc, Xk, Lb);
        }
        d4 Ld = lds_blk(SB(j, j), lane);
        double Qj[4];
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) Qj[sq] = qall[j * 256 + sq * 64 + lane];
        trsm16(acc, Ld, Qj);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int wr = 16 * j + 4 * r + (lane >> 4), wc = 16 * w + (lane & 15);
            store_wt(winv + wr + (size_t)wc * TILE, wr >= wc ? acc[r] : 0.0);
        }
    }
    // (no store pass: every block went to memory when it became final)
#undef SB
}

__global__ void __launch_bounds__(512)
potrf_tile_kernel(double *A, size_t lda, int c0, double *q_out, int *info)
{
    extern __shared__ double smem[];
    potrf_tile_body<false>(A, lda, c0, q_out, info, smem, nullptr);
}

// ---------------------------------------------------------------------------
// Hand-offs between workgroups of DIFFERENT kernels that are resident at the same time (the
// diagonal-tile engine below and the update / panel-solve kernels of the main stream).  Protocol
// (agent scope; per-XCD L2s are not coherent and a CU's L1 is never refreshed by other CUs):
//   producer: payload stored WRITE-THROUGH (sc1: store_wt), every storing wave drains its stores
//             (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane: relaxed agent-scope atomic on the flag word
//             (no release fence: see store_wt)
//   consumer: ONE lane polls the flag relaxed (s_sleep between polls), then ONE acquire fence (L1
//             invalidate), drain, workgroup barrier, then plain loads.
// Every spin is bounded: after ENGINE_TIMEOUT_TICKS of the 100 MHz constant clock the waiter sets the
// abort word and every party leaves; the host then repeats the factorisation on the plain schedule
// instead of hanging the GPU.  A legitimate wait lasts at most one trailing-update launch (< 1 ms).
#ifndef ENGINE_TIMEOUT_TICKS
#define ENGINE_TIMEOUT_TICKS 10000000ull    // 100 ms
#endif
#define GATE_TIMEOUT_TICKS 500000ull        // 5 ms: the engine is resident within microseconds or -- every CU taken by
                                            // someone else's kernels -- not for a long while
// What a wait is bounded by is the waiter's OWN spinning, not the wall clock: a poll costs >= ~0.6 us (s_sleep 16 = 0.43 us plus
// the load's round trip), so `ticks` of the 100 MHz clock are converted into ticks / 100 polls.  Why: the driver now and
// then pauses every queue of the process in mid-kernel -- all waves saved, restored about 0.9 ms later on other CUs (round 4:
// seen in the task stamps of the DAG launch about once per 4000 evaluations on this pool; dag_kernel's header has the rest
// of that story).  A waiter that stands still does not poll, so a pause of whatever length never turns into a time-out; a
// true deadlock still ends after ticks / 100 polls, and WALL_BACKSTOP_TICKS ends anything else.
#define WALL_BACKSTOP_TICKS 500000000ull    // 5 s
// The engine's waits for its INPUT words are paced by the HOST: the launch that raises in[t] may not have been enqueued yet
// when the engine -- resident since before the factorisation began -- asks for it.  A host thread that loses the CPU in the
// middle of enqueueing an evaluation (round 5: the first evaluation behind a LAPACK call on every core of a box whose
// process group is CPU-limited; abort 0x112 = the engine waiting for tile 18 at n = 4096, once in two runs of the test suite,
// never in 6000 evaluations of a loop that does nothing else) must not look like a lost partner: these waits get 3 s of
// polling.  Every genuine circular wait contains a waiter on the OTHER side -- a main-stream kernel waiting for out[] / xr[],
// bounded by ENGINE_TIMEOUT_TICKS -- which gives up first and takes everybody with it.
#define HOST_PACED_TICKS (30ull * ENGINE_TIMEOUT_TICKS)

// (diagnostics) where a wave runs: XCC id in bits 28..31, HW_ID (wave, SIMD, CU, SH, SE ...) below
__device__ __forceinline__ unsigned hw_where()
{
    unsigned x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(x), "=s"(h));
    return (x << 28) | (h & 0x0fffffffu);
}

__device__ __forceinline__ void signal_add(unsigned *word)
{
    // caller: the payload was stored write-through (store_wt), all storing waves have executed
    // s_waitcnt vmcnt(0) and passed a barrier; one lane calls
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One poll of a hand-off word.  An L2-bypassing (sc1) load is what the protocol uses -- and what round 4 caught returning
// a STALE value for as long as the waiter kept polling: 3 of 40 000 evaluations under the dependency-driven schedule ended in a
// 100 ms time-out whose record read "waited for word 9646 >= 2, saw 1, holds 2 now" (dag_wait), whether the word had been
// published by a write-through store or by a read-modify-write atomic.  The stale copy is the poller's: its own refill can
// install the pre-update line in its XCD's L2 just behind the invalidation the update sent, and every later poll then hits
// it.  A read-modify-write atomic executes at the memory side and cannot be served by that line: every 8th poll (~4 us into
// a wait; most waits are over after the first) is a fetch-add of zero.
__device__ __forceinline__ unsigned poll_word(unsigned *word, unsigned it)
{
    if ((it & 7u) == 7u) return __hip_atomic_fetch_add(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane; returns false on abort / timeout.  ACQUIRE = false: the caller reads the handed-off bytes with
// load_wt only
// code: what the abort word is set to on a time-out (who gave up: diagnostic, any non-zero value aborts)
template <bool ACQUIRE = true>
__device__ __forceinline__ bool wait_ge(unsigned *word, unsigned need, unsigned *abort_word, unsigned code = 1u,
                                        unsigned long long ticks = ENGINE_TIMEOUT_TICKS)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned max_polls = (unsigned)(ticks / 100ull);
    for (unsigned it = 0;; ++it) {
        if (poll_word(word, it) >= need) break;
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if (it > max_polls || ((it & 1023u) == 1023u && __builtin_amdgcn_s_memrealtime() - t0 > WALL_BACKSTOP_TICKS)) {
            __hip_atomic_store(abort_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(16);
    }
    if (ACQUIRE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // compiler ordering only
    return true;
}

// Diagonal-block engine: ONE persistent workgroup (8 waves) that factors the 256x256 diagonal blocks
// t = t0, t0+2, ... of a factorisation while the main stream's trailing update runs.  Per block (tiles
// t and t+1):
//   wait in[t]   >= 3 : diagonal tile t is updated            -> factor it (potrf_tile_body), raise out[t]
//   wait in[t+1] >= 7 : tiles (t+1,t) and (t+1,t+1) are updated
//   X = A(t+1,t) L(t)^-T   : 8 waves x 16-row strips in registers, L(t) and its Q operands still in LDS;
//                            stored to global memory and to LDS, raise xr[t]
//   A(t+1,t+1) -= X X^T    : 36 lower 16x16 blocks dealt over the waves, operands from the LDS copy of X
//   factor tile t+1, raise out[t+1]
// so the whole serial part of a panel (two single-workgroup factorisations and the tile between them)
// leaves the main stream, whose panel kernels then only cover the rows below the diagonal block.
// The engine asks for most of a CU's LDS (see launch_potrf_engine), which keeps all but one update
// workgroup off its CU: the fp64 pivot chains of potrf16_regs run on the same DP units as the fp64
// MFMAs and take 3x as long beside them (measured).  Being resident, it needs no stream dependency and
// no room to be found mid-factorisation.
// LDS map (doubles): [0, 36*256) tile image S | [36*256, 37*256) Q of the current block |
// [37*256, 45*256) Q operands of all 8 diagonal blocks | [45*256, 73*256) strips of the tile inverse (DAG blocks) |
// [73*256] ok word; X (64 blocks) overlays from 0.
struct EngineArgs {
    double *A; size_t lda;
    int t0, nt;
    double *dinv;            // 2 x 2048 doubles, parity of the tile index
    int *info;
    unsigned *in, *out, *xr; // per-tile flag words
    unsigned *abort_word;
    unsigned *alive;         // raised once the workgroup is resident (see engine_gate_kernel)
    double *wbuf;            // DAG schedule: W = L^-1 of diagonal tile t goes to wbuf + t * 128 * 128 (column-major, ld 128,
                             // zero above the diagonal for good), published BEFORE out[t]: operand of the panel tasks
    double *pbuf;            // DAG schedule: X = A(t+1,t) L(t)^-T is stored into this second buffer too (same index as in A)
    int dag_until;           // ... for the diagonal tiles t < dag_until only (the head of the factorisation runs the DAG schedule, the
                             // chain-bound rest the classic one, which needs neither)
    unsigned long long *trace;   // diagnostics (may be null): 8 stamps of the 100 MHz clock per tile pair t / 2 -- in[t] seen,
                             // tile t factored, out[t] raised, in[t+1] seen, xr[t] raised, tile t+1 updated, factored, out[t+1] raised
    unsigned long long in_ticks; // bound of the waits for the INPUT words in[t] / in[t+1]: they are paced by the host (the launch that
                                 // raises them may not be enqueued yet), HOST_PACED_TICKS unless a test shortens it
    double *mbox;                // pair mode: the tiles' mailboxes, 44 x 256 doubles each (index: tile), filled with ~0 (potrf_tile_body)
    int partner;                 // pair mode: index of the PAIR PARTNER's workgroup in this launch (engine_partner_loop); 0: none
};

// Following a tile's factorisation through its mailbox (potrf_tile_body: mbox), shared by the engine's partner and the followers of
// potrf_follow_kernel.  In scope: tid, lane, half = tid >> 8 (wave-uniform), mb = the tile's mailbox, double v[5], int *okp (LDS).
// MBOX_FETCH(j): this thread's words of column block j -- blocks (j .. 7, j), then its Q operands: (9 - j) x 256 values, contiguous
// -- into v (value tid + 512 i is element tid & 255 of block (tid >> 8) + 2 i: one 32-bit offset per thread, nothing that would live
// across an unrolled loop).  MBOX_COMPLETE(j, abort word, code): fetch again until none of this WAVE's words is the fill pattern
// (bounded; every eighth retry a read-modify-write: a reader's own refill can leave it a stale line, poll_word); on a time-out or
// an abort *okp = 0 and the caller leaves behind its next barrier.
#define MBOX_FETCH(jn)                                                                                                        \
{                                                                                                                        \
    unsigned mo = 8u * (unsigned)tid;                                                                                    \
    asm volatile("" : "+v"(mo));                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                                      \
        v[i] = 0.0;                                                                                                      \
        if (half + 2 * i <= 8 - (jn))                                                                                    \
            v[i] = load_wt((const double *)((const char *)(mb + MBOX_OFF(jn)) + (mo + 4096u * (unsigned)i)));            \
    }                                                                                                                    \
}
#define MBOX_COMPLETE(jn, ABORTW, CODE)                                                                                                    \
for (unsigned it = 0;; ++it) {                                                                                           \
    bool missing = false;                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                                        \
        if (half + 2 * i <= 8 - (jn)) missing = missing || __double_as_longlong(v[i]) == -1ll;                           \
    if (__builtin_amdgcn_ballot_w64(missing) == 0ull) break;                                                             \
    const bool late_ = it > (unsigned)(ENGINE_TIMEOUT_TICKS / 100ull);                                                   \
    if (late_ || ((it & 7u) == 7u && __hip_atomic_load((ABORTW), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { \
        if (lane == 0) {                                                                                                 \
            if (late_) __hip_atomic_store((ABORTW), (CODE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            *okp = 0;                                                                                                    \
        }                                                                                                                \
        break;                                                                                                           \
    }                                                                                                                    \
    __builtin_amdgcn_s_sleep(2);                                                                                         \
    if ((it & 7u) == 7u) {      /* (a reader's own refill can leave it a stale line: poll_word -- a read-modify-write cannot) */ \
        unsigned mo = 8u * (unsigned)tid;                                                                                \
        asm volatile("" : "+v"(mo));                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                                    \
            if (half + 2 * i <= 8 - (jn) && __double_as_longlong(v[i]) == -1ll)                                          \
                v[i] = __longlong_as_double((long long)__hip_atomic_fetch_or(                                            \
                    (unsigned long long *)((char *)const_cast<double *>(mb + MBOX_OFF(jn)) + (mo + 4096u * (unsigned)i)), 0ull, \
                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));                                                        \
    } else                                                                                                               \
        MBOX_FETCH(jn)                                                                                                \
}

// ... and for workgroups of 256 threads (the panel kernel): value tid + 256 i is element tid of block i; double v9[9].
#define MBOX_FETCH256V(jn, VV)                                                                                                    \
{                                                                                                                             \
    unsigned mo = 8u * (unsigned)tid;                                                                                         \
    asm volatile("" : "+v"(mo));                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 9; ++i) {                                                                           \
        VV[i] = 0.0;                                                                                                          \
        if (i <= 8 - (jn)) VV[i] = load_wt((const double *)((const char *)(mb + MBOX_OFF(jn)) + (mo + 2048u * (unsigned)i))); \
    }                                                                                                                         \
}
#define MBOX_COMPLETE256V(jn, VV, ABORTW, CODE)                                                                                   \
for (unsigned it = 0;; ++it) {                                                                                                \
    bool missing = false;                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 9; ++i)                                                                             \
        if (i <= 8 - (jn)) missing = missing || __double_as_longlong(VV[i]) == -1ll;                                          \
    if (__builtin_amdgcn_ballot_w64(missing) == 0ull) break;                                                                  \
    const bool late_ = it > (unsigned)(ENGINE_TIMEOUT_TICKS / 100ull);                                                        \
    if (late_ || ((it & 7u) == 7u && __hip_atomic_load((ABORTW), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {        \
        if (lane == 0) {                                                                                                      \
            if (late_) __hip_atomic_store((ABORTW), (CODE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                      \
            *okp = 0;                                                                                                         \
        }                                                                                                                     \
        break;                                                                                                                \
    }                                                                                                                         \
    __builtin_amdgcn_s_sleep(2);                                                                                              \
    if ((it & 7u) == 7u) {                                                                                                    \
        unsigned mo = 8u * (unsigned)tid;                                                                                     \
        asm volatile("" : "+v"(mo));                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 9; ++i)                                                                         \
            if (i <= 8 - (jn) && __double_as_longlong(VV[i]) == -1ll)                                                         \
                VV[i] = __longlong_as_double((long long)__hip_atomic_fetch_or(                                                \
                    (unsigned long long *)((char *)const_cast<double *>(mb + MBOX_OFF(jn)) + (mo + 2048u * (unsigned)i)), 0ull, \
                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));                                                             \
    } else                                                                                                                    \
        MBOX_FETCH256V(jn, VV)                                                                                                     \
}

#define MBOX_FETCH256(jn) MBOX_FETCH256V(jn, v9)
#define MBOX_COMPLETE256(jn, ABORTW, CODE) MBOX_COMPLETE256V(jn, v9, ABORTW, CODE)

// The pair partner (round 5, COCONS_ENGINE_PAIR): a second workgroup of the engine's launch, on a CU of its own, that takes the
// SECOND tile of every diagonal block -- and everything between the two tiles -- off the engine's hands, and does the part that
// depends on the first tile's factor WHILE that factor is being formed.  Until now a block was four passes of one workgroup,
// one behind the other: tile t (25 us) | X = A(t+1,t) L(t)^-T (15) | A(t+1,t+1) -= X X^T (11.5) | tile t+1 (25).  But column block
// j of X needs nothing of L(t) beyond ITS column block j, and the update of tile t+1 with column block j of X nothing beyond that:
// the partner holds its strip of A(t+1,t) and its blocks of A(t+1,t+1) in registers, follows the engine's progress word
// (potrf_tile_body: prog) column block by column block -- fetch L(j..7, j) and its Q operands into LDS, solve, exchange X(., j)
// through LDS, update -- and is a few microseconds behind the engine when tile t is done; it then factors tile t+1 straight from
// the image it holds (no trip through memory), while the engine is free for tile t+2.  Same operations on the same operands in
// the same order as the one-workgroup form: the factor is bit-identical.
// LDS of the partner: [0, 36) the image of tile t+1 (while it is factored) | QS | QALL | [45, 63) two stages for a column block of L(t)
// and its Q operands | [63, 71) X(., j) -- both inside the region potrf_tile_body uses for the inverse's strips afterwards.
template <bool DAG>
__device__ __forceinline__ void engine_partner_loop(const EngineArgs &e, double *smem)
{
    double *QALL = smem + 37 * 256;
    double *XI = smem + 45 * 256;
    double *LST = smem + 45 * 256;                 // two stages of 9 blocks
    double *XJ = smem + 63 * 256;
    int *okp = (int *)(smem + 73 * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *A = e.A;
    const size_t lda = e.lda;
    if (tid == 0) {
        __hip_atomic_fetch_add(e.alive + 16 + ((hw_where() >> 28) & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(e.alive + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int t = e.t0; t + 1 < e.nt; t += 2) {
        if (tid == 0) *okp = wait_ge<false>(e.in + t + 1, 7u, e.abort_word, abort_code(ABORT_ENGINE_IN1, (unsigned)t), e.in_ticks) ? 1 : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (*okp == 0) return;
        __syncthreads();
        unsigned long long *tr = (DAG && e.trace && t >= 2) ? e.trace + 8 * (size_t)(t >> 1) : nullptr;
        if (tr && tid == 0) tr[3] = __builtin_amdgcn_s_memrealtime();
        const bool dag_blk = DAG && t >= 2 && t < e.dag_until;      // (the first block's panel is formed by classic kernels: no inverses)
        const int c0 = t * TILE, c1 = (t + 1) * TILE;
        // this wave's 16 x 128 strip of A(t+1,t) and its 4 or 5 lower blocks of A(t+1,t+1)  (addresses: see the strip solve below)
        const double *Sb = A + (size_t)(c1 + 16 * wave) + (size_t)c0 * lda;
        const unsigned ldab = 8u * (unsigned)lda;
        unsigned lo = 8u * (unsigned)(lane & 15) + (unsigned)(lane >> 4) * ldab;
        asm volatile("" : "+v"(lo));
        d4 B[8], C[5];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                B[j][r] = load_wt((const double *)((const char *)Sb + (lo + (unsigned)(16 * j + 4 * r) * ldab)));
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int bb = wave + 8 * i;
            C[i] = (d4){0.0, 0.0, 0.0, 0.0};
            if (bb < 36) {
                const int ib = c_tri_ib[bb], kb = bb - ib * (ib + 1) / 2;
                C[i] = glb_blk_wt(A, lda, c1 + 16 * ib, c1 + 16 * kb, lane);
            }
        }
        // Column block j of L(t) -- blocks (j .. 7, j), then its Q operands: (9 - j) x 256 values, contiguous in the tile's mailbox --
        // travels mailbox -> registers -> LDS stage (two stages, in turn).  A fetch is complete when none of its words is the
        // mailbox's fill pattern any more (potrf_tile_body: mbox); every wave looks at its own words and fetches again until
        // then (bounded like every wait), the barrier behind the stage's stores makes it the workgroup's.  The fetch of column block
        // j + 1 is issued BEFORE the tile update with column block j: while this workgroup lags behind the engine it comes back
        // complete, and a column block costs max(fetch, solve + update); once it has caught up, one round trip behind the
        // engine's stores.  (The first version waited for a progress word, then fetched, solved and updated one after the
        // other: 6.4 us per column block against the engine's 3 .. 4, and 19 us behind it at the end; with the fetch ahead of the
        // update, but still behind a drained flag: 12 us behind.)
        const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
        const double *mb = e.mbox + (size_t)t * (44 * 256);
        double v[5];
        MBOX_FETCH(0)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double *LS = LST + (j & 1) * (9 * 256);
            MBOX_COMPLETE(j, e.abort_word, abort_code(ABORT_PARTNER, (unsigned)t))
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (half + 2 * i <= 8 - j) LS[tid + 512 * i] = v[i];
            __syncthreads();
            if (*okp == 0) return;
            int ln = lane;               // (LDS addresses are formed here, per column block: hoisted out of the loops they spill)
            asm volatile("" : "+v"(ln));
            {
                d4 L = lds_blk(LS, ln);
                double Q[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) Q[s] = LS[(8 - j) * 256 + s * 64 + ln];
                trsm16(B[j], L, Q);
                lds_blk_store(XJ + wave * 256, ln, B[j]);
                d4 NX = -B[j];
#pragma unroll
                for (int jj = j + 1; jj < 8; ++jj) {
                    d4 Lb = lds_blk(LS + (jj - j) * 256, ln);
                    blk_mma(B[jj], NX, Lb);
                }
            }
            __syncthreads();
            if (j < 7) MBOX_FETCH(j + 1)
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int bb = wave + 8 * i;
                if (bb < 36) {
                    const int ib = c_tri_ib[bb], kb = bb - ib * (ib + 1) / 2;
                    d4 P = lds_blk(XJ + ib * 256, ln);
                    d4 Qk = lds_blk(XJ + kb * 256, ln);
                    P = -P;
                    blk_mma(C[i], P, Qk);
                }
            }
        }
        // X to memory (both buffers under the dependency-driven schedule), the updated tile into the image it is factored from
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                store_wt((double *)((char *)const_cast<double *>(Sb) + (lo + (unsigned)(16 * j + 4 * r) * ldab)), B[j][r]);
        if (dag_blk) {
            double *Pb = e.pbuf + (size_t)(c1 + 16 * wave) + (size_t)c0 * lda;
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    store_wt((double *)((char *)Pb + (lo + (unsigned)(16 * j + 4 * r) * ldab)), B[j][r]);
        }
        int ln2 = lane, tl = tid;
        asm volatile("" : "+v"(ln2), "+v"(tl));
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int bb = wave + 8 * i;
            if (bb < 36) lds_blk_store(smem + bb * 256, ln2, C[i]);
        }
        __syncthreads();
        // (the diagonal blocks symmetric, like potrf_tile_body's fetch leaves them: upper half mirrored from the lower)
        {
            const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
            const int k = (tl >> 4) & 15, r = tl & 15;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = half + 2 * i;
                double *blk = smem + (d * (d + 1) / 2 + d) * 256;
                if (r < k) blk[k * 16 + r] = blk[r * 16 + k];
            }
        }
        __syncthreads();
        // (xr[t] is raised from inside the tile factorisation, once the copies of X have drained: nothing on the chain needs them)
        if (tr && tid == 0) { tr[4] = __builtin_amdgcn_s_memrealtime(); tr[5] = tr[4]; }
        // (the second tile goes into a mailbox too: the panel kernel's workgroups follow both tiles, panel_pair_kernel)
        potrf_tile_body<true, true>(A, lda, c1, e.dinv + (size_t)((t + 1) & 1) * 2048, e.info, smem, dag_blk ? QALL : nullptr,
                                    dag_blk ? e.wbuf + (size_t)(t + 1) * TILE * TILE : nullptr, XI,
                                    e.mbox + (size_t)(t + 1) * (44 * 256), e.xr + t);
        if (tr && tid == 0) tr[6] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) signal_add(e.out + t + 1);
        if (tr && tid == 0) tr[7] = __builtin_amdgcn_s_memrealtime();
    }
}

// DAG: the instantiation for the dependency-driven schedule (dag_kernel): tile inverses and the second copy of X.  The
// classic instantiation does not contain those paths at all (they would cost it registers).
template <bool DAG>
__global__ void __launch_bounds__(512)
potrf_engine_kernel(EngineArgs e)
{
    extern __shared__ double smem[];
    double *QALL = smem + 37 * 256;
    double *XS = smem;
    double *XI = smem + 45 * 256;                  // strips of the tile inverse while a tile is factored (28 blocks, potrf_tile_body)
    int *okp = (int *)(smem + 73 * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *A = e.A;
    const size_t lda = e.lda;
    if (blockIdx.x != 0) {
        // The pair partner: workgroup 8 of this launch -- the one the dispatcher deals to the same XCD as workgroup 0 (round robin
        // over the eight XCDs; affinity only: the partner registers where it really runs) --, on a CU of its own like the engine
        // (the launch's LDS request keeps everything else off it).  The others leave at once.
        if (e.partner != 0 && (int)blockIdx.x == e.partner) engine_partner_loop<DAG>(e, smem);
        return;
    }
    const bool pair = e.partner != 0;
    // (the word also says WHERE: 1 + the id of the XCD the workgroup runs on; word 16 + XCD counts the workgroups of this launch
    // per XCD -- the engine, its partner --: dag_kernel keeps those XCDs less than full)
    if (tid == 0) {
        const unsigned x = hw_where() >> 28;
        __hip_atomic_fetch_add(e.alive + 16 + (x & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(e.alive, 1u + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int t = e.t0; t < e.nt; t += 2) {
        if (tid == 0) *okp = wait_ge<false>(e.in + t, 3u, e.abort_word, abort_code(ABORT_ENGINE_IN0, (unsigned)t), e.in_ticks) ? 1 : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (*okp == 0) return;
        unsigned long long *tr = (DAG && e.trace && t >= 2) ? e.trace + 8 * (size_t)(t >> 1) : nullptr;
        if (tr && tid == 0) {
            tr[0] = __builtin_amdgcn_s_memrealtime();
            // (the unused record of pair 0: where the engine ran first, where it runs now, the pair at which that changed)
            const unsigned long long w = hw_where();
            if (e.trace[0] == 0) e.trace[0] = w;
            if (e.trace[1] != w) { e.trace[1] = w; e.trace[2] = (unsigned long long)(t >> 1); }
        }
        const bool dag_blk = DAG && t >= 2 && t < e.dag_until;      // (the first block's panel is formed by classic kernels: no inverses)
        // (DAG blocks: W = L^-1 of the tile comes out of the factorisation itself, complete before out[t])
        potrf_tile_body<true>(A, lda, t * TILE, e.dinv + (size_t)(t & 1) * 2048, e.info, smem, QALL,
                              dag_blk ? e.wbuf + (size_t)t * TILE * TILE : nullptr, XI,
                              pair ? e.mbox + (size_t)t * (44 * 256) : nullptr);
        __syncthreads();
        if (tr && tid == 0) tr[1] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) signal_add(e.out + t);
        if (tr && tid == 0) tr[2] = __builtin_amdgcn_s_memrealtime();
        if (t + 1 >= e.nt) return;
        if (pair) continue;              // (the rest of the block is the partner's)

        if (tid == 0) *okp = wait_ge<false>(e.in + t + 1, 7u, e.abort_word, abort_code(ABORT_ENGINE_IN1, (unsigned)t), e.in_ticks) ? 1 : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (*okp == 0) return;
        if (tr && tid == 0) tr[3] = __builtin_amdgcn_s_memrealtime();
        const int c0 = t * TILE, c1 = (t + 1) * TILE;
        {   // X = A(t+1,t) L(t)^-T : this wave's 16 x 128 strip
            // (addresses: a wave-uniform base and ONE 32-bit per-lane offset, re-derived for the stores -- thirty-two
            // 64-bit pointers held across the solve were what spilled this kernel in round 3: 88 B of scratch per lane,
            // and a first dispatch that has to wait for the runtime to provide scratch memory)
            const double *Sb = A + (size_t)(c1 + 16 * wave) + (size_t)c0 * lda;
            const unsigned ldab = 8u * (unsigned)lda;
            unsigned lo = 8u * (unsigned)(lane & 15) + (unsigned)(lane >> 4) * ldab;
            asm volatile("" : "+v"(lo));           // (not loop-invariant: the 32 addresses must not live across the blocks)
            d4 B[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    B[j][r] = load_wt((const double *)((const char *)Sb + (lo + (unsigned)(16 * j + 4 * r) * ldab)));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                d4 L = lds_blk(smem + (j * (j + 1) / 2 + j) * 256, lane);
                double Q[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) Q[s] = QALL[j * 256 + s * 64 + lane];
                trsm16(B[j], L, Q);
                d4 NX = -B[j];
#pragma unroll
                for (int jj = j + 1; jj < 8; ++jj) {
                    d4 Lb = lds_blk(smem + (jj * (jj + 1) / 2 + j) * 256, lane);
                    blk_mma(B[jj], NX, Lb);
                }
            }
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    store_wt((double *)((char *)const_cast<double *>(Sb) + (lo + (unsigned)(16 * j + 4 * r) * ldab)), B[j][r]);
            if (dag_blk) {
                double *Pb = e.pbuf + (size_t)(c1 + 16 * wave) + (size_t)c0 * lda;
                asm volatile("" : "+v"(lo));
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        store_wt((double *)((char *)Pb + (lo + (unsigned)(16 * j + 4 * r) * ldab)), B[j][r]);
            }
            __syncthreads();                       // every wave is done with the image of L(t)
#pragma unroll
            for (int j = 0; j < 8; ++j) lds_blk_store(XS + (wave * 8 + j) * 256, lane, B[j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) signal_add(e.xr + t);
        if (tr && tid == 0) tr[4] = __builtin_amdgcn_s_memrealtime();
        // A(t+1,t+1) -= X X^T : lower blocks b = wave, wave + 8, ... -- the next block of the wave is fetched while the current
        // one is multiplied (round 5: its memory round trip used to stand in front of every block's products; all of a wave's
        // 4 or 5 blocks up front spilled the kernel)
        {
            d4 cur = glb_blk_wt(A, lda, c1 + 16 * c_tri_ib[wave], c1 + 16 * (wave - c_tri_ib[wave] * (c_tri_ib[wave] + 1) / 2), lane);
            for (int bb = wave; bb < 36; bb += 8) {
                const int ib = c_tri_ib[bb], kb = bb - ib * (ib + 1) / 2;
                d4 nxt = cur;
                if (bb + 8 < 36) {
                    const int ib2 = c_tri_ib[bb + 8], kb2 = bb + 8 - ib2 * (ib2 + 1) / 2;
                    nxt = glb_blk_wt(A, lda, c1 + 16 * ib2, c1 + 16 * kb2, lane);
                }
                d4 acc = cur;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    d4 P = lds_blk(XS + (ib * 8 + j) * 256, lane);
                    d4 Qk = lds_blk(XS + (kb * 8 + j) * 256, lane);
                    P = -P;
                    blk_mma(acc, P, Qk);
                }
                glb_blk_store_wt(A, lda, c1 + 16 * ib, c1 + 16 * kb, lane, acc);
                cur = nxt;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // X in LDS is dead; the tile is re-read from memory
        if (tr && tid == 0) tr[5] = __builtin_amdgcn_s_memrealtime();
        potrf_tile_body<true>(A, lda, c1, e.dinv + (size_t)((t + 1) & 1) * 2048, e.info, smem, dag_blk ? QALL : nullptr,
                              dag_blk ? e.wbuf + (size_t)(t + 1) * TILE * TILE : nullptr, XI);
        if (tr && tid == 0) tr[6] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) signal_add(e.out + t + 1);
        if (tr && tid == 0) tr[7] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------
// Tile factorisation and the panel solve below it in ONE launch (round 5, COCONS_POTRF_FOLLOW; the plain and the band-limited
// schedule): workgroup 0 factors the tile and publishes every finished block in the tile's mailbox (potrf_tile_body: mbox); the
// other workgroups -- 128 rows each, a wave 16 -- hold their strips of the rows below in registers and FOLLOW the factorisation
// column block by column block, like the engine's partner does: X(., j) = (B(., j) - sum_{k<j} X(., k) L(j,k)^T) L(j,j)^-T is
// formed ~a round trip behind column block j of the factor, and the rows are solved ~6 us after the tile is factored.  Until now:
// the tile's launch (25 us), a boundary, then a panel-solve launch that fetched the whole factor before it began (8 .. 13 us) --
// per tile column of the taper path's 79, and twice per block of the plain schedule the batch slots run.  (A first attempt at ONE
// launch -- solve workgroups that waited for a word behind the complete factor -- was slower than two launches and removed.)
// Workgroup 0 is dispatched first and waits for nothing, so the followers' bounded waits never run out unless something else has.
// Same operations in the same order as potrf_tile_kernel | trsm_tile_kernel: bit-identical.
// rows: strips of 64 -- nb1 of them from r0, the others from e0 (launch_trsm_tile's two ranges); a follower takes two.
__global__ void __launch_bounds__(512)
potrf_follow_kernel(double *A, size_t lda, int c0, double *q_out, int *info, double *mbox, int r0, int nb1, int e0, int nstrips,
                    unsigned *abort_word)
{
    extern __shared__ double smem[];
    if (blockIdx.x == 0) {
        potrf_tile_body<false>(A, lda, c0, q_out, info, smem, nullptr, nullptr, nullptr, mbox);
        return;
    }
    double *LST = smem;                            // two stages of 9 blocks
    int *okp = (int *)(smem + 18 * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int strip = 2 * ((int)blockIdx.x - 1) + half;
    const bool valid = strip < nstrips;
    const int rs = (strip < nb1 ? r0 + 64 * strip : e0 + 64 * (strip - nb1)) + 16 * (wave & 3);
    const double *mb = mbox;
    d4 B[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) B[j] = valid ? glb_blk(A, lda, rs, c0 + 16 * j, lane) : (d4){0.0, 0.0, 0.0, 0.0};
    if (tid == 0) *okp = 1;
    __syncthreads();
    double v[5];
    MBOX_FETCH(0)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double *LS = LST + (j & 1) * (9 * 256);
        MBOX_COMPLETE(j, abort_word, abort_code(ABORT_FOLLOW, (unsigned)(c0 / TILE)))
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (half + 2 * i <= 8 - j) LS[tid + 512 * i] = v[i];
        __syncthreads();
        if (*okp == 0) return;
        if (j < 7) MBOX_FETCH(j + 1)               // (in flight during the solve with column block j)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        d4 L = lds_blk(LS, ln);
        double Q[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) Q[s] = LS[(8 - j) * 256 + s * 64 + ln];
        trsm16(B[j], L, Q);
        d4 NX = -B[j];
#pragma unroll
        for (int jj = j + 1; jj < 8; ++jj) {
            d4 Lb = lds_blk(LS + (jj - j) * 256, ln);
            blk_mma(B[jj], NX, Lb);
        }
    }
    if (valid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) glb_blk_store(A, lda, rs, c0 + 16 * j, lane, B[j]);
    }
}

// The 36 lower blocks of the factored diagonal tile at (c0, c0) and its Q operands into LDS (256 threads): EVERY global load is
// issued before the first LDS store -- one round trip.  (Until round 5 the panel kernels fetched block by block, a load and a
// store at a time: 36 dependent round trips, ~10 of the 14.6 us a panel solve took whatever its number of rows; the kernel
// trace of the tail at n = 4096, tools/r5_tail_timeline.sh.)
// WT: the tile comes from the engine inside this launch's lifetime (write-through stores there): L1-bypassing loads here, and the
// wait in front needs no acquire (the hand-off protocol above potrf_engine_kernel; an acquire is ~1.7 us, three per panel launch)
template <bool WT>
__device__ __forceinline__ void fetch_factor_tile(const double *A, size_t lda, int c0, const double *qin, double *SL, double *QS, int tid)
{
    const int i = tid & 15, k = tid >> 4;
    const double *src = A + (size_t)(c0 + i) + (size_t)(c0 + k) * lda;
    double v[36], q[8];
    {
        int b = 0;
#pragma unroll
        for (int ib = 0; ib < 8; ++ib)
#pragma unroll
            for (int kb = 0; kb <= ib; ++kb, ++b) {
                const double *sp = src + (size_t)(16 * ib) + (size_t)(16 * kb) * lda;
                v[b] = WT ? load_wt(sp) : *sp;
            }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = WT ? load_wt(qin + tid + 256 * e) : qin[tid + 256 * e];
#pragma unroll
    for (int b = 0; b < 36; ++b) SL[b * 256 + k * 16 + i] = v[b];
#pragma unroll
    for (int e = 0; e < 8; ++e) QS[tid + 256 * e] = q[e];
}

// ---------------------------------------------------------------------------
// Panel of a two-tile block in ONE launch (engine schedule, round 5): for the rows below the diagonal block
//     X0 = B0 L(t)^-T  |  B1 -= X0 X(t+1,t)^T  |  X1 = B1 L(t+1)^-T
// -- until now three launches (trsm_tile_kernel, the in-panel update_kernel, trsm_tile_kernel), each of which read its strip
// from memory and wrote it back, the second and third behind a drained chip.  A workgroup owns 64 rows (a wave 16) for the whole
// sequence: both 16 x 128 strips stay in registers, the three waits -- out[t], xr[t], out[t+1] -- are met where the data is needed,
// and the rows are done ~8 us after the engine's second tile instead of ~16 + a boundary.  Same operations on the same operands
// in the same order as the three kernels (the product accumulated from zero over ascending k and subtracted once, as
// update_kernel does): bit-identical.
__global__ void __launch_bounds__(256)
panel_pair_kernel(double *A, size_t lda, int c0, int r0, const double *q0, const double *q1, unsigned *out0, unsigned *xrw,
                  unsigned *out1, unsigned *abort_word, const double *mb0, const double *mb1,
                  double *smb, int nstrip, int npub, unsigned *sig, int sig_tile, double *xmb)
{
    // 136 KB: L of the current tile (36 blocks) and its Q operands (8) -- or, between the two solves, all 64 blocks of X(t+1,t)
    __shared__ double SM[68 * 256];
    __shared__ int ok;
    double *SL = SM, *QS = SM + 36 * 256, *XS = SM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c1 = c0 + TILE;
    // Split panel (COCONS_PANEL_SPLIT, xmb != null): a strip is TWO workgroups.  The first (role A, block index = strip) follows tile
    // t and forms X0 -- and publishes every finished 16 x 16 block of it in the strip's exchange mailbox; the second (role B, block
    // index nstrip + strip) holds B1, brings X(t+1,t) into LDS and follows the first: the in-panel product advances one column
    // block of X0 behind its formation (accumulated from zero over ascending k and subtracted once, as ever), then it follows tile
    // t+1 for X1.  With everything there to be read a strip takes solve + 1.5 us + solve instead of solve | fetch | product |
    // solve on one wave per SIMD: ~27 us instead of 41.  The second workgroup puts the fill pattern back into every block it has
    // read (its only reader), the whole region is filled again with the others before the next factorisation.
    const bool split = xmb != nullptr;
    const int nfirst = split ? 2 * nstrip : nstrip;        // workgroups in front of the diagonal-tile ones
    if ((int)blockIdx.x >= nfirst) {
        // ---- the NEXT diagonal block's update, inside this launch (COCONS_PANEL_DIAG): workgroup nfirst + dd takes the dd-th of its
        // ten (three) 64 x 64 tiles, C(ta, tb) -= sum_k X(ta, k) X(tb, k)^T over the sixteen 16-column blocks of this panel -- and
        // FOLLOWS the strips that form them: the first npub strip workgroups publish every finished 16 x 16 block of X in a strip
        // mailbox (smb: strip, column block, wave; filled with ~0 like the tiles' mailboxes), and a wave here reads its five blocks
        // of a column block until none of their words is the fill pattern.  No barrier, no LDS: a wave owns 16 rows of the tile.
        // The tile is complete ~2 round trips behind the second diagonal tile's last block and raises the engine's input word
        // itself; the trailing update that follows leaves these tiles alone (its first tiles, ~10 us of latency-bound products
        // behind a kernel boundary, were what the engine's next block waited for).  Accumulated from zero over ascending k and
        // subtracted once, like update_kernel: bit-identical.
        const int dd = (int)blockIdx.x - nfirst;
        const int ta = c_tri_ib[dd], tb = dd - ta * (ta + 1) / 2;
        const int rI = r0 + 64 * ta + 16 * wave, cJ = r0 + 64 * tb;
        d4 C[4], acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            C[c] = glb_blk(A, lda, rI, cJ + 16 * c, lane);
            acc[c] = (d4){0.0, 0.0, 0.0, 0.0};
        }
        const double *pi = smb + ((size_t)ta * 16 * 4 + wave) * 256 + lane;      // + k * 1024: block (ta, k, wave)
        const double *pj = smb + ((size_t)tb * 16 * 4) * 256 + lane;             // + k * 1024 + c * 256
        double x[2][20];
#define DIAG_FETCH(kk, buf)                                                                                    \
        {                                                                                                      \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) x[buf][r] = load_wt(pi + (size_t)(kk) * 1024 + 64 * r); \
            _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                       \
                _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
                    x[buf][4 + 4 * c + r] = load_wt(pj + (size_t)(kk) * 1024 + 256 * c + 64 * r);                \
        }
        DIAG_FETCH(0, 0)
        bool good = true;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int cur = k & 1;
            for (unsigned it = 0; good; ++it) {
                bool missing = false;
#pragma unroll
                for (int e = 0; e < 20; ++e) missing = missing || __double_as_longlong(x[cur][e]) == -1ll;
                if (__builtin_amdgcn_ballot_w64(missing) == 0ull) break;
                const bool late_ = it > (unsigned)(ENGINE_TIMEOUT_TICKS / 100ull);
                if (late_ || ((it & 7u) == 7u && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    if (late_ && lane == 0) __hip_atomic_store(abort_word, abort_code(ABORT_XCHG, (unsigned)(c0 / TILE)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                if ((it & 7u) == 7u) {
#pragma unroll
                    for (int e = 0; e < 20; ++e)
                        if (__double_as_longlong(x[cur][e]) == -1ll) {
                            const double *ad = e < 4 ? pi + (size_t)k * 1024 + 64 * e
                                                     : pj + (size_t)k * 1024 + 256 * ((e - 4) >> 2) + 64 * ((e - 4) & 3);
                            x[cur][e] = __longlong_as_double((long long)__hip_atomic_fetch_or(
                                (unsigned long long *)const_cast<double *>(ad), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        }
                } else
                    DIAG_FETCH(k, cur)
            }
            if (k + 1 < 16) DIAG_FETCH(k + 1, cur ^ 1)
            d4 Xi = {x[cur][0], x[cur][1], x[cur][2], x[cur][3]};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                d4 Xj = {x[cur][4 + 4 * c], x[cur][5 + 4 * c], x[cur][6 + 4 * c], x[cur][7 + 4 * c]};
                blk_mma(acc[c], Xi, Xj);
            }
        }
#undef DIAG_FETCH
        if (tid == 0) ok = 1;
        __syncthreads();
        if (!good && lane == 0) ok = 0;
        __syncthreads();
        if (!ok) return;
#pragma unroll
        for (int c = 0; c < 4; ++c) glb_blk_store_wt(A, lda, rI, cJ + 16 * c, lane, C[c] - acc[c]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) signal_add(sig + sig_tile + (ta >> 1));
        return;
    }
    const bool roleB = split && (int)blockIdx.x >= nstrip;
    const int strip = (int)blockIdx.x - (roleB ? nstrip : 0);
    const int rs = r0 + 64 * strip + 16 * wave;
    const bool pub = smb != nullptr && strip < npub;
    double *sp = smb + ((size_t)strip * 16 * 4 + wave) * 256;          // + k * 1024: this wave's block of column block k
    double *xp = split ? xmb + ((size_t)strip * 8 * 4 + wave) * 256 : nullptr;      // + k * 1024: the exchange mailbox, likewise
    if (roleB) {
        // ---- role B: B1 -= X0 X(t+1,t)^T behind role A's X0, then X1 = B1 L(t+1)^-T behind tile t+1
        int *okp = &ok;
        d4 B1[8], acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            B1[j] = glb_blk(A, lda, rs, c1 + 16 * j, lane);
            acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
        }
        if (tid == 0) ok = wait_ge<false>(xrw, 1u, abort_word, abort_code(ABORT_INPANEL, (unsigned)(c0 / TILE))) ? 1 : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!ok) return;
        {
            const int i = tid & 15, k = tid >> 4;
            const double *Xg = A + (size_t)(c1 + i) + (size_t)(c0 + k) * lda;
            double st[2][16];
#pragma unroll
            for (int b = 0; b < 16; ++b) st[0][b] = load_wt(Xg + (size_t)(16 * (b >> 3)) + (size_t)(16 * (b & 7)) * lda);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r + 1 < 4) {
#pragma unroll
                    for (int b = 0; b < 16; ++b) {
                        const int bb = 16 * (r + 1) + b;
                        st[(r + 1) & 1][b] = load_wt(Xg + (size_t)(16 * (bb >> 3)) + (size_t)(16 * (bb & 7)) * lda);
                    }
                }
#pragma unroll
                for (int b = 0; b < 16; ++b) XS[(16 * r + b) * 256 + k * 16 + i] = st[r & 1][b];
            }
        }
        __syncthreads();
        {
            // (a wave follows its own 16 rows of X0: no barrier; the block is put back to the fill pattern once it is in registers)
            const double *xq = xp + lane;
            double x[2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[0][r] = load_wt(xq + 64 * r);
            bool good = true;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int cur = kb & 1;
                for (unsigned it = 0; good; ++it) {
                    bool missing = false;
#pragma unroll
                    for (int r = 0; r < 4; ++r) missing = missing || __double_as_longlong(x[cur][r]) == -1ll;
                    if (__builtin_amdgcn_ballot_w64(missing) == 0ull) break;
                    const bool late_ = it > (unsigned)(ENGINE_TIMEOUT_TICKS / 100ull);
                    if (late_ || ((it & 7u) == 7u && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        if (late_ && lane == 0) __hip_atomic_store(abort_word, abort_code(ABORT_STRIPBOX, (unsigned)(c0 / TILE)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        good = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double *ad = xq + (size_t)kb * 1024 + 64 * r;
                        if ((it & 7u) == 7u) {
                            if (__double_as_longlong(x[cur][r]) == -1ll)
                                x[cur][r] = __longlong_as_double((long long)__hip_atomic_fetch_or(
                                    (unsigned long long *)const_cast<double *>(ad), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        } else
                            x[cur][r] = load_wt(ad);
                    }
                }
                if (kb + 1 < 8) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[cur ^ 1][r] = load_wt(xq + (size_t)(kb + 1) * 1024 + 64 * r);
                }
                if (good) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) store_wt(const_cast<double *>(xq) + (size_t)kb * 1024 + 64 * r, __longlong_as_double(-1ll));
                }
                d4 Xk = {x[cur][0], x[cur][1], x[cur][2], x[cur][3]};
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    d4 Xb = lds_blk(XS + (jj * 8 + kb) * 256, lane);
                    blk_mma(acc[jj], Xk, Xb);
                }
            }
            if (!good && lane == 0) ok = 0;
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) B1[jj] = B1[jj] - acc[jj];
        {
            const double *mb = mb1;
            double v9[3][9];
            __syncthreads();             // (every wave is done with X(t+1,t): the stages overlay it; and the product's verdict is in)
            if (!ok) return;
            MBOX_FETCH256V(0, v9[0])
            MBOX_FETCH256V(1, v9[1])
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                double *LS = SM + (j & 1) * (9 * 256);
                MBOX_COMPLETE256V(j, v9[j % 3], abort_word, abort_code(ABORT_PANEL, (unsigned)(c1 / TILE)))
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    if (i <= 8 - j) LS[tid + 256 * i] = v9[j % 3][i];
                __syncthreads();
                if (!ok) return;
                if (j + 2 < 8) MBOX_FETCH256V(j + 2, v9[(j + 2) % 3])
                d4 L = lds_blk(LS, lane);
                double Q[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) Q[s] = LS[(8 - j) * 256 + s * 64 + lane];
                trsm16(B1[j], L, Q);
                if (pub) mbox_store(sp + (size_t)(8 + j) * 1024, lane, B1[j]);
                d4 NX = -B1[j];
#pragma unroll
                for (int jj = j + 1; jj < 8; ++jj) {
                    d4 Lb = lds_blk(LS + (jj - j) * 256, lane);
                    blk_mma(B1[jj], NX, Lb);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) glb_blk_store(A, lda, rs, c1 + 16 * j, lane, B1[j]);
        return;
    }
    d4 B0[8], B1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) B0[j] = glb_blk(A, lda, rs, c0 + 16 * j, lane);
    if (!split) {
#pragma unroll
        for (int j = 0; j < 8; ++j) B1[j] = glb_blk(A, lda, rs, c1 + 16 * j, lane);
    }
    // ---- X0 = B0 L(t)^-T
    int *okp = &ok;
    if (mb0) {
        // the engine's pair mode: both tiles are published in mailboxes while they are formed -- the strip FOLLOWS the tile column
        // block by column block (like potrf_follow_kernel's workgroups) and is solved a round trip behind the tile's last block,
        // where waiting for out[t], fetching the factor and solving took ~8 us behind it
        const double *mb = mb0;
        double v9[3][9];             // (TWO column blocks in flight: with the tile already there a column block costs its solve,
        if (tid == 0) ok = 1;        // not a round trip)
        __syncthreads();
        MBOX_FETCH256V(0, v9[0])
        MBOX_FETCH256V(1, v9[1])
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double *LS = SM + (j & 1) * (9 * 256);
            MBOX_COMPLETE256V(j, v9[j % 3], abort_word, abort_code(ABORT_PANEL, (unsigned)(c0 / TILE)))
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (i <= 8 - j) LS[tid + 256 * i] = v9[j % 3][i];
            __syncthreads();
            if (!ok) return;
            if (j + 2 < 8) MBOX_FETCH256V(j + 2, v9[(j + 2) % 3])
            d4 L = lds_blk(LS, lane);
            double Q[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) Q[s] = LS[(8 - j) * 256 + s * 64 + lane];
            trsm16(B0[j], L, Q);
            if (pub) mbox_store(sp + (size_t)j * 1024, lane, B0[j]);
            if (split) mbox_store(xp + (size_t)j * 1024, lane, B0[j]);
            d4 NX = -B0[j];
#pragma unroll
            for (int jj = j + 1; jj < 8; ++jj) {
                d4 Lb = lds_blk(LS + (jj - j) * 256, lane);
                blk_mma(B0[jj], NX, Lb);
            }
        }
    } else {
    if (tid == 0) ok = wait_ge<false>(out0, 1u, abort_word, abort_code(ABORT_PANEL, (unsigned)(c0 / TILE))) ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!ok) return;
    fetch_factor_tile<true>(A, lda, c0, q0, SL, QS, tid);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        d4 L = lds_blk(SL + (j * (j + 1) / 2 + j) * 256, lane);
        double Q[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) Q[s] = QS[j * 256 + s * 64 + lane];
        trsm16(B0[j], L, Q);
        d4 NX = -B0[j];
#pragma unroll
        for (int jj = j + 1; jj < 8; ++jj) {
            d4 Lb = lds_blk(SL + (jj * (jj + 1) / 2 + j) * 256, lane);
            blk_mma(B0[jj], NX, Lb);
        }
    }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) glb_blk_store(A, lda, rs, c0 + 16 * j, lane, B0[j]);
    if (split) return;               // (role A: the rest of the strip is role B's)
    // ---- B1 -= X0 X(t+1,t)^T: all 64 blocks of X(t+1,t) into LDS (over the image of L(t), which is dead), four rounds of sixteen
    // loads per thread with the next round in flight while one is stored
    if (tid == 0) ok = wait_ge<false>(xrw, 1u, abort_word, abort_code(ABORT_INPANEL, (unsigned)(c0 / TILE))) ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                 // (also: every wave is done with L(t))
    if (!ok) return;
    {
        const int i = tid & 15, k = tid >> 4;
        const double *Xg = A + (size_t)(c1 + i) + (size_t)(c0 + k) * lda;      // block (jj, kb) at + 16 jj + 16 kb lda
        double st[2][16];
#pragma unroll
        for (int b = 0; b < 16; ++b) st[0][b] = load_wt(Xg + (size_t)(16 * (b >> 3)) + (size_t)(16 * (b & 7)) * lda);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r + 1 < 4) {
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const int bb = 16 * (r + 1) + b;
                    st[(r + 1) & 1][b] = load_wt(Xg + (size_t)(16 * (bb >> 3)) + (size_t)(16 * (bb & 7)) * lda);
                }
            }
#pragma unroll
            for (int b = 0; b < 16; ++b) XS[(16 * r + b) * 256 + k * 16 + i] = st[r & 1][b];
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                d4 Xb = lds_blk(XS + (jj * 8 + kb) * 256, lane);
                blk_mma(acc, B0[kb], Xb);
            }
            B1[jj] = B1[jj] - acc;
        }
    }
    // ---- X1 = B1 L(t+1)^-T
    if (mb1) {
        const double *mb = mb1;
        double v9[3][9];
        __syncthreads();             // (every wave is done with X(t+1,t): the stages overlay it)
        MBOX_FETCH256V(0, v9[0])
        MBOX_FETCH256V(1, v9[1])
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double *LS = SM + (j & 1) * (9 * 256);
            MBOX_COMPLETE256V(j, v9[j % 3], abort_word, abort_code(ABORT_PANEL, (unsigned)(c1 / TILE)))
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (i <= 8 - j) LS[tid + 256 * i] = v9[j % 3][i];
            __syncthreads();
            if (!ok) return;
            if (j + 2 < 8) MBOX_FETCH256V(j + 2, v9[(j + 2) % 3])
            d4 L = lds_blk(LS, lane);
            double Q[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) Q[s] = LS[(8 - j) * 256 + s * 64 + lane];
            trsm16(B1[j], L, Q);
            if (pub) mbox_store(sp + (size_t)(8 + j) * 1024, lane, B1[j]);
            d4 NX = -B1[j];
#pragma unroll
            for (int jj = j + 1; jj < 8; ++jj) {
                d4 Lb = lds_blk(LS + (jj - j) * 256, lane);
                blk_mma(B1[jj], NX, Lb);
            }
        }
    } else {
    if (tid == 0) ok = wait_ge<false>(out1, 1u, abort_word, abort_code(ABORT_PANEL, (unsigned)(c1 / TILE))) ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                 // (also: every wave is done with X(t+1,t))
    if (!ok) return;
    fetch_factor_tile<true>(A, lda, c1, q1, SL, QS, tid);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        d4 L = lds_blk(SL + (j * (j + 1) / 2 + j) * 256, lane);
        double Q[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) Q[s] = QS[j * 256 + s * 64 + lane];
        trsm16(B1[j], L, Q);
        d4 NX = -B1[j];
#pragma unroll
        for (int jj = j + 1; jj < 8; ++jj) {
            d4 Lb = lds_blk(SL + (jj * (jj + 1) / 2 + j) * 256, lane);
            blk_mma(B1[jj], NX, Lb);
        }
    }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) glb_blk_store(A, lda, rs, c1 + 16 * j, lane, B1[j]);
}

// The engine needs a CU to itself (its 8 waves take every VGPR of the four SIMDs): once a chip-filling launch
// is running, a CU only empties when that launch drains -- and never while workgroups that WAIT for the
// engine sit on every CU.  So the main stream does not start the factorisation's launches before the
// engine is resident: this one-lane kernel waits for its alive word (bounded like every other wait).
__global__ void __launch_bounds__(64)
engine_gate_kernel(unsigned *alive, unsigned *abort_word, unsigned code, unsigned long long ticks, unsigned nhelp,
                   unsigned *raise_in)
{
    // (nhelp > 0: the engine's launch holds that many further workgroups -- the pair partner --, which count themselves in
    // alive[2] once resident: a partner that found no CU would stop the factorisation as surely as a missing engine)
    // (raise_in: the engine also factors the FIRST diagonal block -- nobody updates it, so its input words are raised here, behind
    // everything the main stream did to the matrix before the factorisation: in[0] = 3, in[1] = 7)
    if (threadIdx.x == 0) {
        bool ok = wait_ge<false>(alive, 1u, abort_word, code, ticks);
        if (ok && nhelp) ok = wait_ge<false>(alive + 2, nhelp, abort_word, code + 1u, ticks);
        if (ok && raise_in) {
            __hip_atomic_store(raise_in, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(raise_in + 1, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------
// Panel solve: rows [r0, r1) of block column c0:  X <- X * L(c0)^-T.
// One workgroup = 64 rows; each wave owns a 16 x 128 strip held in registers (8 blocks).
// LDS holds the 36 lower 16x16 blocks of L and the 8 x 4 Q operands.
__global__ void __launch_bounds__(256)
trsm_tile_kernel(double *A, size_t lda, int c0, int r0, const double *qin, unsigned *wait_word, unsigned *abort_word,
                 int nb1, int e0, int own_world, int own_rank, int own_group)
{
    __shared__ double SL[36 * 256];
    __shared__ double QS[8 * 256];
    __shared__ int ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (own_world > 1) {     // sharded evaluation: only the strips of this rank's 256-row blocks
        const int row = (int)blockIdx.x < nb1 ? r0 + 64 * (int)blockIdx.x : e0 + 64 * ((int)blockIdx.x - nb1);
        if (((row / (2 * TILE) / own_group) % own_world) != own_rank) return;
    }
    if (wait_word) {     // the diagonal tile comes from the engine, which may still be at work
        if (tid == 0) ok = wait_ge<false>(wait_word, 1u, abort_word, abort_code(ABORT_PANEL, (unsigned)(c0 / TILE))) ? 1 : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!ok) return;
        fetch_factor_tile<true>(A, lda, c0, qin, SL, QS, tid);
    } else
        fetch_factor_tile<false>(A, lda, c0, qin, SL, QS, tid);
    // rows: workgroups 0 .. nb1-1 cover [r0, r0 + 64 nb1), the others a second range from e0 (the right-hand-side rows
    // under a band-limited factorisation; nb1 = all of them otherwise)
    const int bx = blockIdx.x;
    const int rs = (bx < nb1 ? r0 + 64 * bx : e0 + 64 * (bx - nb1)) + 16 * wave;
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
// Trailing update: C(ti,tj) -= P(ti,:) P(tj,:)^T over K panel columns.  TM x TM tiles, 4 waves x
// (TM/2 x TM/2); instantiated with TM = 64 (each wave 2 x 2 MFMA blocks, up to 8 workgroups per CU).
// Operand tiles stream through LDS in chunks of KC panel columns (KC = 8: 20 KB), register-staged
// double buffering, one barrier per chunk.

struct UpdArgs {
    double *C; size_t ldc;
    const double *P; size_t ldp;   // panel: element (global row, k) at P[row + k*ldp]
    int K;
    int ti0, tj0, lower_only;      // tile indices in units of TM
    int H, W;                      // lower_only: rows / columns of the trapezoid in tiles, 1-D order over its tiles
    unsigned *sig; int sig_tile;   // engine hand-off: workgroups inside the diagonal block (tiles sig_tile, sig_tile+1)
                                   // add 1 to sig[sig_tile] (tile (t,t)) or sig[sig_tile+1] (tiles (t+1,t), (t+1,t+1))
    unsigned *wait_word, *abort_word;  // engine hand-off: every workgroup first waits for *wait_word >= 1
    int ptiles, world, rank;       // sharded path (world > 1): only tiles whose ROW lies in an owned 256-row block: block b
                                   // belongs to rank (b / ptiles) % world (ptiles = blocks per ownership group)
    const int *pmap;               // sharded path: element offset of each 64-row tile's rows in P (owner-packed panel); null:
                                   // the global row index
    int skip_lo, skip_hi;          // sharded path: tiles with BOTH 64-row and 64-column index in [skip_lo, skip_hi) -- a diagonal
                                   // block its owner has updated ahead of the exchange -- are not updated (empty range: none)
    unsigned *queue; unsigned ntiles;  // dynamic tile order (lower_only launches): shared counter, zero at launch; tiles in all
    int Hb, ext0;                  // rows: local tile rows < Hb count from ti0 (tj0 when lower_only), the others from ext0
                                   // (band-limited factorisation: band rows, then the right-hand-side rows)
    int c_wt;                      // every C tile is read with L2-bypassing loads and stored write-through (set_update_c_wt)
    int skew, kblk;                // packed band buffer (kernels.h band_index): C and P are its unshifted base, the operand
                                   // panel is tile column kblk; rows then count from each tile column's own diagonal tile
};

// ROLE only names the instantiation (0 = trailing update, 1 = in-panel / sharded update) so that
// profiler summaries keep the dominant trailing launches apart from the narrow ones
// NW: waves per workgroup.  4 (KC = 8): each wave a quarter of the tile, 8 workgroups per CU.  8 (KC = 16, 512 threads):
// each wave an eighth (32 x 16), 4 workgroups per CU -- the same waves per SIMD, the same work per wave and barrier, but a
// tile is finished in half the time, so the launch drains for half as long at its end.
template <int TM, int KC, int ROLE, int NW = 4>
__global__ void __launch_bounds__(64 * NW, 8)
update_kernel(UpdArgs a)
{
    constexpr int LDT = TM + 16;   // lanes l and l+16 land 128 B apart mod 256 -> conflict-free b64 reads
    constexpr int WY = NW / 2;     // wave grid 2 x WY: a wave's share is TM/2 rows x TM/WY columns
    constexpr int NB = TM / 32;    // 16x16 blocks per wave along the rows
    constexpr int NBY = TM / WY / 16;   // ... and along the columns
    constexpr int TPC = 64 * NW / KC;   // threads per panel column
    constexpr int RPT = TM / TPC;  // rows staged per thread and side
    __shared__ double sI[2][KC * LDT];
    __shared__ double sJ[2][KC * LDT];
    // (no LDS beyond the two operand rings: 20,480 B is exactly an eighth of a CU's; the word the workgroup
    // has to share -- wait result, next tile index -- lives in the padding of the first staged column,
    // rows TM .. LDT-1, which no staging store and no operand read touches)
    unsigned *share = (unsigned *)&sI[0][TM];
    const int tid = threadIdx.x;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.wait_word) {     // operand tile comes from the engine
        if (tid == 0) *share = wait_ge(a.wait_word, 1u, a.abort_word, abort_code(ABORT_INPANEL, (unsigned)a.tj0)) ? 1u : 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned ok = *share;
        __syncthreads();
        if (!ok) return;
    }

    else if (a.abort_word) {   // engine schedule, nothing to wait for: leave at once when the factorisation was given up
        if (tid == 0) *share = __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ab = *share;
        __syncthreads();
        if (ab) return;
    }

    const int nch = a.K / KC;
    typedef double d2 __attribute__((ext_vector_type(2)));

    // Tile index.  Static (a.queue == nullptr): the workgroup's own block index, one tile.  Dynamic: the
    // launch has about as many workgroups as the chip holds and each takes tiles off a shared counter
    // until it runs dry -- CUs that are slower or taken (the engine owns one) simply take fewer, and no
    // workgroup is placed after the first round (while a second queue holds a resident kernel the
    // dispatcher places workgroups at a quarter of its rate).
    // (The first tile is the block index -- thousands of workgroups asking the one counter at the same
    // moment cost 8 us per launch --, later ones are gridDim.x + the counter's value.)
    unsigned L = blockIdx.x;
    for (;;) {
        // per-thread offsets are re-derived for every tile from a laundered thread index: hoisted out of
        // this loop they live through it and spill (the kernel sits exactly at its 64 registers)
        // (and the thread index itself from the wave index, kept scalar, and the lane count: no register held)
        int lane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        const int wave = wave_s, t2 = 64 * wave + lane;
        const int wi = wave & 1, wj = wave >> 1;
        // staging map: thread -> (panel column kc, RPT consecutive rows)
        const int kc = t2 / TPC, rg = (t2 % TPC) * RPT;
        const int ro = (lane >> 4) * LDT + (lane & 15);
        int ti, tj;
        if (a.lower_only) {
            // 1-D order over the tiles (ti >= tj) of the trapezoid, column by column: column j (0-based
            // from tj0) holds H - j tiles and starts at j H - j (j-1)/2.  Bisection in integers: L is
            // uniform, so this stays on the scalar unit.
            int lo = 0, hi = a.W - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (mid * a.H - mid * (mid - 1) / 2 <= (int)L) lo = mid; else hi = mid - 1;
            }
            const int j = lo;
            const int til = j + ((int)L - (j * a.H - j * (j - 1) / 2));
            tj = a.tj0 + j;
            ti = til < a.Hb ? a.tj0 + til : a.ext0 + (til - a.Hb);
        } else {
            const int til = blockIdx.x;
            ti = til < a.Hb ? a.ti0 + til : a.ext0 + (til - a.Hb);
            tj = a.tj0 + blockIdx.y;
        }
        // sharded path (static launches only): the rows of this rank's 256-row blocks, minus a diagonal block done ahead
        if (a.world > 1 && ((ti * TM / (2 * TILE) / a.ptiles) % a.world) != a.rank) return;
        if (ti >= a.skip_lo && ti < a.skip_hi && tj >= a.skip_lo && tj < a.skip_hi) {
            // (a tile somebody else takes care of -- sharded path: a diagonal block done ahead; engine schedule, round 5: the next
            // diagonal block, updated inside the panel's launch, panel_pair_kernel)
            if (!a.queue) return;
            if (t2 == 0) *share = gridDim.x + __hip_atomic_fetch_add(a.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            L = (unsigned)__builtin_amdgcn_readfirstlane((int)*share);
            __syncthreads();
            if (L >= a.ntiles) break;
            continue;
        }
        // does this tile lie inside the diagonal block the engine is waiting for?
        const int sig_Ti = (ti * TM) / TILE - a.sig_tile, sig_Tj = (tj * TM) / TILE - a.sig_tile;
        const bool sig_wg = a.sig != nullptr && sig_Ti >= 0 && sig_Ti <= 1 && sig_Tj >= 0 && sig_Tj <= sig_Ti;
        const bool wt_wg = sig_wg || a.c_wt;
        // the engine's whole chain starts when these ten tiles are done: let them win the issue arbitration on
        // their CU (beside seven other workgroups a tile takes ~70 us, alone ~10)
        if (sig_wg) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);

        // rows: global, or -- packed band buffer -- local to the tile column they are read from / written to
        int rowI_P = ti * TM, rowJ_P = tj * TM, rowI_C = ti * TM;
        if (a.pmap) { rowI_P = a.pmap[ti]; rowJ_P = a.pmap[tj]; }
        if (a.skew) {
            const bool ext = ti >= a.ext0;                          // a row under the matrix
            const int under = a.skew * TILE + (ti - a.ext0) * TM;
            rowI_P = ext ? under : ti * TM - TILE * a.kblk;
            rowJ_P = tj * TM - TILE * a.kblk;
            rowI_C = ext ? under : ti * TM - TILE * ((tj * TM) / TILE);
        }
        const double *gI = a.P + (size_t)(rowI_P + rg) + (size_t)kc * a.ldp;
        const double *gJ = a.P + (size_t)(rowJ_P + rg) + (size_t)kc * a.ldp;
        d2 stI[RPT / 2], stJ[RPT / 2];

        // (Loading C into the accumulators at the START of the tile and subtracting in the MFMA -- neg:[1,0,0] -- so that the
        // tile ends with stores only was measured in round 3 on one box against this form: SLOWER, the 39 trailing launches
        // 6.65 -> 6.95 ms; sixteen more loads in flight at the start of every tile cost more than the epilogue's round trips.)
        double *Cb = a.C + (size_t)(rowI_C + (TM / 2) * wi) + (size_t)(tj * TM + (TM / WY) * wj) * a.ldc;
        const unsigned ldcb = 8u * (unsigned)a.ldc;                                          // bytes, all of these
        const unsigned cvo = 8u * (unsigned)(lane & 15) + (unsigned)(lane >> 4) * ldcb;
        d4 acc[NB][NBY];
#pragma unroll
        for (int x = 0; x < NB; ++x)
#pragma unroll
            for (int y = 0; y < NBY; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};

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
            const double *bJ = &sJ[cur][ro + (TM / WY) * wj];
#pragma unroll
            for (int s = 0; s < KC / 4; ++s) {
                double pi_[NB], pj_[NBY];
#pragma unroll
                for (int x = 0; x < NB; ++x) pi_[x] = bI[s * 4 * LDT + 16 * x];
#pragma unroll
                for (int y = 0; y < NBY; ++y) pj_[y] = bJ[s * 4 * LDT + 16 * y];
#pragma unroll
                for (int x = 0; x < NB; ++x)
#pragma unroll
                    for (int y = 0; y < NBY; ++y) acc[x][y] = MFMA64(pj_[y], pi_[x], acc[x][y]);
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
        // next tile: asked for now (not earlier: a tile reserved while another is being worked on is a tile
        // an idle workgroup cannot take at the end of the launch), read after the stores that hide the round trip
        unsigned Lnext = 0;
        if (a.queue && t2 == 0)
            Lnext = gridDim.x + __hip_atomic_fetch_add(a.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // C -= acc, one accumulator block (4 elements) at a time: loads first, then the stores (written as
        // `*p -= acc` the compiler must assume that a store aliases the next load and serialises the memory
        // round trips; all sixteen at once would cost the 8th wave per SIMD in registers)
        unsigned cve = cvo;
        asm volatile("" : "+v"(cve));
#pragma unroll
        for (int x = 0; x < NB; ++x)
#pragma unroll
            for (int y = 0; y < NBY; ++y) {
                d4 cv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double *cp = (const double *)((const char *)(Cb + 16 * x) + (cve + (unsigned)(16 * y + 4 * r) * ldcb));
                    cv[r] = a.c_wt ? load_wt(cp) : *cp;
                }
                cv -= acc[x][y];
                if (wt_wg) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        store_wt((double *)((char *)(Cb + 16 * x) + (cve + (unsigned)(16 * y + 4 * r) * ldcb)), cv[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *(double *)((char *)(Cb + 16 * x) + (cve + (unsigned)(16 * y + 4 * r) * ldcb)) = cv[r];
                }
            }
        if (wt_wg) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t2 == 0 && sig_wg) signal_add(a.sig + a.sig_tile + sig_Ti);
        }
        if (!a.queue) break;
        if (t2 == 0) *share = Lnext;
        __syncthreads();
        L = (unsigned)__builtin_amdgcn_readfirstlane((int)*share);     // uniform: keep the tile arithmetic scalar
        if (L >= a.ntiles) break;
        // (the next write to the word comes after the barriers of the next tile, which no wave passes before
        // all have read it here)
    }
}

// ---------------------------------------------------------------------------
// The dependency-driven factorisation behind the first panel: ONE persistent launch for all trailing updates AND all
// panels (round 4).  Until round 3 every block step was update launch | solve | in-panel update | solve, four kernel
// boundaries at which the chip drains (42 us per update launch, 39 of them) and between which ~150 latency-bound
// workgroups of one wave per SIMD have the chip to themselves (56 us per block).  Here the whole thing is a list of
// 64 x 64 TILE TASKS of one and the same form
//        acc(64 x 64) = sum_{k < K} I(i, k) J(j, k)            then   C -= acc   or   X = acc,
// drawn in order from one counter by as many workgroups as the chip holds, each task waiting (bounded) for the words
// that say its inputs are complete.  Per update step s (panel s applied to the trailing matrix), in this order:
//   near tiles   the update tiles in the next panel's columns, the ten inside its diagonal block first -- they feed the
//                engine (as before) and the panel tasks;
//   far tiles    `lead` of them, then
//   panel tasks  of panel s + 1, six per 64-row strip below its diagonal block: with W = L^-1 of the two diagonal tiles
//                from the engine (potrf_engine_kernel<true>)
//                   T1 (two 64-column halves)  X0 = B0 W(t)^T         K = 64 (h + 1): W is lower triangular
//                   T2 (two halves)            B1 -= X0 X(t+1,t)^T    K = 128, X(t+1,t) is the engine's
//                   T3 (two halves)            X1 = B1 W(t+1)^T
//                no dependent chain inside a task, the update tile's inner loop, footprint and occupancy;
//   far tiles    the rest.
// Every dependency points to an EARLIER task of the list (or to the engine, which depends on earlier tasks only), and a
// task is only drawn by a resident workgroup: the earliest unfinished task can always run -- no deadlock, whatever the
// placement.  What waits for what:
//   update tile (s; i, j)   before its product: the strips i and j of panel s complete (pdone >= need; step 0: formed by
//                           earlier kernels); before its epilogue: its own tile through step s - 1 (tdone >= s);
//   T1 (strip i)            out[t] (tile t factored, W(t) published), the two near tiles of row i in tile column t;
//   T2                      both T1 of the strip, xr[t] (X(t+1,t) published), its C tile through step s (before the epilogue);
//   T3                      both T2 of the strip, out[t+1].
// Memory (per-XCD L2s are not coherent inside a launch):  every version of a C tile -- update results, B0, B1 -- is
// written write-through and read with L2-bypassing loads (sc1), by whoever touches it (measured: as fast as cached
// accesses, each byte is used once);  everything that is read MANY times -- the finished panels X0 | X1, X(t+1,t), the
// tile inverses -- is written write-through into memory NO ONE has read since the launch began (a second buffer P shaped
// like the matrix; one W per tile), so plain cached loads cannot find a stale line.  The factor therefore ends up split:
// diagonal blocks in A, everything below them in P (from panel 1 on): launch_finalize takes both.
struct DagStep {
    unsigned base;          // index of the step's first task
    unsigned near;          // near tiles: the first `near` tiles of the trapezoid's column-major order
    unsigned tpos, nT;      // position inside the step of the first panel task, panel tasks
    int H, W;               // trapezoid of the update in 64-tiles: H rows from tile row tj0 down, W columns from tj0
    int tj0;                // = 2 t: first 64-row = first 64-column of the trapezoid (t = first tile of the next block)
    int k0, K;              // the panel: columns [k0, k0 + K)
    int nstrip, two;        // next panel: 64-row strips below its diagonal block; the block has a second tile
    int need;               // pdone count at which a strip of THIS step's panel is complete (step 0: unused)
    int nd_next;            // "early half" tasks in this step's list: the diagonal-block tiles of the NEXT step, first 128 panel columns
    int split;              // this step's diagonal-block tiles only take the last 128 panel columns and add the early half's result
    unsigned p2, p3;        // positions inside the step of the T2 block and of the T3 block (T1 and the early halves sit at tpos):
                            // each group is placed where the chip gets to it about when the engine publishes what it waits for
};

struct DagArgs {
    double *A; size_t lda;
    double *P;                       // second buffer, indexed like A
    const double *Wt;                // tile inverses: Wt + t * 128 * 128
    const DagStep *steps; int nsteps;
    unsigned ntasks;
    unsigned *queue;                 // task counter (zero at launch)
    unsigned *tdone;                 // per 64-tile (i >= j) at i (i + 1) / 2 + j: update steps applied
    unsigned *pdone; int pstride;    // per panel p and strip: panel tasks finished
    unsigned *pall;                  // per panel p: strips complete (a workgroup that has seen pall[p] = all of them asks no more)
    double *partbuf;                 // early halves of the diagonal-block tiles: 2 (step parity) x 16 tiles x 64 x 64 doubles
    unsigned *dcount;                // per step and diagonal-block tile (16 words per step): early halves finished
    unsigned *sig;                   // the engine's in[] words
    unsigned *out, *xr;              // the engine's out[] / xr[] words
    unsigned *abort_word;
    const unsigned *alive;           // the engine's alive word: 1 + the XCD it runs on
    unsigned xcc_quota;              // workgroups that take part on that XCD (0: all)
    unsigned *hw;                    // diagnostics (may be null): per task hw_where() when drawn and when stored
    unsigned long long *trace;       // diagnostics (may be null): per task 4 stamps of the 100 MHz clock -- drawn, inputs
                                     // complete, product done and previous C version there, stored and signalled
    const unsigned *ftab;            // which tile a FAR tile task of a step is (dag_build_steps): ftab[s] = offset of step s's table,
                                     // ftab[ftab[s] + q] = (ti - tj0) << 16 | (tj - tj0) for the q-th far tile of the step's list;
                                     // null: the column-major order of rounds 4-5, decoded arithmetically
    unsigned xcd_g;                  // > 0: tasks are dealt to the XCDs in chunks of 2^xcd_g list positions (see dag_draw); 0: one counter
    unsigned *xcnt;                  // the XCDs' task counters, one cache line each (xcnt + 32 x; zero at launch)
};

// XCD-aware task order (round 6).  Until round 5 every workgroup drew its next task off ONE counter: which XCD worked on which
// tile was arbitrary, the ~255 tiles an XCD had in flight were a random eighth of a 2000-tile window of the column-major order
// -- 14 tile columns x 150 rows: 160 operand strips of 128 KB, five times the XCD's 4 MB L2 --, and the counters showed it: 10.7
// GB fetched per launch against 4.7 GB of C tiles, L2 hit rate 0.51 (profiles/r05_dag_kernel_hbm_traffic.json).  Now
//   (1) list position L belongs to XCD (L >> g) & 7: a workgroup on XCD x draws c off its XCD's OWN counter and takes position
//       ((c >> g) << (g + 3)) | (x << g) | (c & (2^g - 1)) -- chunks of 2^g = 32 consecutive positions per XCD, eight chunks per
//       round of 256.  One atomic per draw, as before; every XCD walks the ONE list in order (its own positions), so the
//       argument that no placement can deadlock stands: the lowest unfinished task belongs to some XCD, whose resident
//       workgroups hold only lower positions of its class and draw it next.  The draw order of chain-critical tasks moves by
//       at most a chunk's worth (~1 us of chip time);
//   (2) the host deals the far tiles of a step to those positions so that what ONE XCD draws in a row is one compact block of
//       tiles (dag_build_far_table: 16 x 16 = 256 tiles, about what an XCD has in flight: 32 operand strips, 4 MB);
//   (3) WORK SHARING is kept: the classes are consumed at the pace of their XCDs, which differ (the engine's XCD contributes
//       208 workgroups, the others 255; an XCD further from the counters' memory draws more slowly) -- every draw also reads ONE
//       other XCD's counter, and a workgroup that finds that class more than DAG_XCD_LAG draws behind its own draws from THERE
//       until it has caught up (it then looks at its own class's counter).  Any workgroup may draw from any class: a position
//       is handed out exactly once either way.  Without it
//       the launch ran at the pace of its slowest XCD: matrix pipe busy 0.73 instead of 0.79 (profiles/r06_xcd_order_counters.txt).
// Two things this cost a day to learn, both about the COUNTERS rather than the order:
//   * the eight counters must not share a cache line with each other or with anything that is LOADED while they are being
//     incremented: atomics and L2-bypassing loads on one line at ~30 + 30 per microsecond made every memory operation of the
//     launch about twice as slow (the whole launch 12.9 instead of 6.1 ms, whatever the order: even one counter drawn through
//     that line) -- each counter now owns a line (xcnt + 32 x);
//   * the look at the other XCD's counter must not wait for the draw: both are issued together with the first poll of the C
//       tile's previous version and waited for once -- a draw that costs two round trips instead of one costs 2.5 % of the launch.
// Measured (the launch replayed alone under the counters, same file): fetched 10.69 -> 5.64 GB per launch, L2 hit rate 0.51 ->
// 0.70, clock under the power limit 1.89 -> 2.02 GHz; alternated with the one counter in one process: +1.5 ... +2.6 % evaluations/s
// at n = 10^4 (four processes on one box; +2.5 / +4.6 and +0.4 on two others).  The values do not depend on it (bit-identical:
// tests/test_gpu_dag.py::test_dag_xcd_aware_order_same_bits).  COCONS_DAG_XCD=0 / COCONS_DAG_ORDER=0: rounds 4-5.
constexpr unsigned DAG_XCD_LAG = 64u;
__device__ __forceinline__ unsigned dag_position(unsigned c, unsigned cls, unsigned g)
{
    return ((c >> g) << (g + 3u)) | (cls << g) | (c & ((1u << g) - 1u));
}

// a bounded wait of dag_kernel: like wait_ge<false>, and when it runs out the waiter leaves a record in words 8 .. 13 of the
// task-word block (a.queue + 8): task, code, index of the word it waited for, value needed, value seen, wall ticks, polls
__device__ __forceinline__ bool dag_wait(const DagArgs &a, unsigned *word, unsigned need, unsigned code, unsigned L)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned it = 0;; ++it) {
        const unsigned v = poll_word(word, it);
        if (v >= need) break;
        if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if (it > (unsigned)(ENGINE_TIMEOUT_TICKS / 100ull) ||
            ((it & 1023u) == 1023u && __builtin_amdgcn_s_memrealtime() - t0 > WALL_BACKSTOP_TICKS)) {
            if (__hip_atomic_exchange(a.abort_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                unsigned *d = a.queue + 8;
                d[0] = L; d[1] = code; d[2] = (unsigned)(word - a.queue); d[3] = need; d[4] = v;
                d[5] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0);      // ticks of 10 ns this wait lasted
                d[6] = it;
            }
            return false;
        }
        __builtin_amdgcn_s_sleep(16);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // compiler ordering only
    return true;
}

__global__ void __launch_bounds__(256, 8)
dag_kernel(DagArgs a)
{
    constexpr int TM = 64, KC = 8, LDT = TM + 16, TPC = 256 / KC, RPT = TM / TPC;
    static_assert(RPT == 2, "one 16-byte load per thread, side and chunk");
    typedef double d2 __attribute__((ext_vector_type(2)));
    __shared__ double sI[2][KC * LDT];
    __shared__ double sJ[2][KC * LDT];
    unsigned *share = (unsigned *)&sI[0][TM];          // (padding of the first staged column: see update_kernel)
    const int tid = threadIdx.x;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const DagStep *__restrict__ steps = a.steps;
    const unsigned ldab = 8u * (unsigned)a.lda;
    int known = 0;                                     // panels 0 .. known are known to be complete (scalar, per workgroup)
    // EVERY task comes off the counter, the first one too (update_kernel hands out the first tile by block index: 2040
    // workgroups asking one word at the same instant cost ~8 us -- once per factorisation here, not per launch): whatever
    // part of the grid the chip holds at a given moment (another process may own CUs) then works on the LOWEST undrawn
    // tasks, and every dependency points downwards in the order -- no placement can deadlock
    //
    // The XCD that also hosts the engine is kept LESS THAN FULL: of the workgroups that land there only the first
    // `xcc_quota` take part, the others leave at once.  Why: the driver now and then pauses all queues of a process in
    // mid-kernel -- every wave is saved and restored about 0.9 ms later, on other CUs than before (on this pool about once
    // per 4000 evaluations; the stamps of such an evaluation show every workgroup standing still for that long and coming
    // back somewhere else).  An XCD that runs kernels of TWO queues does not hold eight of these workgroups on every CU
    // (227 instead of 248 beside the engine's CU; which CUs take seven depends on where the engine sits), so after a
    // restore that moved the engine the same set of workgroups need not fit again: one or two stay saved until
    // somebody leaves -- holding tasks everything else waits for, so nobody leaves until the bounded waits give up (round 4's
    // soak runs: 1 evaluation in 7000 repeated on the plain schedule; tools/dag_abort.py shows the picture).  With the
    // quota that XCD has a few CUs' worth of room in every placement, and no workgroup of the grid is ever waiting in the
    // dispatcher for a slot there (the others XCDs hold their 255 at eight per CU, before and after).
    const unsigned myx = (hw_where() >> 28) & 7u;       // (the XCD this workgroup runs on: a save / restore keeps it there)
    const bool loyal = ((blockIdx.x >> 3) & 1u) == 0u;  // (XCD-aware order: this workgroup never draws from another XCD's class)
    if (tid == 0) {
        bool take = true;
        if (a.xcc_quota) {
            // (alive[16 + x]: workgroups of the engine's launch -- the engine, its partner -- resident on XCD x; every one of
            // them takes a CU, and an XCD that runs two queues holds seven of these workgroups per CU rather than eight)
            const unsigned c = __hip_atomic_load(a.alive + 16 + myx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c) {
                const unsigned less = 7u * (c - 1u);
                const unsigned quota = a.xcc_quota > less + 8u ? a.xcc_quota - less : 8u;
                take = __hip_atomic_fetch_add(a.queue + 16 + myx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < quota;
            }
        }
        unsigned first = 0xffffffffu;
        if (take) {
            if (a.xcd_g) first = dag_position(__hip_atomic_fetch_add(a.xcnt + 32u * myx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), myx, a.xcd_g);
            else first = __hip_atomic_fetch_add(a.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        share[1] = first;
        share[2] = 0u;                                  // (XCD-aware order: 1 + the class the NEXT draw helps out; 0: its own)
    }
    __syncthreads();
    unsigned L = (unsigned)__builtin_amdgcn_readfirstlane((int)share[1]);
    __syncthreads();
    if (L >= a.ntasks) return;
    for (;;) {
        // ---- which task: all of this is wave-uniform and stays on the scalar unit
        int lo = 0, hi = a.nsteps - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (steps[mid].base <= L) lo = mid; else hi = mid - 1;
        }
        const int s = lo;
        const DagStep st = steps[s];
        const unsigned q = L - st.base;
        if (a.trace && tid == 0) {
            a.trace[4 * (size_t)L] = __builtin_amdgcn_s_memrealtime();
            if (a.hw) a.hw[2 * (size_t)L] = hw_where();
        }
        const int t = st.tj0 >> 1;                           // first 128-tile of the next block
        // where in the step's list: tiles | T1, early halves | tiles | T2 | tiles | T3 | tiles
        const unsigned per_u = 2u * (unsigned)st.nstrip, nA = per_u + (unsigned)st.nd_next, perBC = st.two ? per_u : 0u;
        int tkind = -1;                                      // -1: update tile; 0, 1, 2: T1, T2, T3; 3: early half
        unsigned tu = 0, qt_u = q;                           // index inside its group; index among the step's update tiles
        if (q >= st.tpos) {
            if (q < st.tpos + nA) { tu = q - st.tpos; tkind = tu < per_u ? 0 : 3; }
            else if (q < st.p2) qt_u = q - nA;
            else if (q < st.p2 + perBC) { tu = q - st.p2; tkind = 1; }
            else if (q < st.p3) qt_u = q - nA - perBC;
            else if (q < st.p3 + perBC) { tu = q - st.p3; tkind = 2; }
            else qt_u = q - nA - 2u * perBC;
        }
        const bool isT = tkind >= 0;
        const double *gIb, *gJb;
        unsigned ldib, ldjb, ldob = ldab;                    // leading dimensions in bytes (operands, output)
        int K, i_wt = 0, store_only = 0, strip_task = 0;
        double *Cb;                                          // the task's 64 x 64 output tile
        const double *pa = nullptr;                          // finisher of a split diagonal-block tile: the early half's result
        unsigned *w0 = nullptr, *w1 = nullptr, *w2 = nullptr, *we = nullptr, *we2 = nullptr, *dn;
        unsigned n0 = 0, n1 = 0, n2 = 0, ne = 0, dval = 0;   // dval != 0: raise *dn to dval; 0: add 1
        int sigT = -1;                                       // >= 0: the tile lies in the next diagonal block: raise sig[sigT]
        int prio = 0;
        if (!isT) {
            const int qt = (int)qt_u;
            int ti, tj;
            if (a.ftab && qt >= (int)st.near) {
                // a far tile: the host's table says which (XCD-aware order: what one XCD draws in a row is a compact block)
                const unsigned e = a.ftab[a.ftab[s] + (unsigned)(qt - (int)st.near)];
                tj = st.tj0 + (int)(e & 0xffffu); ti = st.tj0 + (int)(e >> 16);
            } else {
                int jl = 0, jh = st.W - 1;
                while (jl < jh) {
                    const int mid = (jl + jh + 1) >> 1;
                    if (mid * st.H - mid * (mid - 1) / 2 <= qt) jl = mid; else jh = mid - 1;
                }
                tj = st.tj0 + jl; ti = tj + (qt - (jl * st.H - jl * (jl - 1) / 2));
            }
            const double *Ps = s == 0 ? a.A : a.P;
            gIb = Ps + (size_t)ti * TM + (size_t)st.k0 * a.lda;
            gJb = Ps + (size_t)tj * TM + (size_t)st.k0 * a.lda;
            ldib = ldab; ldjb = ldab; K = st.K;
            Cb = a.A + (size_t)ti * TM + (size_t)tj * TM * a.lda;
            if (s > known) {
                // (first the word that says the WHOLE panel is there: in the head of the factorisation it nearly always is, and
                // then this workgroup asks nothing more for the rest of the step -- two polls per tile were 5 % of its time)
                w0 = a.pall + s; n0 = (unsigned)st.H;
                w1 = a.pdone + (size_t)s * a.pstride + (ti - st.tj0); n1 = (unsigned)st.need;
                w2 = a.pdone + (size_t)s * a.pstride + (tj - st.tj0); n2 = (unsigned)st.need;
            }
            we = a.tdone + (ti * (ti + 1) / 2 + tj); ne = (unsigned)s;
            dn = we; dval = (unsigned)s + 1u;
            const int Ti = (ti >> 1) - t, Tj = (tj >> 1) - t;
            if (Ti >= 0 && Ti <= st.two && Tj >= 0 && Tj <= Ti) {     // (a last block of ONE tile: rows of right-hand sides may lie below it)
                sigT = t + Ti; prio = 3;
                if (st.split) {
                    // One of the ten tiles the next diagonal block waits for.  Under load a tile's product is memory latency
                    // per 8-column chunk (60 us for 256 columns, whatever its priority), and these ten sit on the chain between
                    // two diagonal blocks: their FIRST 128 panel columns (X0: complete as soon as T1 is, long before the
                    // engine has finished the block) were multiplied by an "early half" task in the previous step's list;
                    // this task takes the last 128 columns and subtracts both halves in a fixed order.
                    const int ta = ti - st.tj0, tb = tj - st.tj0, dd = ta * (ta + 1) / 2 + tb;
                    gIb += (size_t)TILE * a.lda; gJb += (size_t)TILE * a.lda; K = TILE;
                    pa = a.partbuf + ((size_t)(s & 1) * 16 + dd) * (TM * TM);
                    we2 = a.dcount + (size_t)s * 16 + dd;
                }
            }
        } else {
            const int per = 2 * st.nstrip;
            const size_t c_t = (size_t)t * TILE, c_t1 = c_t + TILE;
            prio = 2;
            const int stage = tkind;     // (3: early half of a diagonal-block tile of the NEXT step)
            int strip = 0, h = 0, row64 = 0;
            if (stage != 3) {
                strip = (int)(tu >> 1); h = (int)(tu & 1u);
                row64 = st.tj0 + (st.two ? 4 : 2) + strip;
                strip_task = 1;
            }
            dn = a.pdone + (size_t)(s + 1) * a.pstride + strip;
            if (stage == 3) {
                const int dd = (int)tu - per;
                const int ta = c_tri_ib[dd], tb = dd - ta * (ta + 1) / 2;
                const int tj0n = st.tj0 + 4;                           // the next step's trapezoid (this block has two tiles)
                const size_t k0n = (size_t)(s + 1) * 2 * TILE;         // its panel's first column
                gIb = a.P + (size_t)(tj0n + ta) * TM + k0n * a.lda; ldib = ldab;
                gJb = a.P + (size_t)(tj0n + tb) * TM + k0n * a.lda; ldjb = ldab;
                K = TILE;
                Cb = a.partbuf + ((size_t)((s + 1) & 1) * 16 + dd) * (TM * TM); store_only = 1; ldob = 8u * TM;
                w0 = a.pdone + (size_t)(s + 1) * a.pstride + ta; n0 = 2u;      // both T1 of the two strips: X0 complete
                w1 = a.pdone + (size_t)(s + 1) * a.pstride + tb; n1 = 2u;
                dn = a.dcount + (size_t)(s + 1) * 16 + dd;
            } else if (stage == 0) {     // T1: X0(:, half h of tile t) = B0 W(t)^T
                gIb = a.A + (size_t)row64 * TM + c_t * a.lda; ldib = ldab; i_wt = 1;
                gJb = a.Wt + (size_t)t * TILE * TILE + TM * h; ldjb = 8u * TILE;
                K = TM * (h + 1);
                Cb = a.P + (size_t)row64 * TM + (c_t + TM * h) * a.lda; store_only = 1;
                w0 = a.out + t; n0 = 1u;
                w1 = a.tdone + (row64 * (row64 + 1) / 2 + 2 * t); n1 = (unsigned)s + 1u;
                w2 = w1 + 1; n2 = n1;
            } else if (stage == 1) {     // T2: B1(:, half h of tile t + 1) -= X0 X(t+1,t)^T
                gIb = a.P + (size_t)row64 * TM + c_t * a.lda; ldib = ldab;
                gJb = a.P + (c_t1 + TM * h) + c_t * a.lda; ldjb = ldab;
                K = TILE;
                Cb = a.A + (size_t)row64 * TM + (c_t1 + TM * h) * a.lda;
                w0 = dn; n0 = 2u;
                w1 = a.xr + t; n1 = 1u;
                we = a.tdone + (row64 * (row64 + 1) / 2 + 2 * (t + 1) + h); ne = (unsigned)s + 1u;
            } else {                     // T3: X1(:, half h of tile t + 1) = B1 W(t+1)^T
                gIb = a.A + (size_t)row64 * TM + c_t1 * a.lda; ldib = ldab; i_wt = 1;
                gJb = a.Wt + (size_t)(t + 1) * TILE * TILE + TM * h; ldjb = 8u * TILE;
                K = TM * (h + 1);
                Cb = a.P + (size_t)row64 * TM + (c_t1 + TM * h) * a.lda; store_only = 1;
                w0 = dn; n0 = 4u;
                w1 = a.out + t + 1; n1 = 1u;
            }
        }
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        else if (prio == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(0);

        // ---- inputs of the product complete?  (one lane; the words are nearly always there already)
        if (w0) {
            if (tid == 0) {
                unsigned ok;
                if (!isT) {
                    // update tile: whole panel there (2)?  else its two strips (1)
                    if (__hip_atomic_load(w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n0) ok = 2u;
                    else ok = (dag_wait(a, w1, n1, abort_code(0xb00u, (unsigned)s), L) &&
                               dag_wait(a, w2, n2, abort_code(0xc00u, (unsigned)s), L)) ? 1u : 0u;
                } else {
                    bool o = dag_wait(a, w0, n0, abort_code(0xa00u, (unsigned)s), L);
                    if (o && w1) o = dag_wait(a, w1, n1, abort_code(0xb00u, (unsigned)s), L);
                    if (o && w2) o = dag_wait(a, w2, n2, abort_code(0xc00u, (unsigned)s), L);
                    ok = o ? 1u : 0u;
                }
                *share = ok;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned ok = (unsigned)__builtin_amdgcn_readfirstlane((int)*share);
            __syncthreads();
            if (!ok) return;
            if (ok == 2u) known = s;
        }

        if (a.trace && tid == 0) a.trace[4 * (size_t)L + 1] = __builtin_amdgcn_s_memrealtime();
        // per-thread offsets from a laundered lane index (see update_kernel: hoisted, they spill)
        int lane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        const int wave = wave_s, t2 = 64 * wave + lane;
        const int wi = wave & 1, wj = wave >> 1;
        const int kc = t2 / TPC, rg = (t2 % TPC) * RPT;
        const int ro = (lane >> 4) * LDT + (lane & 15);
        const char *pI = (const char *)gIb + (8u * (unsigned)rg + (unsigned)kc * ldib);
        const char *pJ = (const char *)gJb + (8u * (unsigned)rg + (unsigned)kc * ldjb);
        const unsigned cIb = (unsigned)KC * ldib, cJb = (unsigned)KC * ldjb;
        const int nch = K / KC;
        d2 stI, stJ;
        d4 acc[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
        if (i_wt) { stI[0] = load_wt((const double *)pI); stI[1] = load_wt((const double *)pI + 1); }
        else stI = *(const d2 *)pI;
        stJ = *(const d2 *)pJ;
        *(d2 *)(&sI[0][kc * LDT + rg]) = stI;
        *(d2 *)(&sJ[0][kc * LDT + rg]) = stJ;
        __syncthreads();
        for (int ch = 0; ch < nch; ++ch) {
            const int cur = ch & 1;
            if (ch + 1 < nch) {
                const char *qI = pI + (unsigned)(ch + 1) * cIb;
                const char *qJ = pJ + (unsigned)(ch + 1) * cJb;
                if (i_wt) { stI[0] = load_wt((const double *)qI); stI[1] = load_wt((const double *)qI + 1); }
                else stI = *(const d2 *)qI;
                stJ = *(const d2 *)qJ;
            }
            const double *bI = &sI[cur][ro + (TM / 2) * wi];
            const double *bJ = &sJ[cur][ro + (TM / 2) * wj];
#pragma unroll
            for (int k4 = 0; k4 < KC / 4; ++k4) {
                const double p0 = bI[k4 * 4 * LDT], p1 = bI[k4 * 4 * LDT + 16];
                const double q0 = bJ[k4 * 4 * LDT], q1 = bJ[k4 * 4 * LDT + 16];
                acc[0][0] = MFMA64(q0, p0, acc[0][0]);
                acc[0][1] = MFMA64(q1, p0, acc[0][1]);
                acc[1][0] = MFMA64(q0, p1, acc[1][0]);
                acc[1][1] = MFMA64(q1, p1, acc[1][1]);
            }
            if (ch + 1 < nch) {
                *(d2 *)(&sI[cur ^ 1][kc * LDT + rg]) = stI;
                *(d2 *)(&sJ[cur ^ 1][kc * LDT + rg]) = stJ;
            }
            __syncthreads();
        }
        // ---- the next task is asked for now (see update_kernel), and the C tile's previous version must be there
        if (t2 == 0) {
            // (the draw, the look at another XCD's counter and the first poll of `we` are issued back to back and waited for
            // together: one round trip, as with the one counter of rounds 4-5)
            unsigned Ln, c = 0u, cy = 0u, cls = 0u, y = 0u;
            unsigned help = 0u;
            if (a.xcd_g) {
                // this XCD's next list position -- or, while another class is behind, that class's.  The look goes to one of the
                // other seven classes in turn, and to this XCD's own while it helps another
                help = loyal ? 0u : share[2];
                cls = help ? help - 1u : myx;
                y = help ? myx : ((myx + 1u + L % 7u) & 7u);
                c = __hip_atomic_fetch_add(a.xcnt + 32u * cls, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cy = __hip_atomic_load(a.xcnt + 32u * y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                Ln = 0u;
            } else
                Ln = __hip_atomic_fetch_add(a.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (in flight during the poll)
            unsigned ok = 1u;
            if (we) ok = dag_wait(a, we, ne, abort_code(0xd00u, (unsigned)s), L) ? 1u : 0u;
            if (ok && we2) ok = wait_ge<false>(we2, 1u, a.abort_word, abort_code(0xe00u, (unsigned)s)) ? 1u : 0u;   // (no record: a second dag_wait costs the kernel its scratch-free allocation)
            if (a.xcd_g) {
                Ln = dag_position(c, cls, a.xcd_g);
                // (a class is behind when its counter is: all classes hold the same share of every stretch of the list; a class
                // that has run off the end of the list needs no help.  Help STICKS: a workgroup that has found a class more than
                // DAG_XCD_LAG draws behind its own keeps drawing from it until it is within half of that -- an XCD with few
                // workgroups (another process's kernels on its CUs; the quota test's eight) is then carried by all the others,
                // and no class ever runs a step ahead of another, which is what keeps the dependency waits short)
                // LOYAL workgroups never help: every other workgroup of an XCD (by block index) draws from its own class only.
                // Without them the argument of (1) fails -- all workgroups of class k may be away in other classes' tasks, blocked
                // at dependencies on class k's undrawn head, with nobody left to draw it: a deadlock the bounded waits end, seen
                // once in 300 evaluations of tools/diag/quota_stress.py (task 20592 waited 63 ms for a strip whose tasks nobody
                // had drawn).  A loyal workgroup holds only positions of its own class below the class's head; if that head is the
                // lowest unfinished task everything below it is finished, so the loyal workgroup is free and draws it.
                if (!help) share[2] = (!loyal && c > cy + DAG_XCD_LAG && dag_position(cy, y, a.xcd_g) < a.ntasks) ? y + 1u : 0u;
                else {
                    share[2] = (c + DAG_XCD_LAG / 2u < cy && dag_position(c + 1u, cls, a.xcd_g) < a.ntasks) ? help : 0u;
                    if (Ln >= a.ntasks) {            // (the helped class has run off the end: back to this XCD's own)
                        cls = myx;
                        c = __hip_atomic_fetch_add(a.xcnt + 32u * cls, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        Ln = dag_position(c, cls, a.xcd_g);
                    }
                }
                if (Ln >= a.ntasks && cls == myx) {
                    // this XCD's positions are used up: the tail of the list belongs to whoever has workgroups left
                    for (unsigned d = 1; d < 8u && Ln >= a.ntasks; ++d) {
                        const unsigned z = (myx + d) & 7u;
                        if (dag_position(__hip_atomic_load(a.xcnt + 32u * z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), z, a.xcd_g) < a.ntasks)
                            Ln = dag_position(__hip_atomic_fetch_add(a.xcnt + 32u * z, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), z, a.xcd_g);
                    }
                }
            }
            share[0] = ok; share[1] = Ln;
            if (a.trace) a.trace[4 * (size_t)L + 2] = __builtin_amdgcn_s_memrealtime();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned okE = share[0];
        const unsigned Lnext = share[1];
        if (!okE) return;
        // ---- epilogue: the wave's 32 x 32 part of the tile, one accumulator block at a time (loads first, then stores)
        double *Cw = (double *)((char *)(Cb + (TM / 2) * wi) + (size_t)((TM / 2) * wj) * ldob);
        unsigned cve = 8u * (unsigned)(lane & 15) + (unsigned)(lane >> 4) * ldob;
        asm volatile("" : "+v"(cve));
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                d4 cv;
                if (store_only) cv = acc[x][y];
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cv[r] = load_wt((const double *)((const char *)(Cw + 16 * x) + (cve + (unsigned)(16 * y + 4 * r) * ldob)));
                    if (pa) {
                        // (offsets re-derived here from the laundered output offset: nothing extra lives through the product)
                        const double *Pw = pa + (TM / 2) * wi + (TM / 2) * wj * TM;
                        int ln;
                        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                        const unsigned cvp = 8u * (unsigned)(ln & 15) + (unsigned)(ln >> 4) * (8u * TM);
                        d4 pv;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            pv[r] = load_wt((const double *)((const char *)(Pw + 16 * x) + (cvp + (unsigned)(16 * y + 4 * r) * (8u * TM))));
                        cv -= pv + acc[x][y];                 // (early half + late half, then off C: one fixed order)
                    } else cv -= acc[x][y];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    store_wt((double *)((char *)(Cw + 16 * x) + (cve + (unsigned)(16 * y + 4 * r) * ldob)), cv[r]);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t2 == 0) {
            if (sigT >= 0) signal_add(a.sig + sigT);
            // (every word another workgroup polls is updated with a read-modify-write atomic, which executes at the memory side.
            // Round 4's first version published tdone with a write-through STORE: such a store leaves the line valid in the
            // storing XCD's L2, and a workgroup of that XCD polling a NEIGHBOURING word of the line -- another tile's -- could
            // then read a stale value for as long as it kept polling: 2 time-outs in 16 000 evaluations, caught with
            // dag_wait's record: "waited for word 9646 >= 2, saw 1, holds 2 now".)
            if (dval) __hip_atomic_fetch_max(dn, dval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (__hip_atomic_fetch_add(dn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (st.two ? 6u : 2u) && strip_task)
                __hip_atomic_fetch_add(a.pall + s + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the strip is complete
            if (a.trace) {
                a.trace[4 * (size_t)L + 3] = __builtin_amdgcn_s_memrealtime();
                if (a.hw) a.hw[2 * (size_t)L + 1] = hw_where();
            }
        }
        L = (unsigned)__builtin_amdgcn_readfirstlane((int)Lnext);
        if (L >= a.ntasks) break;
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
__global__ void __launch_bounds__(1024)
finalize_kernel(const double *A, size_t lda, int c0, int c1, int n, int row0, int nr, double *out, int skew, int npad,
                const double *A2, int a2_cols)
{
    // A2 != NULL (dense layout only): the factor of the dependency-driven schedule -- the part of a column BELOW the 256 x 256
    // diagonal block it runs through lives in A2 (columns [256, a2_cols): the panels its tasks formed), everything else in A
    __shared__ double red[16];
    const int b = blockIdx.x;
    const int hi = c1 < n ? c1 : n;
    double s = 0.0;
    // every element sits in a cache line of its own (stride lda): 1024 threads with four loads in flight each -- with 256
    // threads and one load at a time the kernel was 25 us of serial round trips at n = 10^4, on the critical path of
    // every evaluation
    constexpr int U = 4;
    const int step = (int)blockDim.x;
    if (b == nr * nr) {
        for (int c = c0 + (int)threadIdx.x; c < hi; c += U * step) {
            double v[U];
#pragma unroll
            for (int q = 0; q < U; ++q) v[q] = (c + q * step < hi) ? A[band_index(c + q * step, c + q * step, lda, skew, npad)] : 1.0;
#pragma unroll
            for (int q = 0; q < U; ++q) s += log(v[q]);
        }
    } else {
        const int ra = row0 + b / nr, rb = row0 + b % nr;
        for (int c = c0 + (int)threadIdx.x; c < hi; c += U * step) {
            double va[U], vb[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const bool in = c + q * step < hi;
                const int cc = c + q * step;
                int below0 = 2 * TILE * (cc / (2 * TILE) + 1);                // first row below column cc's diagonal block
                if (npad > 0 && below0 > npad) below0 = npad;                 // (a last block of one tile)
                const bool in2 = A2 && cc >= 2 * TILE && cc < a2_cols;
                const double *Sa = (in2 && ra >= below0) ? A2 : A;
                const double *Sb = (in2 && rb >= below0) ? A2 : A;
                va[q] = in ? Sa[band_index(ra, cc, lda, skew, npad)] : 0.0;
                vb[q] = in ? Sb[band_index(rb, cc, lda, skew, npad)] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < U; ++q) s += va[q] * vb[q];
        }
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
                  double *scratch, int cchunk, int skew, int npad, const double *A2, int a2_cols)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int cb = blockIdx.y * cchunk;
    const int ce = (cb + cchunk < n) ? cb + cchunk : n;
    if (i >= m) return;
    // (A2: the factor of the dependency-driven schedule -- below the diagonal blocks the columns [256, a2_cols) live in the
    // second buffer, launch_finalize; the rows read here lie under the matrix, and a chunk of 256 columns is one panel)
    if (A2 && cb >= 2 * TILE && cb < a2_cols) A = A2;
    double s = 0.0, q = 0.0;
    for (int c = cb; c < ce; ++c) {
        double v = A[band_index(row0 + i, c, lda, skew, npad)];
        double y = A[band_index(rowy, c, lda, skew, npad)];
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
// the attribute that allows more than 64 KB of dynamic LDS is per kernel and device: set once, not per launch
static void set_dynamic_lds_once(const void *kernel, size_t bytes, std::atomic<unsigned long long> &done_mask)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && ((done_mask.load(std::memory_order_relaxed) >> dev) & 1ull)) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (dev >= 0 && dev < 64) done_mask.fetch_or(1ull << dev, std::memory_order_relaxed);
}

void launch_potrf_tile(double *A, size_t lda, int c0, double *dinv, int *info, hipStream_t s)
{
    const size_t shm = (36 + 1) * 256 * sizeof(double);   // 75,776 B of dynamic LDS (> the 64 KB default)
    static std::atomic<unsigned long long> attr_done{0};
    set_dynamic_lds_once((const void *)potrf_tile_kernel, shm, attr_done);
    hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(512), shm, s, A, lda, c0, dinv, info);
}

__global__ void __launch_bounds__(64)
raise_word_kernel(unsigned *word)
{
    if (threadIdx.x == 0) __hip_atomic_store(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

void launch_raise_word(unsigned *word, hipStream_t s) { hipLaunchKernelGGL(raise_word_kernel, dim3(1), dim3(64), 0, s, word); }

// Can a kernel on stream `first` and a kernel launched AFTER it on stream `second` run at the same time?  HIP multiplexes
// its streams onto a handful of hardware queues, and two streams that share one run strictly one kernel after the other:
// a resident engine on one of them would then block the very kernels it waits for (seen in round 4: a synchronous copy on
// the NULL stream shifted the assignment, and a handle's two streams landed on one queue).  A one-lane kernel on `first`
// waits (bounded: 2 ms) for a word that a kernel on `second` raises; words[0] = the word, words[1] = 1 if the wait ran out.
// Both streams must be idle; returns 1 (concurrent), 0 (serialised) or -1 (HIP error).
int streams_run_concurrently(hipStream_t first, hipStream_t second, unsigned *words)
{
    if (hipMemsetAsync(words, 0, 2 * sizeof(unsigned), second) != hipSuccess) return -1;
    if (hipStreamSynchronize(second) != hipSuccess) return -1;
    hipLaunchKernelGGL(engine_gate_kernel, dim3(1), dim3(64), 0, first, words, words + 1, 1u, 200000ull, 0u, (unsigned *)nullptr);
    hipLaunchKernelGGL(raise_word_kernel, dim3(1), dim3(64), 0, second, words);
    unsigned h[2] = {0, 0};
    if (hipStreamSynchronize(first) != hipSuccess || hipStreamSynchronize(second) != hipSuccess) return -1;
    if (hipMemcpyAsync(h, words, sizeof h, hipMemcpyDeviceToHost, second) != hipSuccess) return -1;
    if (hipStreamSynchronize(second) != hipSuccess) return -1;
    return h[1] == 0 ? 1 : 0;
}

void launch_engine_gate(unsigned *alive, unsigned *abort_word, hipStream_t s, bool last_tile, bool patient, int nhelp,
                        unsigned *raise_in)
{
    // last_tile: not the start-up gate (is the engine resident? 5 ms, code 0x600) but the wait of the reductions for the
    // engine's LAST diagonal tile when no panel kernel has waited for it (code 0x900, the hand-offs' 100 ms bound)
    // patient: the start-up gate of a handle's FIRST engine-schedule operation -- 50 ms instead of 5: whatever a first
    // dispatch on a fresh stream may still cost the runtime (the warm-up launch of the handle has paid what it can) must
    // not be mistaken for "every CU is taken by someone else"
    hipLaunchKernelGGL(engine_gate_kernel, dim3(1), dim3(64), 0, s, alive, abort_word, last_tile ? ABORT_LAST_TILE : ABORT_GATE,
                       last_tile ? ENGINE_TIMEOUT_TICKS : (patient ? 10 * GATE_TIMEOUT_TICKS : GATE_TIMEOUT_TICKS),
                       (unsigned)(last_tile || nhelp < 0 ? 0 : nhelp), last_tile ? nullptr : raise_in);
}

// Dynamic LDS of the engine.  136 KB: the 128 KB LDS copy of X plus the flag word; leaves room for ONE update workgroup
// beside the engine.  (Asking for all 160 KB measured 7 % slower trailing updates chip-wide while the engine was
// resident; 76 .. 152 KB did not.)  The function attribute that allows it is set once per device, not per launch.
static size_t engine_lds_bytes()
{
    static size_t shm = 0;
    if (!shm) {
        shm = 150 * 1024;       // (round 5: + the 28 blocks of the tile inverse's strips; rounds 2-4: 136 KB)
        const char *x = getenv("COCONS_ENGINE_LDS");
        if (x && (size_t)atol(x) >= shm) shm = (size_t)atol(x);
    }
    return shm;
}

// t0 >= nt: the kernel is launched all the same, raises its alive word and leaves at once -- the WARM-UP launch of a
// handle (api.hip): whatever the first dispatch of this kernel on this stream costs the runtime (queue set-up, code
// object, LDS configuration) is paid there and not inside the bounded gate of the first engine-schedule operation.
void launch_potrf_engine(double *A, size_t lda, int t0, int nt, double *dinv, int *info,
                         unsigned *in, unsigned *out, unsigned *xr, unsigned *abort_word, unsigned *alive, hipStream_t s,
                         double *wbuf, double *pbuf, int dag_until, unsigned long long *trace, double *mbox, int in_wait_ms)
{
    EngineArgs e;
    e.A = A; e.lda = lda; e.t0 = t0; e.nt = nt; e.dinv = dinv; e.info = info;
    e.in = in; e.out = out; e.xr = xr; e.abort_word = abort_word; e.alive = alive;
    e.wbuf = wbuf; e.pbuf = pbuf; e.dag_until = dag_until; e.trace = trace;
    e.in_ticks = in_wait_ms > 0 ? 100000ull * (unsigned long long)in_wait_ms : HOST_PACED_TICKS;
    // (mbox: pair mode -- the partner is workgroup 8: with the round robin over the eight XCDs, the next one on workgroup 0's XCD)
    e.mbox = mbox;
    e.partner = mbox ? 8 : 0;
    const int grid = e.partner ? e.partner + 1 : 1;
    const size_t shm = engine_lds_bytes();
    if (wbuf && pbuf) {
        static std::atomic<unsigned long long> attr_done{0};
        set_dynamic_lds_once((const void *)potrf_engine_kernel<true>, shm, attr_done);
        hipLaunchKernelGGL(potrf_engine_kernel<true>, dim3(grid), dim3(512), shm, s, e);
    } else {
        static std::atomic<unsigned long long> attr_done{0};
        set_dynamic_lds_once((const void *)potrf_engine_kernel<false>, shm, attr_done);
        hipLaunchKernelGGL(potrf_engine_kernel<false>, dim3(grid), dim3(512), shm, s, e);
    }
}

void launch_trsm_tile(double *A, size_t lda, int c0, int r0, int r1, const double *dinv, hipStream_t s,
                      unsigned *wait_word, unsigned *abort_word, int band_r1, int ext_r0, int own_world, int own_rank,
                      int own_group)
{
    // rows [r0, r1), or -- band-limited -- [r0, band_r1) and [ext_r0, r1)
    int nb1 = ((band_r1 >= 0 ? band_r1 : r1) - r0) / 64, nb2 = band_r1 >= 0 ? (r1 - ext_r0) / 64 : 0;
    if (nb1 < 0) nb1 = 0;
    if (nb2 < 0) nb2 = 0;
    if (nb1 + nb2 <= 0) return;
    hipLaunchKernelGGL(trsm_tile_kernel, dim3(nb1 + nb2), dim3(256), 0, s, A, lda, c0, r0, dinv, wait_word, abort_word,
                       nb1, ext_r0, own_world, own_rank, own_group < 1 ? 1 : own_group);
}

// rows like launch_trsm_tile: [r0, r1), or -- band-limited -- [r0, band_r1) and [ext_r0, r1); mbox: the tile's mailbox, filled
// with the pattern ~0 by the caller
void launch_potrf_follow(double *A, size_t lda, int c0, int r0, int r1, double *dinv, int *info, double *mbox,
                         unsigned *abort_word, hipStream_t s, int band_r1, int ext_r0)
{
    int nb1 = ((band_r1 >= 0 ? band_r1 : r1) - r0) / 64, nb2 = band_r1 >= 0 ? (r1 - ext_r0) / 64 : 0;
    if (nb1 < 0) nb1 = 0;
    if (nb2 < 0) nb2 = 0;
    const int nstrips = nb1 + nb2;
    static std::atomic<unsigned long long> attr_done{0};
    const size_t shm = 76 * 1024;               // the tile's image and its Q operands (74 KB); a follower's two stages: 36 KB
    set_dynamic_lds_once((const void *)potrf_follow_kernel, shm, attr_done);
    hipLaunchKernelGGL(potrf_follow_kernel, dim3(1 + (nstrips + 1) / 2), dim3(512), shm, s, A, lda, c0, dinv, info, mbox, r0, nb1,
                       ext_r0, nstrips, abort_word);
}

void launch_panel_pair(double *A, size_t lda, int c0, int r0, int r1, const double *q0, const double *q1, unsigned *out0,
                       unsigned *xr, unsigned *out1, unsigned *abort_word, hipStream_t s, const double *mb0, const double *mb1,
                       double *smb, int ndiag, unsigned *sig, int sig_tile, double *xmb)
{
    const int nb = (r1 - r0) / 64;
    if (nb <= 0) return;
    const bool split = xmb && mb0 && mb1;         // (two workgroups per strip: see the kernel)
    // (ndiag = 10 or 3: the next diagonal block -- two tiles or one -- is updated by as many extra workgroups, which follow the
    // first 4 or 2 strips through the strip mailbox smb; needs the tiles' mailboxes)
    const bool diag = smb && mb0 && mb1 && ndiag > 0 && nb >= (ndiag == 10 ? 4 : 2);
    hipLaunchKernelGGL(panel_pair_kernel, dim3((split ? 2 * nb : nb) + (diag ? ndiag : 0)), dim3(256), 0, s, A, lda, c0, r0, q0, q1,
                       out0, xr, out1, abort_word, mb0, mb1, diag ? smb : nullptr, nb, diag ? (ndiag == 10 ? 4 : 2) : 0, sig, sig_tile,
                       split ? xmb : nullptr);
}

// waves per workgroup of the trailing update (COCONS_UPD_WAVES: 4 or 8, see update_kernel's NW)
static int upd_waves = -1;
static long long upd_w8_max_tiles = -1;      // 8-wave workgroups only for launches of at most this many tiles (0 = all)
void set_update_waves(int nw) { upd_waves = nw == 8 ? 8 : 4; }
void set_update_w8_max_tiles(int ntiles) { upd_w8_max_tiles = ntiles < 0 ? 0 : ntiles; }
static int upd_c_wt = 0;
void set_update_c_wt(int on) { upd_c_wt = on ? 1 : 0; }

void launch_update_from(double *A, size_t lda, const double *P, size_t ldp, int K,
                        int ti0, int ti1, int tj0, int tj1, bool lower_only, hipStream_t s,
                        int ptiles, int world, int rank, unsigned *sig, int sig_tile,
                        unsigned *wait_word, unsigned *abort_word, unsigned *queue, int band_hi, int ext0,
                        int skew, int kblk, int trim64, const int *pmap, int skip_lo, int skip_hi)
{
    // tile rows [ti0, ti1), or -- band-limited -- [ti0, band_hi) and [ext0, ti1)
    const bool band = band_hi >= 0;
    const int rows_band = (band ? band_hi : ti1) - ti0, rows_ext = band ? ti1 - ext0 : 0;
    if (rows_band + rows_ext <= 0 || rows_band < 0 || rows_ext < 0 || tj1 <= tj0 || K <= 0) return;
    trim64 = trim64 ? 1 : 0;
    const int rows64 = 2 * (rows_band + rows_ext) - trim64;        // 64-row tiles the launch covers
    if (rows64 <= 0) return;
    if (upd_waves < 0) { const char *e = getenv("COCONS_UPD_WAVES"); set_update_waves(e ? atoi(e) : 8); }
    if (upd_w8_max_tiles < 0) { const char *e = getenv("COCONS_UPD_W8_MAX_TILES"); set_update_w8_max_tiles(e ? atoi(e) : 3500); }
    bool use_w8 = false;
    UpdArgs a;
    a.queue = nullptr; a.ntiles = 0;
    a.skew = skew; a.kblk = kblk; a.c_wt = upd_c_wt;
    a.Hb = 2 * rows_band; a.ext0 = 2 * ext0;
    a.C = A; a.ldc = lda; a.P = P; a.ldp = ldp; a.K = K;
    a.lower_only = lower_only ? 1 : 0;
    a.ptiles = ptiles < 1 ? 1 : ptiles; a.world = world; a.rank = rank; a.pmap = pmap; a.skip_lo = skip_lo; a.skip_hi = skip_hi;
    a.sig = sig; a.sig_tile = sig_tile;
    a.wait_word = wait_word; a.abort_word = abort_word;
    a.H = 0; a.W = 0;
    // 64 x 64 tiles throughout (the 128 x 128 shape measured 31 TFLOP/s against 50): tile indices in
    // units of 64 from here on
    a.ti0 = 2 * ti0; a.tj0 = 2 * tj0;
    dim3 grid(rows64, 2 * (tj1 - tj0));
    if (lower_only) {
        // requires ti0 >= tj0 == first column: the trapezoid rows tj0..ti1, columns tj0..tj1
        if (ti0 != tj0) { a.lower_only = 0; }    // strictly-below rectangle: every tile does work
        else {
            const long long H = rows64, W = 2LL * (tj1 - tj0);
            a.H = (int)H; a.W = (int)W;
            const long long total = W * H - W * (W - 1) / 2;
            static int slots = 0;
            if (!slots) {
                int dev = 0, cus = 256;
                hipGetDevice(&dev);
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                slots = 8 * cus;
            }
            grid = dim3((unsigned)total, 1);
            if (queue && world == 1) {
                // dynamic tile order: as many workgroups as the chip holds (8 per CU), tiles off *queue (zero now)
                // (one CU's worth fewer: the engine owns a CU, and a workgroup that is not resident from the
                // start would take its first, static tile late)
                const bool w8 = upd_waves == 8 && K >= 2 * TILE && !skew &&
                                (upd_w8_max_tiles == 0 || total <= upd_w8_max_tiles);
                use_w8 = w8;
                const int cap = w8 ? slots / 2 - 4 : slots - 8;       // resident workgroups: 4 or 8 per CU, one CU's worth fewer
                if (total > cap + (w8 ? 4 : 8)) {
                    a.queue = queue; a.ntiles = (unsigned)total;
                    grid = dim3((unsigned)cap, 1);
                }
            }
        }
    }
    const bool trailing = (K > TILE) && world == 1;     // (the first trailing update has K = 256 - front padding: still role 0)
    if (trailing && use_w8 && a.lower_only) hipLaunchKernelGGL((update_kernel<64, 16, 0, 8>), grid, dim3(512), 0, s, a);
    else if (trailing) hipLaunchKernelGGL((update_kernel<64, 8, 0>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((update_kernel<64, 8, 1>), grid, dim3(256), 0, s, a);
}

// sharded evaluation: gather the solved rows this rank owns into its slot of the owner-packed panel buffer (see kernels.h)
__global__ void __launch_bounds__(256)
pack_rows_kernel(const double *A, size_t lda, int col0, int ncols, double *dst, size_t ldp, const int *pmap, int ti_lo,
                 long long slot_lo, long long slot_hi)
{
    const int ti = ti_lo + (int)blockIdx.x;
    const long long off = pmap[ti];
    if (off < slot_lo || off >= slot_hi) return;           // another rank's rows
    const int rho = threadIdx.x & 63, cq = threadIdx.x >> 6;
    for (int c = 4 * (int)blockIdx.y + cq; c < ncols; c += 4 * (int)gridDim.y)
        dst[off + rho + (size_t)c * ldp] = A[(size_t)(64 * ti + rho) + (size_t)(col0 + c) * lda];
}

void launch_pack_rows(const double *A, size_t lda, int col0, int ncols, double *dst, size_t ldp, const int *pmap, int ti_lo,
                      int ti_hi, long long slot_lo, long long slot_hi, hipStream_t s)
{
    if (ti_hi <= ti_lo || ncols <= 0) return;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(ti_hi - ti_lo, 8), dim3(256), 0, s, A, lda, col0, ncols, dst, ldp, pmap, ti_lo,
                       slot_lo, slot_hi);
}

// Which tile the q-th FAR tile task of a step is (DagArgs::ftab).  The far tiles of a step -- the trapezoid behind the next
// panel's columns -- are cut into blocks of bw tile columns x bh tile rows (the block on the diagonal first in every block column,
// then down), and the blocks are dealt to the XCDs as they come: with tasks dealt to XCDs in chunks of 2^xcd_g list positions
// (dag_kernel: position L belongs to XCD (L >> g) & 7), the far tile tasks of ONE XCD's positions, taken in list order, walk
// through one block after the other -- what an XCD has in flight at any moment is about one block: bw + bh operand strips of
// 128 KB instead of an eighth of every strip of a 2000-tile window.  At the end of a step an XCD whose block runs out takes the
// rest of the fullest block.  xcd_g = 0 (one counter for all): the blocks simply follow each other; bw = 0: column-major, the
// order of rounds 4-5.
static void dag_build_far_table(const std::vector<DagStepHost> &steps, std::vector<unsigned> &tab, int xcd_g, int bw, int bh)
{
    tab.assign(steps.size(), 0u);
    for (size_t s = 0; s < steps.size(); ++s) {
        const DagStepHost &st = steps[s];
        tab[s] = (unsigned)tab.size();
        const int nc = std::min(st.W, st.two ? 4 : 2);           // the next panel's columns: the near tiles
        std::vector<std::vector<unsigned>> blocks;
        if (bw <= 0) {
            blocks.emplace_back();
            for (int jl = nc; jl < st.W; ++jl)
                for (int il = jl; il < st.H; ++il) blocks.back().push_back((unsigned)il << 16 | (unsigned)jl);
        } else {
            for (int c0 = nc; c0 < st.W; c0 += bw) {
                const int c1 = std::min(c0 + bw, st.W);
                for (int r0 = c0; r0 < st.H; r0 += bh) {
                    const int r1 = std::min(r0 + bh, st.H);
                    std::vector<unsigned> b;
                    for (int il = r0; il < r1; ++il)
                        for (int jl = c0; jl < c1 && jl <= il; ++jl) b.push_back((unsigned)il << 16 | (unsigned)jl);
                    if (!b.empty()) blocks.push_back(std::move(b));
                }
            }
        }
        const long long ntile = (long long)st.W * st.H - (long long)st.W * (st.W - 1) / 2;
        const long long nfar = ntile - st.near;
        const unsigned nA = 2u * (unsigned)st.nstrip + (unsigned)st.nd_next, perBC = st.two ? 2u * (unsigned)st.nstrip : 0u;
        const bool panels = st.nT > 0;
        // walk the step's list positions in order; every far tile position takes the next tile of its XCD's current block
        size_t next_block = 0;
        size_t cur[8], pos_in[8];
        for (int x = 0; x < 8; ++x) { cur[x] = (size_t)-1; pos_in[x] = 0; }
        std::vector<size_t> taken_back(blocks.size(), 0);         // tiles taken off a block's END by XCDs that ran out
        long long q = 0;
        const unsigned total = (unsigned)ntile + st.nT;
        for (unsigned pos = 0; pos < total && q < ntile; ++pos) {
            if (panels && ((pos >= st.tpos && pos < st.tpos + nA) || (pos >= st.p2 && pos < st.p2 + perBC) ||
                           (pos >= st.p3 && pos < st.p3 + perBC)))
                continue;                                         // a panel task
            const long long qt = q++;                             // the position's index among the step's update tiles
            if (qt < (long long)st.near) continue;
            const int x = xcd_g > 0 ? (int)(((st.base + pos) >> xcd_g) & 7u) : 0;
            unsigned tile = 0;
            bool got = false;
            while (!got) {
                if (cur[x] != (size_t)-1 && pos_in[x] + taken_back[cur[x]] < blocks[cur[x]].size()) {
                    tile = blocks[cur[x]][pos_in[x]++];
                    got = true;
                } else if (next_block < blocks.size()) {
                    cur[x] = next_block++; pos_in[x] = 0;
                } else {
                    // no fresh block left: the last tile of the block with the most tiles still to go
                    size_t best = (size_t)-1, left = 0;
                    for (int y = 0; y < 8; ++y)
                        if (cur[y] != (size_t)-1) {
                            const size_t l = blocks[cur[y]].size() - pos_in[y] - taken_back[cur[y]];
                            if (l > left) { left = l; best = cur[y]; }
                        }
                    if (best == (size_t)-1) break;                // (cannot happen: as many far positions as far tiles)
                    tile = blocks[best][blocks[best].size() - 1 - taken_back[best]++];
                    got = true;
                }
            }
            tab.push_back(tile);
        }
        (void)nfar;
    }
}

// ---- the dependency-driven schedule: table of steps (host) and launch --------------------------------------------
// Steps for a factorisation with nt column tiles and mt row tiles (trim64: the last 64 rows hold nothing), first panel
// (tiles 0, 1) already formed in place; kskip leading columns of it are unit vectors (front padding) and are skipped.
// lead: far tiles of a step in front of its panel tasks.  Returns the number of tasks.
unsigned dag_build_steps(int nt, int mt, int trim64, int kskip, int lead, int min_tiles, int split, std::vector<DagStepHost> &out,
                         int lead2, int lead3, std::vector<unsigned> *ftab, int xcd_g, int bw, int bh)
{
    out.clear();
    if (ftab) ftab->clear();
    unsigned base = 0;
    int prev_two = 1;
    for (int k = 0; k + 2 < nt; k += 2) {
        const int t = k + 2;
        {   // the DAG schedule covers the head of the factorisation: steps of at least min_tiles update tiles (behind them a step
            // is bound by its dependency chain, not by the chip, and the classic schedule's chain is the shorter one)
            const long long H0 = 2LL * (mt - t) - (trim64 ? 1 : 0), W0 = 2LL * (nt - t);
            if (W0 * H0 - W0 * (W0 - 1) / 2 < min_tiles) break;
        }
        DagStepHost st;
        st.tj0 = 2 * t;
        st.H = 2 * (mt - t) - (trim64 ? 1 : 0);
        st.W = 2 * (nt - t);
        st.two = t + 1 < nt ? 1 : 0;
        const long long tiles = (long long)st.W * st.H - (long long)st.W * (st.W - 1) / 2;
        const int nc = std::min(st.W, st.two ? 4 : 2);
        st.near = (unsigned)(nc * st.H - nc * (nc - 1) / 2);
        st.nstrip = std::max(0, st.H - (st.two ? 4 : 2));
        st.nT = (unsigned)(st.nstrip * (st.two ? 6 : 2));
        st.k0 = k * TILE + (k == 0 ? kskip : 0);
        st.K = 2 * TILE - (k == 0 ? kskip : 0);
        st.need = prev_two ? 6 : 2;
        st.nd_next = 0; st.split = 0;
        st.tpos = 0; st.p2 = 0; st.p3 = 0;
        st.base = 0;
        prev_two = st.two;
        (void)tiles;
        out.push_back(st);
    }
    if (!out.empty()) {      // the panel behind the last step is formed by whoever continues (classic kernels): no panel tasks
        DagStepHost &l = out.back();
        l.nT = 0; l.nstrip = 0;
    }
    // the diagonal-block tiles of step s + 1 in two halves: the early one rides in step s's list behind its T1 tasks.  (steps
    // from 1 on: K = 256 there; step s must form the panel of s + 1, i.e. have panel tasks)
    if (split)
        for (size_t s = 0; s + 1 < out.size(); ++s)
            if (out[s].nstrip > 0 && out[s].two) {
                out[s].nd_next = out[s + 1].two ? 10 : 3;
                out[s + 1].split = 1;
            }
    for (auto &st : out) {
        const long long ntile = (long long)st.W * st.H - (long long)st.W * (st.W - 1) / 2;
        st.nT += (unsigned)st.nd_next;
        const long long nA = 2LL * st.nstrip + st.nd_next;
        const long long perBC = st.two ? 2LL * st.nstrip : 0;
        const long long far = ntile - st.near;
        st.tpos = st.near + (unsigned)std::min<long long>(far, lead);
        st.base = base;
        base += (unsigned)ntile + st.nT;
        // T2 `lead2` far tiles behind the T1 group, T3 `lead3` behind the T2 group (as far as the step has far tiles left)
        long long rem = ntile - (long long)st.tpos;          // update tiles behind the T1 group (tpos counts tiles only so far)
        if (st.nT == 0) { st.p2 = st.p3 = st.tpos; continue; }
        const long long d2 = std::min<long long>(rem, perBC ? lead2 : 0);
        rem -= d2;
        const long long d3 = std::min<long long>(rem, perBC ? lead3 : 0);
        st.p2 = (unsigned)(st.tpos + nA + d2);
        st.p3 = (unsigned)(st.p2 + perBC + d3);
    }
    if (ftab) dag_build_far_table(out, *ftab, xcd_g, bw, bh);
    return base;
}

void launch_dag(double *A, size_t lda, double *P, const double *Wt, const DagStepHost *dsteps, int nsteps, unsigned ntasks,
                unsigned *queue, unsigned *tdone, unsigned *pdone, int pstride, unsigned *pall, double *partbuf, unsigned *dcount,
                unsigned *sig, unsigned *out, unsigned *xr, unsigned *abort_word, hipStream_t s, unsigned long long *trace,
                const unsigned *alive, int xcc_quota, unsigned *hw, const unsigned *ftab, int xcd_g, unsigned *xcnt)
{
    static_assert(sizeof(DagStepHost) == sizeof(DagStep), "host and device step records");
    if (nsteps <= 0 || ntasks == 0) return;
    DagArgs a;
    a.A = A; a.lda = lda; a.P = P; a.Wt = Wt;
    a.steps = (const DagStep *)dsteps; a.nsteps = nsteps; a.ntasks = ntasks;
    a.queue = queue; a.tdone = tdone; a.pdone = pdone; a.pstride = pstride; a.pall = pall;
    a.partbuf = partbuf; a.dcount = dcount;
    a.sig = sig; a.out = out; a.xr = xr; a.abort_word = abort_word; a.trace = trace; a.hw = trace ? hw : nullptr;
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 256;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = 8 * cus;
    }
    const unsigned cap = (unsigned)(slots - 8);          // one CU's worth fewer: the engine owns a CU
    a.alive = alive; a.xcc_quota = (alive && xcc_quota > 0) ? (unsigned)xcc_quota : 0u;
    a.ftab = ftab; a.xcd_g = (ftab && xcnt && xcd_g > 0) ? (unsigned)xcd_g : 0u;
    a.xcnt = xcnt;
    hipLaunchKernelGGL(dag_kernel, dim3(ntasks < cap ? ntasks : cap), dim3(256), 0, s, a);
}

void launch_update(double *A, size_t lda, int k0, int K, int ti0, int ti1, int tj0, int tj1,
                   bool lower_only, hipStream_t s, unsigned *sig, int sig_tile,
                   unsigned *wait_word, unsigned *abort_word, unsigned *queue, int band_hi, int ext0,
                   int skew, int trim64, int skip_lo, int skip_hi)
{
    launch_update_from(A, lda, A + (size_t)k0 * lda, lda, K, ti0, ti1, tj0, tj1, lower_only, s, 1, 1, 0, sig, sig_tile,
                       wait_word, abort_word, queue, band_hi, ext0, skew, k0 / TILE, trim64, nullptr, skip_lo, skip_hi);
}

void launch_finalize_cols(const double *A, size_t lda, int c0, int c1, int n, int row0, int nr,
                          double *out, hipStream_t s)
{
    hipLaunchKernelGGL(finalize_kernel, dim3(nr * nr + 1), dim3(1024), 0, s, A, lda, c0, c1, n, row0, nr, out, 0, 0,
                       (const double *)nullptr, 0);
}

void launch_finalize(const double *A, size_t lda, int n, int row0, int nr, double *out, hipStream_t s, int skew, int npad,
                     const double *A2, int a2_cols)
{
    hipLaunchKernelGGL(finalize_kernel, dim3(nr * nr + 1), dim3(1024), 0, s, A, lda, 0, n, n, row0, nr, out, skew, npad,
                       skew ? (const double *)nullptr : A2, a2_cols);
}

size_t row_reduce_scratch_doubles(int n, int m)
{
    return (size_t)2 * (size_t)m * (size_t)((n + 255) / 256);
}

void launch_row_reduce(const double *A, size_t lda, int n, int rowy, int row0, int m,
                       double *stoch, double *quad, double *scratch, hipStream_t s, int skew, int npad,
                       const double *A2, int a2_cols)
{
    if (m <= 0) return;
    const int cchunk = 2 * TILE, nchunks = (n + cchunk - 1) / cchunk;
    hipLaunchKernelGGL(row_reduce_kernel, dim3((m + 255) / 256, nchunks), dim3(256), 0, s,
                       A, lda, n, rowy, row0, m, scratch, cchunk, skew, npad, skew ? (const double *)nullptr : A2, a2_cols);
    hipLaunchKernelGGL(row_reduce_final_kernel, dim3((m + 255) / 256), dim3(256), 0, s,
                       scratch, m, nchunks, stoch, quad);
}

}  // namespace cocons

