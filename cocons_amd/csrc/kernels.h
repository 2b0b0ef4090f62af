// kernels.h -- argument blocks and launchers shared by the HIP sources and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <vector>
#include "../../include/cocons_hip.h"

namespace cocons {

enum SmoothKind : int {
    SMOOTH_ZERO = 0,           // fixed-smoothness branch: vector stays zero (cocons_full.cpp:83-88)
    SMOOTH_LOGISTIC_SQRT = 1,  // sqrt(Pexpfma_new_smoothness) (:93)
    SMOOTH_EXP = 2             // classic: Pexpfma_new(smooth) (:524)
};

// theta-derived coefficient vectors, formed on the host exactly as the reference forms
// them (2*scale_je, 2*scale_je+aniso, 0.5*std.dev; cocons_full.cpp:64-66,101-104) and
// passed by value in the kernel-argument segment (no per-evaluation H2D copy).
struct ThetaVecs {
    double tilt[COCONS_P_MAX];
    double two_scale_je[COCONS_P_MAX];
    double aniso[COCONS_P_MAX];
    double sqrt_vector[COCONS_P_MAX];
    double half_sd[COCONS_P_MAX];
    double nugget[COCONS_P_MAX];
    double smooth[COCONS_P_MAX];
    double sd[COCONS_P_MAX];
};

struct LocArgs {
    int n, p;
    const double *X; int ldx;       // n x p column-major
    const double *locs; int ldl;    // n x 2 column-major
    double *out; size_t stride;     // LOCP_FIELDS x stride SoA
    int smooth_kind;
    double smooth_min, smooth_max;
    ThetaVecs th;
};

struct PairArgs {
    int n;                 // number of (column-side) locations
    int m;                 // rect only: number of row-side locations
    const double *rows;    // SoA of the row side (== cols for the symmetric kernel)
    const double *cols;    // SoA of the column side
    size_t stride;         // SoA stride of the column side (and rows for sym)
    size_t stride_rows;    // SoA stride of the row side (rect)
    double *out; size_t ld;
    int nrows_out, ncols_out;   // extent to write (>= n pads with identity / zeros)
    int bj0;               // sym: first 64-wide tile column to assemble (sharded path), else 0
    int H;                 // sym: tile rows of the trapezoid (set by the launcher)
    int blocked;           // sym: 8 x 8 pair patches per wave step (set by the launcher, Bessel modes)
    double gr;             // global_range
    double nu_fixed;       // closed-form modes
    double pad_diag;       // sym: diagonal of the identity padding beyond n (0 = the default, 1.0)
    // sym, sharded evaluation: only the tile ROWS of the 256-row blocks this rank owns are assembled -- block b belongs to
    // rank (b / own_group) % own_world (own_world <= 1: every row)
    int own_world = 0, own_rank = 0, own_group = 1;
};

struct RhsArgs {
    int n, p;
    const double *X; int ldx;
    double mean[COCONS_P_MAX];
    int use_trend;
    const double *src; int lds;     // n x nrows column-major (z or x_betas)
    double *out; size_t ld;
    int row0, nrows;       // nrows source rows, written at out rows row0..
    int nrows_zero;        // further rows (row0+nrows ..) cleared to zero
    int col0, ncols_out;   // column range [col0, ncols_out)
    int skew, npad;        // packed band target (band_index): 0 = dense
};

void launch_loc_params(const LocArgs &a, hipStream_t s);
void launch_pair_sym(int mode, bool mirror, const PairArgs &a, hipStream_t s);
void launch_pair_rect(int mode, const PairArgs &a, hipStream_t s);
void launch_rhs_rows(const RhsArgs &a, hipStream_t s);
// entries of the sparse/taper covariance for a CSR pattern (1-based indices, device arrays)
void launch_taper(int mode, bool pred, int nrows, int nnz, const int *ci, const int *rp, const double *rows,
                  size_t stride_rows, const double *cols, size_t stride, double nu_fixed, double *out, hipStream_t s,
                  const double *tapv = nullptr, double *A = nullptr, size_t lda = 0, int row0 = 0,    // A: dense target, see TaperArgs
                  int skew = 0, int npad = 0);                                                       // packed band target (band_index)
// rows idx[0..nidx) of the dense covariance (cor != 0: of cov2cor of it); out row b at out + b * n
// zero the tiles inside the envelope (d_hi: device copy of FactorView::hi; max_band = max over c of hi[c] - c)
void launch_band_zero(double *A, size_t lda, const int *d_hi, int nt, int max_band, hipStream_t s, int skew = 0);
// identity on the padding diagonal of a taper handle's buffer
void launch_front_identity(double *A, size_t lda, int pad0, int rows, hipStream_t s);   // columns [0, pad0): unit vectors
void launch_pad_identity(double *A, size_t lda, int n, int npad, hipStream_t s, int skew = 0);
void launch_cov_rows(int mode, int n, int nidx, const int *idx, const double *loc, size_t stride, double gr,
                     double nu_fixed, int cor, double *out, hipStream_t s);
// out[i] = 2^(1-nu)/Gamma(nu) u^nu K_nu(u) by the device routine of the pair kernels (diagnostic)
void launch_matern_points(int n, const double *nu, const double *x, double *out, hipStream_t s);

// ---- factorisation (chol.hip) -------------------------------------------------
constexpr int TILE = 128;          // tile edge of the blocked factorisation

// Packed band storage of a band-limited factorisation (taper handles): tile column c (128 columns) keeps only the rows
// the factor can touch -- `skew` tile rows from its diagonal tile down, then the rows under the matrix -- so the leading
// dimension is skew * 128 + (rows under the matrix) instead of the matrix order, and element (i, j) sits at
//     i_local + j * ld,   i_local = i - 128 (j / 128)  for a row of the matrix (i < npad),
//                                   skew * 128 + (i - npad)  for a row under it.
// skew = 0: the ordinary dense layout.  Kernels that work inside ONE tile column get a shifted base pointer
// (band_base) and keep their global row indices; the rows under the matrix then start at row (c + skew) * 128.
__host__ __device__ inline size_t band_index(int i, int j, size_t ld, int skew, int npad)
{
    if (skew == 0) return (size_t)i + (size_t)j * ld;
    const int il = i < npad ? i - TILE * (j / TILE) : skew * TILE + (i - npad);
    return (size_t)il + (size_t)j * ld;
}
// base pointer with which tile column c of a packed band buffer is addressed by GLOBAL row and column indices
inline double *band_base(double *A, int c, int skew) { return skew ? A - (ptrdiff_t)TILE * c : A; }

// Factor the 128x128 diagonal tile at (c0,c0) in place (lower), write the inverses of
// its eight 16x16 diagonal blocks to dinv (8*256 doubles).  info: atomicMin of the
// 1-based failing column (initialise to INT_MAX).
void launch_potrf_tile(double *A, size_t lda, int c0, double *dinv, int *info, hipStream_t s);
// launch_potrf_tile and launch_trsm_tile (rows as there) in ONE launch: the solve workgroups follow the factorisation through
// the tile's mailbox (ENGINE_MBOX_DOUBLES doubles, every byte 0xff beforehand); bit-identical to the two launches
void launch_potrf_follow(double *A, size_t lda, int c0, int r0, int r1, double *dinv, int *info, double *mbox,
                         unsigned *abort_word, hipStream_t s, int band_r1 = -1, int ext_r0 = 0);
// Resident diagonal-BLOCK engine (one workgroup on a CU of its own) for the 256 x 256 diagonal blocks
// starting at tile t0 (even): see potrf_engine_kernel.  Flag words, all zero at launch:
//   in[t]    raised by the update kernels (launch_update's sig / sig_tile): 3 = tile (t,t) updated;
//   in[t+1]  7 = tiles (t+1,t) and (t+1,t+1) updated;
//   out[t], out[t+1]  raised to 1 when the factor of that diagonal tile (and its Q operands in
//                     dinv + (tile & 1) * 2048) is published;
//   xr[t]    raised to 1 when X = A(t+1,t) L(t)^-T is published.
// abort_word: set by any party whose bounded wait ran out; everybody leaves when it is non-zero.
//   alive    raised by the engine once it is resident; launch_engine_gate(alive, ...) holds a stream until then
// t0 >= nt: a warm-up launch -- the kernel raises alive and leaves (first-dispatch costs paid outside any bounded wait)
// wbuf, pbuf != NULL: the engine of the DAG schedule (launch_dag) -- it also publishes W = L^-1 of every diagonal tile t at
// wbuf + t * 128 * 128 (zeroed once by the caller; complete before out[t]) and a second copy of X in pbuf (shaped like A)
void launch_potrf_engine(double *A, size_t lda, int t0, int nt, double *dinv, int *info,
                         unsigned *in, unsigned *out, unsigned *xr, unsigned *abort_word, unsigned *alive, hipStream_t s,
                         double *wbuf = nullptr, double *pbuf = nullptr, int dag_until = 0,
                         unsigned long long *trace = nullptr,
                         double *mbox = nullptr,                         // pair mode (engine_partner_loop): the tiles' mailboxes
                                                                         // (ENGINE_MBOX_DOUBLES each, index = tile), every byte 0xff at
                                                                         // launch; a second workgroup takes the second tile of every block
                         int in_wait_ms = 0);                            // > 0 (tests): bound of the engine's waits for its input words
                                                                         // in milliseconds instead of the host-paced 3 s
// Abort codes of the bounded hand-off waits (the abort word behind the info word: who gave up).  CLASS in bits 8..11, an index
// -- the tile or, for the persistent launch, the step -- in the low byte, masked so that no index can spill into another class
// (until round 5 the engine's partner reported 0x700 + tile while the followers used the fixed codes 0x7d0 / 0x7e0 / 0x7f0: from
// tile 208 on a partner's time-out read as a follower's).  One decoder for the host: abort_class().
//   0x100 / 0x200  engine waiting for in[t] / in[t+1] (host-paced)      0x300  panel solve waiting for the engine's tile
//   0x400  split panel's second workgroup waiting in its exchange mailbox   0x500  in-panel update waiting for xr[t]
//   0x600  start-up gate (engine / partner not resident)                 0x700  the engine's partner following tile t
//   0x800  a follower of potrf_follow_kernel (plain / band-limited schedule)  0x900  the reductions waiting for the last tile
//   0xa00 .. 0xe00  waits of the persistent launch (dag_kernel: index = step)  0xf00  next-diagonal-block workgroups of the
//   panel launch waiting in the strip mailbox
constexpr unsigned ABORT_ENGINE_IN0 = 0x100u, ABORT_ENGINE_IN1 = 0x200u, ABORT_PANEL = 0x300u, ABORT_XCHG = 0x400u, ABORT_INPANEL = 0x500u,
                   ABORT_GATE = 0x600u, ABORT_PARTNER = 0x700u, ABORT_FOLLOW = 0x800u, ABORT_LAST_TILE = 0x900u, ABORT_STRIPBOX = 0xf00u;
__host__ __device__ inline unsigned abort_code(unsigned cls, unsigned index) { return cls | (index & 0xffu); }
inline unsigned abort_class(unsigned code) { return code & 0xf00u; }
constexpr size_t ENGINE_MBOX_DOUBLES = 44 * 256;
// nhelp > 0: also waits until that many further workgroups of the engine's launch (the pair partner) are resident
// raise_in != NULL: the gate also raises in[0] = 3, in[1] = 7 (the engine factors the first diagonal block too: launch_potrf_engine t0 = 0)
void launch_engine_gate(unsigned *alive, unsigned *abort_word, hipStream_t s, bool last_tile = false, bool patient = false, int nhelp = 0,
                        unsigned *raise_in = nullptr);
void launch_raise_word(unsigned *word, hipStream_t s);      // *word = 1 (agent scope) by a one-lane kernel: "everything in front of me on this stream is done"
// 1: a kernel on `first` and a kernel launched behind it on `second` overlap (the streams sit on different hardware queues);
// 0: they run one after the other; -1: HIP error.  words: two device words; both streams idle.
int streams_run_concurrently(hipStream_t first, hipStream_t second, unsigned *words);
// rows [r0, r1) x cols [c0, c0+128):  X <- X * L(c0)^{-T}, L read from A(c0,c0).
// wait_word != NULL: the tile comes from the engine -- every workgroup first waits for *wait_word >= 1
// band_r1 >= 0 (band-limited factorisation): rows [r0, band_r1) and [ext_r0, r1) instead of [r0, r1).
// own_world > 1 (sharded evaluation): only the 64-row strips inside 256-row blocks b with (b / own_group) % own_world == own_rank
void launch_trsm_tile(double *A, size_t lda, int c0, int r0, int r1, const double *dinv, hipStream_t s,
                      unsigned *wait_word = nullptr, unsigned *abort_word = nullptr, int band_r1 = -1, int ext_r0 = 0,
                      int own_world = 0, int own_rank = 0, int own_group = 1);
// The panel of a two-tile block of the engine schedule in one launch: rows [r0, r1) of the tile columns at c0 and c0 + 128,
// X0 = B0 L(c0)^-T | B1 -= X0 X(t+1,t)^T | X1 = B1 L(c0+128)^-T, waiting for out0 / xr / out1 where each is needed (dense, unsharded;
// q0, q1: the Q operands of the two diagonal tiles).  Bit-identical to launch_trsm_tile | launch_update (K = 128) | launch_trsm_tile.
// mb0, mb1 (both or neither): the two tiles' mailboxes (the engine's pair mode): the strips follow the tiles while they are formed
void launch_panel_pair(double *A, size_t lda, int c0, int r0, int r1, const double *q0, const double *q1, unsigned *out0,
                       unsigned *xr, unsigned *out1, unsigned *abort_word, hipStream_t s, const double *mb0 = nullptr,
                       const double *mb1 = nullptr,
                       double *smb = nullptr, int ndiag = 0, unsigned *sig = nullptr, int sig_tile = 0,
                       double *xmb = nullptr);   // split panel: exchange mailboxes, PANEL_XMBOX_DOUBLES per 64-row strip, every byte
                                                 // 0xff (the second workgroup of a strip puts the pattern back as it reads)
constexpr size_t PANEL_XMBOX_DOUBLES = 8 * 4 * 256;
// smb (PANEL_SMBOX_DOUBLES doubles, every byte 0xff beforehand) + ndiag = 10 or 3: the launch also updates the NEXT diagonal
// block (two tiles or one) with this panel and raises sig[sig_tile] (+3) / sig[sig_tile + 1] (+7) like launch_update's tiles
// inside the diagonal block do; the update launch that follows must leave those tiles alone (skip_lo / skip_hi)
constexpr size_t PANEL_SMBOX_DOUBLES = 4 * 16 * 4 * 256;
// C(i,j) -= sum_{k in [k0,k0+K)} A(i,k) A(j,k) for tiles with tile-row in [ti0,ti1),
// tile-col in [tj0,tj1); lower_only keeps ti >= tj.  All tile indices in units of TILE.
// sig / sig_tile: hand-off to the engine (sig = the in[] array, sig_tile = even tile of the diagonal block);
// wait_word: an operand tile comes from the engine -- every workgroup first waits for *wait_word >= 1.
// queue: a device word that is ZERO when the launch starts -- the launch then takes about as many workgroups
// as the chip holds and they draw the tiles of the trapezoid from that counter (lower_only launches).
void launch_update(double *A, size_t lda, int k0, int K, int ti0, int ti1, int tj0, int tj1,
                   bool lower_only, hipStream_t s, unsigned *sig = nullptr, int sig_tile = -1,
                   unsigned *wait_word = nullptr, unsigned *abort_word = nullptr, unsigned *queue = nullptr,
                   int band_hi = -1, int ext0 = 0,       // band_hi >= 0: tile rows [ti0, band_hi) and [ext0, ti1)
                   int skew = 0,           // packed band buffer (band_index; with band_hi / ext0): A is its unshifted base
                   int trim64 = 0,         // 1: the last 64 rows of the row range hold nothing (a 128-row tile of right-hand
                                           // sides of which at most 64 rows are used): they are not updated
                   int skip_lo = 0, int skip_hi = 0);   // tiles with both 64-row and 64-column index in [skip_lo, skip_hi) are left
                                                        // alone (the next diagonal block, when the panel's launch has updated it)
// waves per workgroup of the trailing-update kernel: 4 or 8 (512 threads, KC = 16: half the tile latency; default for
// launches of at most set_update_w8_max_tiles tiles, 0 = every launch)
void set_update_waves(int nw);
void set_update_w8_max_tiles(int ntiles);
void set_update_c_wt(int on);           // (experiment) every C tile through L2-bypassing loads and write-through stores
// like launch_update but the (i,k) and (j,k) operands come from a separate panel buffer P, used by the sharded path:
// world > 1: only the tiles whose ROW lies in a 256-row block b with (b / group) % world == rank are updated (row-block
// ownership); pmap (device, one int per 64-row tile, may be null = global row index): element offset of that tile's rows in
// P, whose columns are ldp apart -- the gathered panel is packed by owner (api.hip); [skip_lo, skip_hi): the tiles whose 64-row
// AND 64-column index lie in that range -- a diagonal block its owner has updated ahead of the exchange -- are left out.
void launch_update_from(double *A, size_t lda, const double *P, size_t ldp, int K,
                        int ti0, int ti1, int tj0, int tj1, bool lower_only, hipStream_t s,
                        int group, int world, int rank, unsigned *sig = nullptr, int sig_tile = -1,
                        unsigned *wait_word = nullptr, unsigned *abort_word = nullptr, unsigned *queue = nullptr,
                        int band_hi = -1, int ext0 = 0, int skew = 0, int kblk = 0, int trim64 = 0,
                        const int *pmap = nullptr, int skip_lo = 0, int skip_hi = 0);
// sharded evaluation: the solved rows this rank owns of the 256-column panel at column col0 (ncols columns), gathered into
// its slot of the owner-packed exchange buffer: for every 64-row tile ti in [ti_lo, ti_hi) with pmap[ti] inside the slot
// [slot_lo, slot_hi): dst[pmap[ti] + rho + c * ldp] = A[(64 ti + rho) + (col0 + c) lda]
void launch_pack_rows(const double *A, size_t lda, int col0, int ncols, double *dst, size_t ldp, const int *pmap, int ti_lo,
                      int ti_hi, long long slot_lo, long long slot_hi, hipStream_t s);

// ---- dependency-driven schedule (chol.hip: dag_kernel) -- ONE persistent launch for every trailing update and every
// panel behind the first one.  DagStepHost mirrors the device record (see DagStep in chol.hip for the meaning).
struct DagStepHost {
    unsigned base, near, tpos, nT;
    int H, W, tj0, k0, K, nstrip, two, need, nd_next, split;
    unsigned p2, p3;
};
// only the leading steps with at least min_tiles update tiles are taken (the head of the factorisation); the last of them has
// no panel tasks: the panel behind it is left to the caller's classic kernels
// split != 0: the diagonal-block tiles of the steps from 1 on are computed in two halves (nd_next / split, see DagStep)
// lead: far tiles of a step in front of its T1 tasks (and the early halves); lead2 / lead3: far tiles between them and the T2
// tasks, between those and the T3 tasks
unsigned dag_build_steps(int nt, int mt, int trim64, int kskip, int lead, int min_tiles, int split, std::vector<DagStepHost> &out,
                         int lead2 = 0, int lead3 = 0,
                         std::vector<unsigned> *ftab = nullptr,    // out: which tile every FAR tile task is (launch_dag's ftab; chol.hip:
                                                                   // dag_build_far_table) for tasks dealt to the XCDs in chunks of
                         int xcd_g = 0, int bw = 16, int bh = 16); // 2^xcd_g list positions (0: one counter), blocks of bw x bh tiles
// dsteps: DEVICE copy of the table.  queue, tdone (2 mt (2 mt + 1) / 2 words), pdone ((nsteps + 1) * pstride words,
// pstride >= 2 mt), pall (nsteps + 1 words): zero at launch.  sig / out / xr: the engine's words (launch_potrf_engine with wbuf = Wt, pbuf = P).
void launch_dag(double *A, size_t lda, double *P, const double *Wt, const DagStepHost *dsteps, int nsteps, unsigned ntasks,
                unsigned *queue, unsigned *tdone, unsigned *pdone, int pstride, unsigned *pall, double *partbuf, unsigned *dcount,
                unsigned *sig, unsigned *out, unsigned *xr, unsigned *abort_word, hipStream_t s, unsigned long long *trace = nullptr,
                const unsigned *alive = nullptr, int xcc_quota = 0, unsigned *hw = nullptr,
                const unsigned *ftab = nullptr, int xcd_g = 0,      // DEVICE copy of dag_build_steps' far-tile table; chunk exponent
                unsigned *xcnt = nullptr);                          // the XCDs' task counters: 8 x 32 words (a cache line each), zero at launch
                // alive: the engine's alive word (1 + its XCD); xcc_quota: workgroups of the launch that take part on that XCD
                // partbuf: 2 x 16 x 64 x 64 doubles; dcount: 16 words per step (+ 1 step), zero at launch

// reductions: out[0] = sum_{i<n} log(A(i,i)); out[1 + a*nr + b] = sum_{c<n} A(row0+a,c) A(row0+b,c)
void launch_finalize(const double *A, size_t lda, int n, int row0, int nr, double *out, hipStream_t s,
                     int skew = 0, int npad = 0,           // packed band source (band_index)
                     const double *A2 = nullptr,           // factor of the DAG schedule: below the diagonal blocks the columns
                     int a2_cols = 0);                     // [256, a2_cols) live in A2
// partial version over columns [c0,c1) accumulating into out (atomic adds), sharded path
void launch_finalize_cols(const double *A, size_t lda, int c0, int c1, int n, int row0, int nr,
                          double *out, hipStream_t s);
// per-row reductions for predict: stoch[i] = sum_c A(rowy,c) A(row0+i,c); quad[i] = sum_c A(row0+i,c)^2
// deterministic two-stage reduction; scratch must hold row_reduce_scratch_doubles(n, m) doubles
size_t row_reduce_scratch_doubles(int n, int m);
void launch_row_reduce(const double *A, size_t lda, int n, int rowy, int row0, int m,
                       double *stoch, double *quad, double *scratch, hipStream_t s, int skew = 0, int npad = 0,
                       const double *A2 = nullptr, int a2_cols = 0);   // factor of the DAG schedule, as for launch_finalize

// Y = L E + trend (lower factor L in A), E n x nsim, Y n x nsim
void launch_trmm_lower(const double *A, size_t lda, int n, const double *E, int lde, int nsim,
                       const double *trend, double *Y, int ldy, hipStream_t s);

}  // namespace cocons
