// api.hip -- C ABI of the dense hot path (see include/cocons_hip.h for the contract and the
// reference interface each entry point replaces).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <time.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "../../include/cocons_hip_diag.h"
#include "matern_device.hpp"   // PairMode, LOCP_FIELDS (host-visible enums)

using namespace cocons;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, const char *what = "")
{
    char buf[512];
    snprintf(buf, sizeof buf, fmt, what);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                              \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) {                                                  \
            char b__[512];                                                        \
            snprintf(b__, sizeof b__, "%s failed: %s (%s:%d)", #expr,             \
                     hipGetErrorString(e__), __FILE__, __LINE__);                 \
            g_err = b__;                                                          \
            return -100 - (int)e__;                                               \
        }                                                                         \
    } while (0)

extern "C" const char *cocons_last_error(void) { return g_err.c_str(); }
extern "C" int cocons_abi_version(void) { return 1; }

extern "C" int cocons_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_err = hipGetErrorString(e); return -1; }
    return n;
}

// src/cocons_full.cpp:12-30 -- O(p) host arithmetic, kept on the host.
extern "C" double cocons_sumsmoothlone(const double *x, int len, double lambda, double alpha)
{
    double sum = 0;
    for (int w = 0; w < len; ++w) {
        if (std::abs(x[w]) > 1e-4)
            sum = sum + std::abs(x[w]);
        else
            sum = sum + std::pow(alpha, -1) * (std::log(1 + std::exp(-alpha * x[w])) + std::log(1 + std::exp(alpha * x[w])));
    }
    return lambda * sum;
}

// ---------------------------------------------------------------------------
// theta -> kernel arguments, exactly the host-side preamble of the reference functions
enum { ENGINE_ABORT = -5 };
enum { TH_SD = 0, TH_SCALE = 1, TH_ANISO = 2, TH_TILT = 3, TH_SMOOTH = 4, TH_NUGGET = 5 };

struct ModeSel {
    int mode;          // PairMode
    int smooth_kind;   // SmoothKind
    double nu_fixed;
    double gr;
};

// The mailboxes of the factorisation use the all-ones bit pattern as "not written yet" (chol.hip: the data is its own flag).
// That pattern is a quiet NaN no arithmetic PRODUCES -- the hardware's own NaN is 0x7ff8000000000000 -- but NaN payloads
// PROPAGATE, so an all-ones NaN in the caller's data or parameters could reach a factor block and be waited for until the bounded
// wait gives up (a time-out and a repeat, never a wrong value).  Everything that enters the device is therefore canonicalised:
// an all-ones NaN becomes the standard quiet NaN (R's NA_real_ and NaN are other patterns and pass unchanged).
static inline double canon_nan(double v)
{
    unsigned long long b;
    memcpy(&b, &v, sizeof b);
    return b == ~0ull ? std::numeric_limits<double>::quiet_NaN() : v;
}

static void make_theta_vecs(const double *theta_in, int p, ThetaVecs &tv)
{
    double theta[6 * COCONS_P_MAX];
    for (int i = 0; i < 6 * p; ++i) theta[i] = canon_nan(theta_in[i]);
    memset(&tv, 0, sizeof tv);
    for (int i = 0; i < p; ++i) {
        double sje = (i == 0) ? 0.0 : theta[TH_SCALE * p + i];      // cocons_full.cpp:49,64
        tv.tilt[i] = theta[TH_TILT * p + i];
        tv.two_scale_je[i] = 2 * sje;                               // :101
        tv.aniso[i] = theta[TH_ANISO * p + i];
        tv.sqrt_vector[i] = 2 * sje + theta[TH_ANISO * p + i];      // :66
        tv.half_sd[i] = 0.5 * theta[TH_SD * p + i];                 // :104
        tv.nugget[i] = theta[TH_NUGGET * p + i];
        tv.smooth[i] = theta[TH_SMOOTH * p + i];
        tv.sd[i] = theta[TH_SD * p + i];
    }
}

// which = 0 cov_rns, 1 cov_rns_classic, 2 cov_rns_pred
static ModeSel select_mode(const double *theta, int p, const double *smooth_limits, int which)
{
    ModeSel m;
    m.gr = canon_nan(1 / std::exp(-2 * theta[TH_SCALE * p + 0]));   // :62, :351, :501
    m.nu_fixed = 0.0;
    if (which == 1) { m.mode = MODE_MEAN; m.smooth_kind = SMOOTH_EXP; return m; }
    if (which == 2) { m.mode = MODE_GEOM; m.smooth_kind = SMOOTH_LOGISTIC_SQRT; return m; }
    bool fix = true;                                                // allzeroelements, types.h:56-63
    for (int i = 1; i < p; ++i)
        if (theta[TH_SMOOTH * p + i] != 0) fix = false;
    if (fix && smooth_limits[0] == smooth_limits[1]) {              // :85-88
        double v = smooth_limits[0];
        m.nu_fixed = v;
        m.smooth_kind = SMOOTH_ZERO;
        if (std::fabs(v - 0.5) < 1e-6) m.mode = MODE_HALF;          // types.h:65-70
        else if (std::fabs(v - 1.5) < 1e-6) m.mode = MODE_THREEHALF;
        else if (std::fabs(v - 2.5) < 1e-6) m.mode = MODE_FIVEHALF;
        else m.mode = MODE_GEOM;   // quirk: zero smooth vector -> u = 0 -> every entry = diag_ii
    } else {
        m.mode = MODE_GEOM;
        m.smooth_kind = SMOOTH_LOGISTIC_SQRT;
    }
    return m;
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---------------------------------------------------------------------------
// RCCL, loaded on first use (dlopen): the library itself has no load-time dependency on it, and a
// process that never shards never touches it.
struct RcclApi {
    void *h;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char *(*GetErrorString)(ncclResult_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *);
    ncclResult_t (*CommAbort)(ncclComm_t);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *);      // optional (RCCL >= 2.18): may be null
};

static RcclApi *rccl_api()
{
    static RcclApi api;
    static int state = 0;      // 0 untried, 1 ok, -1 failed
    if (state == 0) {
        state = -1;
        // RCCL must sit on the SAME HIP / HSA runtime this library runs on.  A process may hold two ROCm
        // stacks (PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64 / librccl): a bare
        // dlopen("librccl.so.1") then returns whichever was loaded first, possibly one whose HSA copy was
        // never initialised ("no ROCm-capable device is detected").  So: look beside the libamdhip64 that
        // hipGetDeviceCount resolves to, and only then fall back to the search path.
        std::vector<std::string> names;
        {
            Dl_info di;
            if (dladdr((void *)&hipGetDeviceCount, &di) && di.dli_fname) {
                std::string dir(di.dli_fname);
                size_t slash = dir.rfind('/');
                if (slash != std::string::npos) {
                    dir.resize(slash + 1);
                    names.push_back(dir + "librccl.so.1");
                    names.push_back(dir + "librccl.so");
                }
            }
        }
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        names.push_back("/opt/rocm/lib/librccl.so.1");
        void *h = nullptr;
        for (const std::string &nm : names)
            if ((h = dlopen(nm.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) { g_err = std::string("cannot load RCCL: ") + dlerror(); return nullptr; }
        api.h = h;
#define RSYM(field, name) *(void **)(&api.field) = dlsym(h, name); if (!api.field) { g_err = "RCCL symbol missing: " name; return nullptr; }
        RSYM(GetUniqueId, "ncclGetUniqueId")
        RSYM(CommInitRank, "ncclCommInitRank")
        RSYM(CommInitAll, "ncclCommInitAll")
        RSYM(CommDestroy, "ncclCommDestroy")
        RSYM(Broadcast, "ncclBroadcast")
        RSYM(AllReduce, "ncclAllReduce")
        RSYM(AllGather, "ncclAllGather")
        RSYM(GroupStart, "ncclGroupStart")
        RSYM(GroupEnd, "ncclGroupEnd")
        RSYM(GetErrorString, "ncclGetErrorString")
        RSYM(CommCount, "ncclCommCount")
        RSYM(CommUserRank, "ncclCommUserRank")
        RSYM(CommCuDevice, "ncclCommCuDevice")
        RSYM(CommAbort, "ncclCommAbort")
        RSYM(Send, "ncclSend")
        RSYM(Recv, "ncclRecv")
#undef RSYM
        *(void **)(&api.CommSplit) = dlsym(h, "ncclCommSplit");
        state = 1;
    }
    return state == 1 ? &api : nullptr;
}

static void rccl_comm_destroy(ncclComm_t c)
{
    RcclApi *R = rccl_api();
    if (R && c) R->CommDestroy(c);
}

#define NCCLCHK(expr)                                                             \
    do {                                                                          \
        ncclResult_t r__ = (expr);                                                \
        if (r__ != ncclSuccess) {                                                 \
            char b__[512];                                                        \
            snprintf(b__, sizeof b__, "%s failed: %s (%s:%d)", #expr,             \
                     rccl_api() ? rccl_api()->GetErrorString(r__) : "?", __FILE__, __LINE__); \
            g_err = b__;                                                          \
            return -200 - (int)r__;                                               \
        }                                                                         \
    } while (0)

// ---------------------------------------------------------------------------
struct cocons_fit {
    int n, p, r, q, device;
    pid_t pid;
    int npad, nt;            // padded order, tiles of 128
    int rhs_cap;             // rows reserved under the matrix (multiple of 128)
    int rhs_act;             // rows under the matrix the CURRENT operation uses (multiple of 128, <= rhs_cap)
    size_t lda;
    hipStream_t stream;
    bool own_stream;
    double *dX, *dlocs, *dz, *dxb;
    double *dloc;            // LOCP_FIELDS x npad
    double *dA;
    double *dinv;            // 2 x 8 x 256
    int *dinfo;
    double *dout;            // reductions
    double *hout;            // pinned mirror
    int *hinfo, *hinfo_init = nullptr;
    double smooth_limits[2];
    size_t out_cap;
    // predict scratch
    double *dlocp, *dXp, *dlocsp, *dstoch, *dquad, *dred;
    int pred_cap;
    // sharded state
    int rank, world, nrhs_cur;
    int nslot;                    // > 0: the last nslot of the npad rows / columns are SLOTS (npad = n + nslot): columns with a huge
                                  // diagonal and nothing else, rows that hold the right-hand sides of an evaluation (at most nslot
                                  // of them) INSIDE the last tile of the matrix -- no tile row under the matrix (enqueue_eval)
    int n_user, pad0;             // n = pad0 + n_user: dense handles keep pad0 = npad - n_user placeholder observations IN FRONT
                                  // of the caller's (their columns are made unit vectors before every factorisation,
                                  // launch_front_identity), so that n == npad and no padding sits in the trailing matrix
    double *xbuf[2];
    size_t xbuf_bytes;
    bool xbuf_own;
    hipEvent_t ev[8];
    hipStream_t stream2;          // stream the resident diagonal-tile engine is launched on
    hipEvent_t ev_eng;            // orders the engine launch behind the reset of its flag words
    unsigned *dflags;             // flags_cap words each: in[t], out[t], xr[t] (see launch_potrf_engine); 64: the alive word;
                                  // flags_cap: tile counters of the trailing updates
    int flags_cap;
    bool engine_ok;               // false: this handle never uses the resident engine (batch slots, band-limited taper fits)
    bool engine_live;             // the engine of the NEXT factorize call is already launched (engine_start)
    bool engine_used;             // the factorisation enqueued last runs on the engine schedule
    int border_clean = -1;        // nr >= 0: the rows [nr, rhs_act) under the matrix are known to be exactly zero in every column
                                  // (they were zeroed, and a SUCCESSFUL factorisation keeps zero rows zero): the next
                                  // evaluation with the same nr does not zero them again (-1: unknown)
    int border_pending = -1;      // what border_clean becomes when the operation in flight turns out to have succeeded
    bool engine_active_last;      // the last COMPLETED operation ran on the engine schedule (cocons_fit_engine_state)
    int engine_skip;              // operations still to run on the plain schedule after a hand-off timed out (back-off)
    int engine_fails;             // consecutive time-outs (the back-off doubles with each, up to 64 operations)
    int engine_retries;           // time-outs in the life of the handle, each answered by one repeat on the plain schedule
    int engine_last_abort;        // abort word of the last time-out (who gave up: see info_status)
    long long engine_ops;         // operations enqueued on the engine schedule so far (the first one's gate is patient)
    // dependency-driven schedule (factorize_dag): second buffer shaped like dA, tile inverses, task words, step table
    double *dP; size_t dP_elems;
    double *dWt; int dWt_tiles;
    double *dpart;                // early halves of the split diagonal-block tiles (2 x 16 x 64 x 64 doubles)
    unsigned *ddag; size_t ddag_words;      // [queue (64 words)] [tdone] [pdone]
    void *ddag_steps; int dag_nsteps; unsigned dag_ntasks;
    int dag_key[13];              // (nt, mt, trim, kskip, lead, min_tiles, split, lead2, lead3, order, xcd, bw, bh) the step table was built for
    unsigned *ddag_ftab; size_t ddag_ftab_words;   // which tile every far tile task is (dag_build_steps' table), device copy
    int dag_xcd_g;                // chunk exponent of the XCD-aware deal the table was built for (0: one counter)
    bool dag_have_ftab;           // the current step table comes with a far-tile table
    size_t ddag_xcnt_off;         // offset (words) of the XCDs' task counters inside ddag
    unsigned long long *ddag_trace; size_t dag_trace_tasks;   // diagnostics (cocons_debug_tune("dag_trace", 1)): 4 stamps per task
    size_t dag_trace_elems;       // allocated 64-bit words of ddag_trace (5 per task + 8 per tile pair)
    bool dag_next;                // the engine launched by engine_start is the DAG schedule's (publishes W and the second X)
    int engine_pair_live;         // 1: the engine launched for the next factorisation has a pair partner (it counts itself in alive[2])
    int engine_t0;                // first tile of the engine launched for the next factorisation: 0 (it factors the first diagonal block too) or 2
    size_t smb_off, smb_elems;           // inside dmbox: strip mailboxes, one per diagonal block (the panel launch's next-diagonal-block
                                         // update), and xmb_off: the panel launch's exchange mailboxes, one per 64-row strip (split panel)
    size_t xmb_off, xmb_elems;
    double *dmbox; size_t dmbox_elems;   // one mailbox per tile (mbox_reset): the engine's pair mode, the panel kernel and potrf_solve's
                                         // followers read a tile's factor from there while it is being formed
    bool follow_used, follow_off;        // the operation being enqueued used launch_potrf_follow; it timed out once on this handle: off
    double enq_host_us; long long enq_calls;      // (diagnostics) host time spent enqueueing evaluations, calls: cocons_debug_host_enqueue
    bool dag_used;                // the factorisation enqueued last ran the DAG schedule: its factor is split over dA and dP
    double dag_flops; int dag_events;   // profile runs: update flops inside the DAG launch; 1 = the first event pair is that launch
    // taper fit (cocons_fit_create_taper): the spam pattern (1-based CSR) with the taper's entries; the
    // -2 log-likelihood is then that of the TAPERED covariance, evaluated through the dense factorisation
    int taper_nnz;                // > 0: taper fit
    int *d_tci, *d_trp;
    double *d_tval;               // taper entries (constant)
    std::vector<int> *taper_hi;   // envelope of the (reordered) pattern per tile column: see FactorView::hi
    int *d_thi; int taper_maxband; // device copy of taper_hi and max_c (hi[c] - c)
    int skew;                     // > 0: the factorisation buffer is PACKED (kernels.h band_index): every tile column keeps
                                  // `skew` (= taper_maxband) tile rows from its diagonal tile down plus the rows under the
                                  // matrix -- O(n x bandwidth) doubles instead of n^2
    std::vector<int> *taper_inv;  // position of the caller's observation i in the handle's order (reverse Cuthill-McKee)
    // collectives of the natively sharded evaluation (see "native sharded evaluation" below)
    int coll_kind;                // 0 none, 1 RCCL communicator, 2 caller-provided transport
    int coll_rank, coll_world;
    ncclComm_t comm;
    bool comm_own;                // the communicator was created by cocons_fit_comm_init (destroy it with the fit)
    cocons_bcast_fn cb_bcast;
    cocons_allreduce_fn cb_allreduce;
    cocons_allgather_fn cb_allgather;
    struct ShardState *shard;     // plan, buffers and events of the sharded evaluation (row-block ownership)
    void *cb_user;
    hipStream_t cstream;          // stream the bulk exchange (all-gather of the solved rows) is issued on
    hipStream_t cstream_l;        // stream the 0.56 MB broadcasts of the factored diagonal blocks are issued on: the chain from one
                                  // diagonal block to the next never queues behind an all-gather (== cstream when the
                                  // broadcasts have no communicator of their own)
    ncclComm_t comm_l;            // RCCL: a second communicator over the same ranks (ncclCommSplit) for those broadcasts --
                                  // operations of ONE communicator are serialised whatever stream they are given; null: comm
    bool comm_l_own;
    double *dcoll;                // device staging of the final all-reduce (RCCL)
    double upd_flops;             // algorithmic flops of the event-timed trailing updates (profile runs)
    // host copies of the inputs + lazily created clones: the slots of cocons_neg2loglik_batch
    std::vector<double> *h_locs, *h_X, *h_z, *h_xb;
    std::vector<cocons_fit *> *slots;
    bool sorted;                  // observations are stored in Morton order (see fit_create_impl)
    cocons_fit *unsorted;         // lazily created clone in the ORIGINAL order (marginal simulation)
    std::recursive_mutex *op_mu;  // held by every entry point for as long as it works on this handle (FIT_ENTER), and by another
                                  // handle's stream self-test while it launches probe kernels on this handle's streams
                                  // (engine_warm: try_lock under the registry's lock -- a busy handle is not probed, a probed one
                                  // can neither be used nor destroyed until the probe is over).  A pointer: the struct is memset
};

static void shard_state_free(struct ShardState *S);

// Every live handle of the process: engine_warm tests a new handle's streams against the streams of the others (a resident
// engine of one handle must not share a hardware queue with the main stream of another: the batch slots and callers with
// several handles in flight run exactly that combination).
static std::mutex g_reg_mutex;
static std::vector<cocons_fit *> g_registry;
static void registry_add(cocons_fit *f) { std::lock_guard<std::mutex> lk(g_reg_mutex); g_registry.push_back(f); }
static void registry_remove(cocons_fit *f)
{
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    g_registry.erase(std::remove(g_registry.begin(), g_registry.end(), f), g_registry.end());
}

static int fit_check(cocons_fit *f)
{
    if (!f) return fail(-1, "null fit handle");
    if (f->pid != getpid())
        return fail(-2, "fit handle was created in another process (fork); create it in the worker");
    hipError_t e = hipSetDevice(f->device);
    if (e != hipSuccess) return fail(-3, "hipSetDevice: %s", hipGetErrorString(e));
    return 0;
}

// Every public entry point that works on a handle: validate it, then hold its operation lock until the call returns.
// THREADING CONTRACT (include/cocons_hip.h): one handle serves one call at a time -- a second thread that enters with the
// same handle waits here --; different handles may be used, created and destroyed from different threads concurrently.
#define FIT_ENTER(f)                                                   \
    if (int rc__ = fit_check(f)) return rc__;                          \
    std::lock_guard<std::recursive_mutex> op_guard__(*(f)->op_mu)

static int fit_alloc_matrix(cocons_fit *f, int rhs_rows)
{
    int cap = round_up(rhs_rows > 0 ? rhs_rows : 1, TILE);
    f->rhs_act = cap;        // a buffer grown by an earlier predict call must not slow later evaluations
    f->border_clean = -1; f->border_pending = -1;      // every user of the rows under the matrix comes through here
    if (f->dA && cap <= f->rhs_cap) return 0;
    if (f->dA) { HIPCHK(hipFree(f->dA)); f->dA = nullptr; }
    f->rhs_cap = cap;
    f->lda = (size_t)(f->skew > 0 ? f->skew * TILE : f->npad) + cap;
    HIPCHK(hipMalloc(&f->dA, f->lda * (size_t)f->npad * sizeof(double)));
    // never-written parts must not hold NaN bit patterns: a band-limited factorisation only clears its envelope, and
    // 0 * garbage must stay 0 whatever the allocator hands back
    HIPCHK(hipMemsetAsync(f->dA, 0, f->lda * (size_t)f->npad * sizeof(double), f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return 0;
}

extern "C" void cocons_fit_destroy(cocons_fit *f)
{
    if (!f) return;
    // out of the registry first (no stream self-test of another thread can find the handle any more), then wait for one that
    // found it earlier and is still launching probe kernels on its streams (engine_warm holds op_mu for that long)
    registry_remove(f);
    if (f->op_mu) { f->op_mu->lock(); f->op_mu->unlock(); }
    if (f->pid == getpid()) {
        hipSetDevice(f->device);
        if (f->stream) hipStreamSynchronize(f->stream);
        if (f->stream2) hipStreamSynchronize(f->stream2);
        hipFree(f->dX); hipFree(f->dlocs); hipFree(f->dz); hipFree(f->dxb); hipFree(f->dloc);
        hipFree(f->dA); hipFree(f->dinv); hipFree(f->dinfo);      // (dout is part of dinfo's allocation)
        hipFree(f->dlocp); hipFree(f->dXp); hipFree(f->dlocsp); hipFree(f->dstoch); hipFree(f->dquad); hipFree(f->dred);
        if (f->xbuf_own) { hipFree(f->xbuf[0]); hipFree(f->xbuf[1]); }
        hipHostFree(f->hinfo);                                  // (hout, hinfo_init: the same allocation)
        for (auto &e : f->ev) if (e) hipEventDestroy(e);
        if (f->ev_eng) hipEventDestroy(f->ev_eng);
        hipFree(f->dflags);
        if (f->dmbox) hipFree(f->dmbox);
        hipFree(f->dP); hipFree(f->dWt); hipFree(f->ddag); hipFree(f->ddag_steps); hipFree(f->ddag_trace); hipFree(f->dpart);
        hipFree(f->ddag_ftab);
        hipFree(f->d_tci); hipFree(f->d_trp); hipFree(f->d_tval); hipFree(f->d_thi);
        if (f->cstream_l && f->cstream_l != f->cstream) { hipStreamSynchronize(f->cstream_l); hipStreamDestroy(f->cstream_l); }
        if (f->cstream) { hipStreamSynchronize(f->cstream); hipStreamDestroy(f->cstream); }
        shard_state_free(f->shard);
        hipFree(f->dcoll);
        if (f->comm_l && f->comm_l_own) rccl_comm_destroy(f->comm_l);
        if (f->comm && f->comm_own) rccl_comm_destroy(f->comm);
        if (f->slots) { for (auto c : *f->slots) cocons_fit_destroy(c); delete f->slots; f->slots = nullptr; }
        if (f->unsorted) { cocons_fit_destroy(f->unsorted); f->unsorted = nullptr; }
        if (f->stream2) hipStreamDestroy(f->stream2);
        if (f->own_stream && f->stream) hipStreamDestroy(f->stream);
    }
    delete f->h_locs; delete f->h_X; delete f->h_z; delete f->h_xb;
    delete f->taper_hi; delete f->taper_inv;
    delete f->op_mu;
    delete f;
}

static int engine_warm(cocons_fit *f);

static cocons_fit *fit_create_impl(int n, int p, int r, int q, const double *locs,
                                   const double *X, const double *z, const double *x_betas,
                                   const double *smooth_limits, int device, bool allow_sort, bool defer_matrix = false,
                                   bool want_engine = true, bool return_locked = false)
{
    if (n <= 0 || p <= 0 || p > COCONS_P_MAX || r < 0 || q < 0 || !locs || !X || !smooth_limits ||
        (r > 0 && !z) || (q > 0 && !x_betas)) {
        fail(-1, "cocons_fit_create: bad argument");
        return nullptr;
    }
    cocons_fit *f = new cocons_fit();
    memset(f, 0, sizeof *f);
    f->op_mu = new std::recursive_mutex();
    f->n = n; f->p = p; f->r = r; f->q = q;
    f->device = device < 0 ? 0 : device;
    f->pid = getpid();
    f->npad = round_up(n, TILE);
    f->nt = f->npad / TILE;
    f->smooth_limits[0] = smooth_limits[0];
    f->smooth_limits[1] = smooth_limits[1];
    f->world = 1;
#define CK(expr)                                                                  \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) {                                                  \
            fail(-100, "cocons_fit_create: %s", hipGetErrorString(e__));          \
            cocons_fit_destroy(f);                                                \
            return nullptr;                                                       \
        }                                                                         \
    } while (0)
    CK(hipSetDevice(f->device));
    // every stream the library creates is non-blocking: nothing here ever joins the NULL stream
    CK(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    f->own_stream = true;
    // Spatial (Morton / Z-order) permutation of the observations.  -2 loglik, the kriging outputs and
    // the simulated fields do not depend on the order of the observed locations (a symmetric
    // permutation of Sigma), but the Bessel kernels run faster when neighbouring indices are
    // neighbouring points (8 x 8 pair patches then see similar distances).  Everything the handle
    // keeps -- locs, X, z, x_betas, host copies -- is stored in the permuted order.
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    {
        const char *e = getenv("COCONS_SPATIAL_SORT");
        if (allow_sort && (e ? atoi(e) : 1) && n > 64) {
            double lo[2] = {locs[0], locs[n]}, hi[2] = {locs[0], locs[n]};
            for (int i = 0; i < n; ++i)
                for (int d = 0; d < 2; ++d) {
                    double v = locs[i + (size_t)d * n];
                    if (v < lo[d]) lo[d] = v;
                    if (v > hi[d]) hi[d] = v;
                }
            std::vector<unsigned long long> key(n);
            bool finite = true;
            for (int i = 0; i < n && finite; ++i) {
                unsigned q[2];
                for (int d = 0; d < 2; ++d) {
                    double span = hi[d] - lo[d];
                    double t = span > 0 ? (locs[i + (size_t)d * n] - lo[d]) / span : 0.0;
                    if (!(t >= 0.0 && t <= 1.0)) { finite = false; t = 0; }
                    q[d] = (unsigned)(t * 65535.0);
                }
                unsigned long long k = 0;
                for (int b = 0; b < 16; ++b)
                    k |= ((unsigned long long)((q[0] >> b) & 1u) << (2 * b)) | ((unsigned long long)((q[1] >> b) & 1u) << (2 * b + 1));
                key[i] = (k << 32) | (unsigned)i;          // ties keep the input order
            }
            if (finite) {
                std::sort(key.begin(), key.end());
                // keep the caller's order when it is already as coherent as the Morton order (e.g. a
                // regular grid listed row by row): compare the path lengths through the points
                auto path = [&](auto idx) {
                    double s = 0;
                    for (int i = 0; i + 1 < n; ++i) {
                        int a = idx(i), b = idx(i + 1);
                        double dx = locs[a] - locs[b], dy = locs[a + (size_t)n] - locs[b + (size_t)n];
                        s += std::sqrt(dx * dx + dy * dy);
                    }
                    return s;
                };
                double p_in = path([&](int i) { return i; });
                double p_mo = path([&](int i) { return (int)(key[i] & 0xffffffffu); });
                if (p_mo < 0.8 * p_in)
                    for (int i = 0; i < n; ++i) perm[i] = (int)(key[i] & 0xffffffffu);
            }
        }
    }
    f->h_locs = new std::vector<double>(locs, locs + (size_t)2 * n);      // host copies: ORIGINAL order
    f->h_X = new std::vector<double>(X, X + (size_t)n * p);
    f->h_z = new std::vector<double>();
    if (r > 0) f->h_z->assign(z, z + (size_t)n * r);
    f->h_xb = new std::vector<double>();
    if (q > 0) f->h_xb->assign(x_betas, x_betas + (size_t)n * q);
    f->sorted = false;
    for (int i = 0; i < n; ++i)
        if (perm[i] != i) { f->sorted = true; break; }
    // Identity padding in FRONT (handles whose internal order is theirs to choose, i.e. allow_sort): pad0 placeholder
    // observations (copies of the first one; their rows and columns are overwritten by unit vectors before every
    // factorisation) precede the caller's, so that the internal problem has exactly npad = n observations and no padding
    // rides through every trailing update.  Order-dependent entry points then go through the unsorted twin, like after a
    // Morton sort.  COCONS_FRONT_PAD=0: padding behind the observations (rounds 1-2).
    f->n_user = n;
    {
        const char *e = getenv("COCONS_FRONT_PAD");
        f->pad0 = (allow_sort && (e ? atoi(e) : 1)) ? f->npad - n : 0;
        // ... except for a few SLOTS kept behind the observations: rows that carry the right-hand sides of an evaluation
        // through the factorisation as part of the matrix's last tile, instead of a tile row of their own under it (one
        // row in use of 64: 1.9 % of the trailing updates' arithmetic at n = 10 000).  COCONS_RHS_SLOTS=0: off.
        const char *e2 = getenv("COCONS_RHS_SLOTS");
        const int need = round_up(r + (q > p ? q : p) > 0 ? r + (q > p ? q : p) : 1, 16);
        f->nslot = (f->pad0 >= need && r > 0 && (e2 ? atoi(e2) : 1)) ? need : 0;
        f->pad0 -= f->nslot;
    }
    const int pad0 = f->pad0, nint = n + pad0;
    if (pad0 > 0) f->sorted = true;
    auto permute_pad = [&](const double *src, int ncol, bool zero_pad) {
        std::vector<double> out((size_t)nint * ncol);
        for (int c = 0; c < ncol; ++c) {
            // (canon_nan: no all-ones NaN enters the device -- the mailboxes' "not written yet" pattern, see there)
            for (int i = 0; i < pad0; ++i) out[(size_t)i + (size_t)c * nint] = zero_pad ? 0.0 : canon_nan(src[(size_t)perm[0] + (size_t)c * n]);
            for (int i = 0; i < n; ++i) out[(size_t)(pad0 + i) + (size_t)c * nint] = canon_nan(src[(size_t)perm[i] + (size_t)c * n]);
        }
        return out;
    };
    std::vector<double> plocs = permute_pad(locs, 2, false), pX = permute_pad(X, p, false), pz, pxb;
    if (r > 0) pz = permute_pad(z, r, true);
    if (q > 0) pxb = permute_pad(x_betas, q, true);
    locs = plocs.data(); X = pX.data();
    if (r > 0) z = pz.data();
    if (q > 0) x_betas = pxb.data();
    n = nint;                     // from here on: the internal problem
    f->n = n;
    CK(hipMalloc(&f->dX, (size_t)n * p * sizeof(double)));
    CK(hipMalloc(&f->dlocs, (size_t)n * 2 * sizeof(double)));
    CK(hipMemcpyAsync(f->dX, X, (size_t)n * p * sizeof(double), hipMemcpyHostToDevice, f->stream));
    CK(hipMemcpyAsync(f->dlocs, locs, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, f->stream));
    if (r > 0) {
        CK(hipMalloc(&f->dz, (size_t)n * r * sizeof(double)));
        CK(hipMemcpyAsync(f->dz, z, (size_t)n * r * sizeof(double), hipMemcpyHostToDevice, f->stream));
    }
    if (q > 0) {
        CK(hipMalloc(&f->dxb, (size_t)n * q * sizeof(double)));
        CK(hipMemcpyAsync(f->dxb, x_betas, (size_t)n * q * sizeof(double), hipMemcpyHostToDevice, f->stream));
    }
    CK(hipStreamSynchronize(f->stream));      // the staging vectors above go out of scope
    CK(hipMalloc(&f->dloc, (size_t)LOCP_FIELDS * f->npad * sizeof(double)));
    CK(hipMalloc(&f->dinv, 2 * 8 * 256 * sizeof(double)));
    // the two info words -- [0] failing minor (atomicMin), [1] abort word of the engine hand-offs -- sit in the 8 bytes in
    // front of the reduction outputs, on the device and in the pinned host mirror: an evaluation brings both home in ONE copy
    int nr_max = r + (q > p ? q : p);
    f->out_cap = (size_t)(1 + nr_max * nr_max) * (size_t)(f->nt + 2);
    {
        double *dbase = nullptr, *hbase = nullptr;
        CK(hipMalloc(&dbase, (f->out_cap + 1) * sizeof(double)));
        CK(hipHostMalloc(&hbase, (f->out_cap + 2) * sizeof(double)));
        f->dinfo = (int *)dbase; f->dout = dbase + 1;
        f->hinfo = (int *)hbase; f->hout = hbase + 1;
        f->hinfo_init = (int *)(hbase + 1 + f->out_cap);      // constant {0x7f7f7f7f, 0}: reset_info's source
        f->hinfo_init[0] = 0x7f7f7f7f; f->hinfo_init[1] = 0;
    }
    for (auto &e : f->ev) CK(hipEventCreate(&e));
    // (the engine's stream is created by engine_warm, and only for handles that may use the engine: every stream a process
    // holds takes a place in the round robin over the few hardware queues -- a batch slot on the plain schedule that created
    // one pushed the NEXT slot's main stream onto a queue already taken, and kernels of streams that share a queue run one
    // after the other: four "concurrent" slots ran on two queues, round 5's kernel trace)
    f->engine_ok = want_engine;
    CK(hipEventCreateWithFlags(&f->ev_eng, hipEventDisableTiming));
    // (two streams per handle and no more: every stream a process holds competes for the few hardware queues -- a third one
    // per handle, for a panel-overlap experiment since removed, halved the throughput of the batch slots)
    // (a taper handle allocates its buffer once the envelope of its pattern is known: packed, it is a fraction of n^2)
    if (!defer_matrix && fit_alloc_matrix(f, nr_max) != 0) { cocons_fit_destroy(f); return nullptr; }
#undef CK
    if (engine_warm(f) != 0) { cocons_fit_destroy(f); return nullptr; }
    // (return_locked: the caller goes on building the handle -- a batch slot, whose main stream may still be redrawn --, so it
    // enters the registry with its operation lock held and no other thread's stream self-test can touch it before it is done)
    if (return_locked) f->op_mu->lock();
    registry_add(f);
    return f;
}

extern "C" cocons_fit *cocons_fit_create(int n, int p, int r, int q, const double *locs,
                                         const double *X, const double *z, const double *x_betas,
                                         const double *smooth_limits, int device)
{
    return fit_create_impl(n, p, r, q, locs, X, z, x_betas, smooth_limits, device, true);
}

// Taper fit: the handle of an optimisation of GetNeg2loglikelihoodTaper (R/neg2loglikelihood.R:20-53).  The
// pattern (colindices / rowpointers, 1-based as spam stores them, symmetric, diagonal stored) and the taper's
// entries are those of `ref_taper`; cocons_neg2loglik_dense on this handle returns
//   sum_k [ n log 2 pi + 2 sum log diag chol(S) + resid_k' S^-1 resid_k ],   S = taper o cov_rns_taper(theta),
// the value spam's sparse Cholesky gives, obtained here through the DENSE factorisation of S (zeros stored):
// valid while n^2 doubles fit the device, and an n = 10^4 evaluation costs what a dense one costs.  The
// observations keep the caller's order (the pattern refers to it).
extern "C" cocons_fit *cocons_fit_create_taper(int n, int p, int r, const double *locs, const double *X, const double *z,
                                               const double *smooth_limits, int device, int nnz, const int *colindices,
                                               const int *rowpointers, const double *taper_entries)
{
    if (n <= 0 || nnz <= 0 || !colindices || !rowpointers || !taper_entries || r < 1) {
        fail(-1, "cocons_fit_create_taper: bad argument");
        return nullptr;
    }
    if (rowpointers[0] != 1 || rowpointers[n] != nnz + 1) {
        fail(-1, "cocons_fit_create_taper: rowpointers do not match nnz (1-based CSR expected)");
        return nullptr;
    }
    for (int i = 0; i < n; ++i) {
        bool diag = false;
        if (rowpointers[i + 1] < rowpointers[i]) { fail(-1, "cocons_fit_create_taper: rowpointers decrease"); return nullptr; }
        for (int w = rowpointers[i] - 1; w < rowpointers[i + 1] - 1; ++w) {
            if (colindices[w] < 1 || colindices[w] > n) { fail(-1, "cocons_fit_create_taper: column index out of range"); return nullptr; }
            if (colindices[w] == i + 1) diag = true;
        }
        if (!diag) {
            char msg[96];
            snprintf(msg, sizeof msg, "row %d stores no diagonal entry", i + 1);
            fail(-1, "cocons_fit_create_taper: %s", msg);
            return nullptr;
        }
    }
    // Order the observations by reverse Cuthill-McKee on the pattern (COCONS_TAPER_RCM=0: keep the caller's order):
    // the value does not depend on the order, the envelope of the factor does, and the factorisation below
    // only touches tiles inside it.
    std::vector<int> perm(n), inv(n);            // perm[new] = old, inv[old] = new
    {
        const char *e = getenv("COCONS_TAPER_RCM");
        const bool rcm = e ? atoi(e) != 0 : true;
        if (!rcm) { for (int i = 0; i < n; ++i) perm[i] = i; }
        else {
            std::vector<int> deg(n), order;
            std::vector<char> seen(n, 0);
            order.reserve(n);
            for (int i = 0; i < n; ++i) deg[i] = rowpointers[i + 1] - rowpointers[i];
            std::vector<int> nb;
            auto bfs = [&](int root, std::vector<int> &out) {        // Cuthill-McKee order of root's component
                const size_t first = out.size();
                out.push_back(root); seen[root] = 1;
                for (size_t h = first; h < out.size(); ++h) {
                    const int u = out[h];
                    nb.clear();
                    for (int w = rowpointers[u] - 1; w < rowpointers[u + 1] - 1; ++w) {
                        const int v2 = colindices[w] - 1;
                        if (!seen[v2]) { seen[v2] = 1; nb.push_back(v2); }
                    }
                    std::sort(nb.begin(), nb.end(), [&](int a2, int b2) { return deg[a2] != deg[b2] ? deg[a2] < deg[b2] : a2 < b2; });
                    for (int v2 : nb) out.push_back(v2);
                }
            };
            std::vector<int> byd(n);
            for (int i = 0; i < n; ++i) byd[i] = i;
            std::sort(byd.begin(), byd.end(), [&](int a2, int b2) { return deg[a2] != deg[b2] ? deg[a2] < deg[b2] : a2 < b2; });
            for (int c = 0; c < n; ++c) {
                const int start = byd[c];
                if (seen[start]) continue;
                // pseudo-peripheral root: the last vertex of a first sweep from the component's minimum-degree vertex
                std::vector<int> probe;
                bfs(start, probe);
                const int root = probe.back();
                for (int u : probe) seen[u] = 0;
                bfs(root, order);
            }
            for (int i = 0; i < n; ++i) perm[i] = order[n - 1 - i];
        }
        for (int i = 0; i < n; ++i) inv[perm[i]] = i;
    }
    std::vector<double> pl((size_t)2 * n), pX((size_t)p * n), pz((size_t)r * n);
    for (int i = 0; i < n; ++i) {
        const int o = perm[i];
        pl[i] = locs[o]; pl[(size_t)n + i] = locs[(size_t)n + o];
        for (int c = 0; c < p; ++c) pX[(size_t)c * n + i] = X[(size_t)c * n + o];
        for (int c = 0; c < r; ++c) pz[(size_t)c * n + i] = z[(size_t)c * n + o];
    }
    std::vector<int> prp(n + 1), pci(nnz);
    std::vector<double> pte(nnz);
    prp[0] = 1;
    for (int i = 0, w2 = 0; i < n; ++i) {
        const int o = perm[i];
        for (int w = rowpointers[o] - 1; w < rowpointers[o + 1] - 1; ++w, ++w2) {
            pci[w2] = inv[colindices[w] - 1] + 1;
            pte[w2] = taper_entries[w];
        }
        prp[i + 1] = w2 + 1;
    }
    cocons_fit *f = fit_create_impl(n, p, r, 0, pl.data(), pX.data(), pz.data(), nullptr, smooth_limits, device, false, true,
                                    true, true);       // (registered with its operation lock held: it is still being built)
    if (!f) return nullptr;
    // envelope per tile column: row i of the factor is non-zero from its first stored column on
    f->taper_hi = new std::vector<int>(f->nt, 0);
    f->taper_inv = new std::vector<int>(inv);
    {
        std::vector<int> &hi = *f->taper_hi;
        for (int c = 0; c < f->nt; ++c) hi[c] = c + 1 < f->nt ? c + 1 : f->nt;
        for (int i = 0; i < n; ++i) {
            int first = i;
            for (int w = prp[i] - 1; w < prp[i + 1] - 1; ++w) if (pci[w] - 1 < first) first = pci[w] - 1;
            const int ti = i / TILE;
            for (int c = first / TILE; c <= ti; ++c) if (hi[c] < ti + 1) hi[c] = ti + 1;
        }
        const char *e = getenv("COCONS_TAPER_BAND");
        if (e && atoi(e) == 0) hi.clear();           // dense factorisation of the tapered matrix
        if (!hi.empty()) {
            // The schedule works on 256-column blocks and updates the square [t, hb) x [t, hb) with a block's panel:
            // make the bound per block (both tile columns, at least the next diagonal block) and monotone, so that every
            // tile an update touches lies inside the bound of its own column -- which is what gets zeroed.
            std::vector<int> h2(f->nt);
            int run = 0;
            for (int c = 0; c < f->nt; ++c) {
                const int k = c & ~1;
                int hb = hi[k];
                if (k + 1 < f->nt && hi[k + 1] > hb) hb = hi[k + 1];
                const int need = k + 4 < f->nt ? k + 4 : f->nt;
                if (hb < need) hb = need;
                if (hb > f->nt) hb = f->nt;
                if (hb > run) run = hb;
                h2[c] = run;
            }
            hi = h2;
            f->taper_maxband = 0;
            for (int c = 0; c < f->nt; ++c) if (hi[c] - c > f->taper_maxband) f->taper_maxband = hi[c] - c;
            // packed band storage unless switched off (COCONS_TAPER_PACKED=0: the dense n x n buffer, only its band used)
            const char *pk = getenv("COCONS_TAPER_PACKED");
            if (!(pk && atoi(pk) == 0) && f->taper_maxband < f->nt) f->skew = f->taper_maxband;
        }
    }
    if (fit_alloc_matrix(f, r + p) != 0) { f->op_mu->unlock(); cocons_fit_destroy(f); return nullptr; }
    // the device keeps the lower triangle of the pattern only (the upper half is never evaluated)
    {
        int w2 = 0;
        for (int i = 0; i < n; ++i) {
            const int a0 = prp[i] - 1, a1 = prp[i + 1] - 1;
            prp[i] = w2 + 1;
            for (int w = a0; w < a1; ++w)
                if (pci[w] - 1 <= i) { pci[w2] = pci[w]; pte[w2] = pte[w]; ++w2; }
        }
        prp[n] = w2 + 1;
        nnz = w2;
    }
    bool ok = hipMalloc(&f->d_tci, (size_t)nnz * sizeof(int)) == hipSuccess &&
              hipMalloc(&f->d_trp, (size_t)(n + 1) * sizeof(int)) == hipSuccess &&
              hipMalloc(&f->d_tval, (size_t)nnz * sizeof(double)) == hipSuccess &&
              hipMemcpy(f->d_tci, pci.data(), (size_t)nnz * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(f->d_trp, prp.data(), (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(f->d_tval, pte.data(), (size_t)nnz * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && !f->taper_hi->empty())
        ok = hipMalloc(&f->d_thi, (size_t)f->nt * sizeof(int)) == hipSuccess &&
             hipMemcpy(f->d_thi, f->taper_hi->data(), (size_t)f->nt * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        fail(-100, "cocons_fit_create_taper: device allocation or upload failed");
        f->op_mu->unlock();
        cocons_fit_destroy(f);
        return nullptr;
    }
    f->taper_nnz = nnz;
    // inside a narrow envelope the trailing updates are too short to hide the engine's hand-offs behind (4.9 against
    // 4.7 ms at n = 10^4): plain schedule
    if (!f->taper_hi->empty()) f->engine_ok = false;
    f->op_mu->unlock();
    return f;
}

// Does the handle hold exactly these data (bitwise)?  The R glue's handle cache asks on a miss of its address check
// (glue/cocons_hip_glue.c): O(n) host work, no device call.  1 = the same, 0 = different (or a taper handle / bad argument).
extern "C" int cocons_fit_same_data(cocons_fit *f, int n, int p, int r, int q, const double *locs, const double *X,
                                    const double *z, const double *x_betas, const double *smooth_limits)
{
    if (!f || !locs || !X || !smooth_limits || f->taper_nnz > 0) return 0;
    if (f->n_user != n || f->p != p || f->r != r || f->q != q) return 0;
    if (f->smooth_limits[0] != smooth_limits[0] || f->smooth_limits[1] != smooth_limits[1]) return 0;
    if (memcmp(f->h_locs->data(), locs, (size_t)2 * n * sizeof(double)) != 0) return 0;
    if (memcmp(f->h_X->data(), X, (size_t)n * p * sizeof(double)) != 0) return 0;
    if (r > 0 && (!z || memcmp(f->h_z->data(), z, (size_t)n * r * sizeof(double)) != 0)) return 0;
    if (q > 0 && (!x_betas || f->h_xb->size() != (size_t)n * q ||
                  memcmp(f->h_xb->data(), x_betas, (size_t)n * q * sizeof(double)) != 0)) return 0;
    return 1;
}

extern "C" void *cocons_fit_stream(cocons_fit *f) { return f ? (void *)f->stream : nullptr; }

extern "C" int cocons_fit_set_stream(cocons_fit *f, void *stream)
{
    FIT_ENTER(f);
    // both streams idle before the swap: no event wait of the panel stream may refer to work on a
    // stream that is about to be destroyed
    HIPCHK(hipStreamSynchronize(f->stream));
    if (f->stream2) HIPCHK(hipStreamSynchronize(f->stream2));
    if (f->own_stream) { HIPCHK(hipStreamDestroy(f->stream)); f->own_stream = false; }
    f->stream = (hipStream_t)stream;
    return 0;
}

extern "C" int cocons_fit_sync(cocons_fit *f)
{
    FIT_ENTER(f);
    HIPCHK(hipStreamSynchronize(f->stream));
    return 0;
}

// ---------------------------------------------------------------------------
// assembly of Sigma (+ identity padding) into the factorisation buffer, lower triangle.
// bj0 / ncols restrict the 64-wide tile columns (sharded path); full range otherwise.
static void assemble_sigma(cocons_fit *f, const double *theta, int which, int col0, int col1)
{
    ThetaVecs tv;
    make_theta_vecs(theta, f->p, tv);
    ModeSel ms = select_mode(theta, f->p, f->smooth_limits, which);
    LocArgs la;
    la.n = f->n; la.p = f->p;
    la.X = f->dX; la.ldx = f->n;
    la.locs = f->dlocs; la.ldl = f->n;
    la.out = f->dloc; la.stride = f->npad;
    la.smooth_kind = ms.smooth_kind;
    la.smooth_min = f->smooth_limits[0]; la.smooth_max = f->smooth_limits[1];
    la.th = tv;
    launch_loc_params(la, f->stream);
    PairArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.n = f->n; pa.m = f->n;
    pa.rows = f->dloc; pa.cols = f->dloc;
    pa.stride = f->npad; pa.stride_rows = f->npad;
    pa.out = f->dA; pa.ld = f->lda;
    pa.nrows_out = f->npad; pa.ncols_out = col1;
    pa.bj0 = col0 / 64;
    if (pa.bj0 < f->pad0 / 64) pa.bj0 = f->pad0 / 64;      // 64-wide tile rows / columns that are placeholders only: not assembled
    pa.pad_diag = f->nslot > 0 ? 1e300 : 1.0;               // slot columns: a diagonal no right-hand side can turn negative
    pa.gr = ms.gr; pa.nu_fixed = ms.nu_fixed;
    launch_pair_sym(ms.mode, false, pa, f->stream);
}

// Taper fit: Sigma_tap = taper o cov_rns_taper(theta) (R/neg2loglikelihood.R:25-31) as a dense lower triangle.
// Parameters exactly as cocons_cov_rns_taper prepares them (FULL scale vector, src/cocons_taper.cpp:207).
static int assemble_sigma_taper(cocons_fit *f, const double *theta)
{
    ThetaVecs tv;
    make_theta_vecs(theta, f->p, tv);
    for (int i = 0; i < f->p; ++i) tv.two_scale_je[i] = canon_nan(2 * theta[TH_SCALE * f->p + i]);
    ModeSel ms = select_mode(theta, f->p, f->smooth_limits, 0);
    LocArgs la;
    la.n = f->n; la.p = f->p;
    la.X = f->dX; la.ldx = f->n;
    la.locs = f->dlocs; la.ldl = f->n;
    la.out = f->dloc; la.stride = f->npad;
    la.smooth_kind = ms.smooth_kind;
    la.smooth_min = f->smooth_limits[0]; la.smooth_max = f->smooth_limits[1];
    la.th = tv;
    launch_loc_params(la, f->stream);
    // zero what the factorisation will read: the tiles inside the envelope (the rows under the matrix are written in
    // full by the right-hand-side kernel), or the whole buffer when the factorisation is not band-limited
    if (f->d_thi) launch_band_zero(f->dA, f->lda, f->d_thi, f->nt, f->taper_maxband, f->stream, f->skew);
    else HIPCHK(hipMemsetAsync(f->dA, 0, f->lda * (size_t)f->npad * sizeof(double), f->stream));
    launch_taper(ms.mode, false, f->n, f->taper_nnz, f->d_tci, f->d_trp, f->dloc, f->npad, f->dloc, f->npad,
                 ms.nu_fixed, nullptr, f->stream, f->d_tval, f->dA, f->lda, 0, f->skew, f->npad);
    launch_pad_identity(f->dA, f->lda, f->n, f->npad, f->stream, f->skew);
    return 0;
}

static int no_taper(cocons_fit *f, const char *who)
{
    if (f->taper_nnz > 0) return fail(-1, "%s: not available on a taper fit (cocons_neg2loglik_dense is)", who);
    return 0;
}

// right-hand-side rows under the matrix: rows npad.. : z columns (minus trend), then xb columns
static void assemble_rhs(cocons_fit *f, const double *mean, bool use_trend, const double *xb, int nxb,
                         int col0, int col1, bool zero_rest = true, bool slots = false)
{
    if (slots) { zero_rest = false; col1 = f->n; }      // (the assembly zeroed the slot rows; their own columns are not touched)
    RhsArgs ra;
    memset(&ra, 0, sizeof ra);
    ra.n = f->n; ra.p = f->p; ra.X = f->dX; ra.ldx = f->n;
    ra.use_trend = use_trend ? 1 : 0;
    if (use_trend) for (int i = 0; i < f->p; ++i) ra.mean[i] = canon_nan(mean[i]);
    ra.src = f->dz; ra.lds = f->n;
    ra.out = f->dA; ra.ld = f->lda;
    ra.skew = f->skew; ra.npad = f->npad;
    ra.row0 = slots ? f->n : f->npad; ra.nrows = f->r;
    ra.nrows_zero = (nxb > 0 || !zero_rest) ? 0 : f->rhs_act - f->r;
    ra.col0 = col0; ra.ncols_out = col1;
    launch_rhs_rows(ra, f->stream);
    if (nxb > 0) {
        ra.use_trend = 0;
        ra.src = xb; ra.lds = f->n;
        ra.row0 = (slots ? f->n : f->npad) + f->r; ra.nrows = nxb;
        ra.nrows_zero = zero_rest ? f->rhs_act - f->r - nxb : 0;
        launch_rhs_rows(ra, f->stream);
    }
}

// Bordered right-looking factorisation, outer block = 2 tiles (256 columns).
//   panel(k)  : potrf(k) | trsm(k) | update tile column k+1 (K=128) | potrf(k+1) | trsm(k+1)
//   U1(k)     : update of the NEXT block's two tile columns with panel k (K = 256)
//   U2(k)     : update of everything right of that
// Look-ahead: panel(k+2) runs on a second stream as soon as U1(k) is done, concurrently
// with U2(k) on the main stream; U1(k+2) waits for it.  mt = total tile rows (matrix + rhs
// rows).  Optional per-launch timing of U2 via events (ev_upd): appended (start, stop).
// the matrix a factorisation runs on: column tiles nt, row tiles mt (>= nt: rows under the square)
struct FactorView {
    double *A;
    size_t lda;
    int nt, mt;
    const int *hi = nullptr;   // band-limited factorisation (taper handles): hi[c] = one past the last tile row of tile
                               // column c that can be non-zero in the factor (envelope of the pattern); nullptr = dense
    int skew = 0;              // > 0: A is a packed band buffer (kernels.h band_index) of `skew` tile rows per tile column
    int trim = 0;              // 1: the last 64 of the mt * 128 rows hold nothing (the tile of right-hand sides has at most 64
                               // rows in use): no kernel of the factorisation touches them
    bool dag_ok = false;       // the caller reads the factor through launch_finalize(..., A2 = dP) only: the dependency-driven
                               // schedule may be used (its factor is split over two buffers)
};

static FactorView main_view(cocons_fit *f)
{
    FactorView v;
    v.A = f->dA; v.lda = f->lda; v.nt = f->nt; v.mt = f->nt + f->rhs_act / TILE;
    v.hi = (f->taper_hi && !f->taper_hi->empty()) ? f->taper_hi->data() : nullptr;
    v.skew = f->skew;
    return v;
}

// one past the last band row tile of the 256-column block starting at tile k (at least the next diagonal block, so
// that the update which the engine's hand-off hangs on always covers it); -1 = dense
static int band_hi(const FactorView &v, int k)
{
    if (!v.hi) return -1;
    int h = v.hi[k];
    if (k + 1 < v.nt && v.hi[k + 1] > h) h = v.hi[k + 1];
    const int need = k + 4 < v.nt ? k + 4 : v.nt;
    if (h < need) h = need;
    return h < v.nt ? h : v.nt;
}

// tile factorisation + the panel solve below it.  (ONE launch for the two -- the solve's workgroups fetch their rows, wait for
// a word the factorising workgroup raises, take L and solve -- was built and measured in round 5: SLOWER, taper path 4.18 -> 4.57
// ms, batch at n = 4096 1066 -> 1031 evaluations/s: the boundary between the two launches costs less than the write-through
// factor and the serialised fetch of L behind the word; removed.)
// (Round 5, later: ONE launch after all -- not behind a word but FOLLOWING the factorisation through the tile's mailbox, the way the
// engine's partner does: potrf_follow_kernel.  The mailboxes are filled by mbox_reset at the start of the factorisation.)
static bool follow_on(cocons_fit *f);
static void potrf_solve(cocons_fit *f, double *A, size_t lda, int tile, int r0, int r1, double *q, hipStream_t s, int br, int er)
{
    if (follow_on(f) && ((size_t)tile + 1) * ENGINE_MBOX_DOUBLES <= f->smb_off) {
        f->follow_used = true;
        launch_potrf_follow(A, lda, tile * TILE, r0, r1, q, f->dinfo,
                            f->dmbox + (size_t)tile * ENGINE_MBOX_DOUBLES, (unsigned *)(f->dinfo + 1), s, br, er);
        return;
    }
    launch_potrf_tile(A, lda, tile * TILE, q, f->dinfo, s);
    launch_trsm_tile(A, lda, tile * TILE, r0, r1, q, s, nullptr, nullptr, br, er);
}

static void panel_ops(cocons_fit *f, const FactorView &v, int k, hipStream_t s)
{
    const int nt = v.nt, mt = v.mt;
    double *A = v.A;
    const size_t lda = v.lda;
    double *q0 = f->dinv, *q1 = f->dinv + 8 * 256;
    const int hb = band_hi(v, k);                       // rows [.., hb) of the band, then the rows under the matrix [nt, mt)
    const int br = hb >= 0 ? hb * TILE : -1, er = nt * TILE;
    const int r1 = mt * TILE - 64 * v.trim;
    potrf_solve(f, A, lda, k, (k + 1) * TILE, r1, q0, s, br, er);
    if (k + 1 < nt) {
        launch_update(A, lda, k * TILE, TILE, k + 1, mt, k + 1, k + 2, true, s, nullptr, -1, nullptr, nullptr, nullptr, hb, nt,
                      0, v.trim);
        potrf_solve(f, A, lda, k + 1, (k + 2) * TILE, r1, q1, s, br, er);
    }
}

// Schedule switches: read from the environment once per process, and settable afterwards through cocons_debug_tune (the
// diagnostics header) so that variants can be timed in alternation inside ONE process on ONE device.
struct Tunables {
    int engine = 1;          // COCONS_ENGINE: 1 = diagonal blocks are factored by the resident engine beside the updates
    int upd_dynamic = 1;     // COCONS_UPD_DYNAMIC
    int dag = 1;             // COCONS_DAG: 1 = the head of the factorisation under the dependency-driven schedule (one persistent
                             // launch for its updates and panels, dag_kernel); 0 = the classic schedule throughout
    // where a step's panel tasks sit in its list: `lead` far tiles, T1 (+ early halves), `lead2` far tiles, T2, `lead3` far
    // tiles, T3 -- each group about where the chip gets to it when the engine publishes what it waits for (the chip draws ~32
    // tasks per us; first tile out ~85 us into a step, strip (t+1, t) ~18 us later, second tile ~60 us after that), so that
    // the workgroups that draw them neither wait with a slot in hand nor come late.  One block at 3600 (round 4's first
    // form): -1.4 %; at 2400: -0.9 %; everything between (800 .. 2000, 400 .. 900, 1800 .. 2400) measures alike.
    int dag_lead = 1600, dag_lead2 = 600, dag_lead3 = 1800;
    int dag_min_tiles = 2000;  // COCONS_DAG_MIN_TILES: the DAG launch covers the leading steps of at least this many update tiles
                             // (n = 10^4: 24 of the 39 steps, 94 % of the flops; below n ~ 4200 no step at all).  3000 until the
                             // engine became a pair (round 5): with the shorter chain the break-even moved back, 1400 .. 2200
                             // measure alike, +0.4 % over 3000)
    int dag_split = 1;       // COCONS_DAG_SPLIT: the diagonal-block tiles of a DAG step in two halves, the first one off the chain
    int dag_xcd = 1;         // COCONS_DAG_XCD: 1 = XCD-aware task order of the persistent launch (round 6; chol.hip: dag_position) -- list
                             // positions dealt to the XCDs in chunks of 32, the far tiles of a step dealt so that one XCD's tiles in
                             // flight form one block of dag_bw x dag_bh tiles, a class that falls behind helped by the others: fetched
                             // bytes per launch halve, +2 % evaluations/s at n = 10^4; 0 = one counter for all (rounds 4-5).
                             // dag_order (COCONS_DAG_ORDER): 0 = far tiles column-major as in rounds 4-5
    int dag_order = 1, dag_bw = 16, dag_bh = 16;
    int dag_xcd_min_quota = 128;
    int dag_xcc_quota = -1;  // workgroups of the DAG launch that take part on the engine's XCD (of the 255 that land there; 0: all;
                             // -1: derived from the device, dag_xcc_quota() -- 208 on MI355X)
    int engine_block0 = 1;   // COCONS_ENGINE_BLOCK0: 1 = the engine factors the FIRST diagonal block too (its input words raised by the gate
                             // kernel) and that block's panel is the one-launch panel of every other block; 0 = the first block on the
                             // plain schedule (tile | solve | in-panel update | tile | solve on the main stream), the engine from block 1
    int engine_pair = 1;     // COCONS_ENGINE_PAIR: 1 = the engine is a PAIR of workgroups -- the second one follows the first tile's
                             // factorisation column block by column block (strip solve, tile update) and factors the second tile
                             // (chol.hip: engine_partner_loop); 0 = one workgroup does the four passes one behind the other
    int panel_fused = 1;     // COCONS_PANEL_FUSED: 1 = the panel of a two-tile block of the engine schedule is ONE launch (chol.hip:
                             // panel_pair_kernel); 0 = solve | in-panel update | solve, three launches
    int panel_follow = 1;    // COCONS_PANEL_FOLLOW: 1 = the one-launch panel's strips follow the engine's tiles through their mailboxes
                             // (pair mode) instead of waiting for out[t] / out[t+1] and fetching the factor
    int panel_split = 32;    // COCONS_PANEL_SPLIT: a strip of the one-launch panel is TWO workgroups -- the first follows tile t (X0), the second
                             // follows the first through an exchange mailbox (the in-panel product while X0 is being formed), then tile
                             // t+1 -- in panels of at least this many 64-row strips (0: never, 1: always).  It pays where the panel stands
                             // exposed behind a long update launch (n = 4096: +1.9 %, 6400: +1.5 %, 10^4: +0.6 %) and costs where the engine
                             // is the bound anyway (always on: n = 2116 -4.3 %, n = 1024 -2.2 %); an update of 32 strips' trapezoid is ~30 us
    int panel_diag = 1;      // COCONS_PANEL_DIAG: 1 = the one-launch panel also updates the NEXT diagonal block (extra workgroups that
                             // follow its first strips through a strip mailbox) and the update launch behind it leaves those tiles alone
    int potrf_follow = 1;    // COCONS_POTRF_FOLLOW: 1 = a tile factorisation and the panel solve below it are ONE launch whose solve
                             // workgroups follow the factorisation through a mailbox (chol.hip: potrf_follow_kernel; the plain and
                             // the band-limited schedule); 0 = two launches
    int dag_trace = 0;       // (diagnostics) time stamps per task, cocons_debug_dag_trace
    int gate_sabotage = 0;   // (tests) the next N engine-schedule factorisations wait at the gate for a word nobody raises:
                             // a genuine 5 ms time-out, abort code 0x600, to exercise the fall-back and its book-keeping
    // (tests) a LATE HOST: the thread that enqueues a factorisation sleeps host_delay_us microseconds in front of the launches that
    // raise the engine's input word in[host_delay_tile] (COCONS_DEBUG_HOST_DELAY_US / _TILE) -- what a host thread throttled in
    // mid-enqueue looks like to the resident engine (round 5's recorded time-out 0x112) --, and
    // engine_in_wait_ms > 0 puts the bound of the engine's input waits back to that many milliseconds (rounds 2-4: 100)
    int host_delay_us = 0, host_delay_tile = -1, engine_in_wait_ms = 0;
    bool init = false;
};
static Tunables &tun()
{
    static Tunables t;
    if (!t.init) {
        auto rd = [](const char *name, int &v) { const char *e = getenv(name); if (e) v = atoi(e); };
        rd("COCONS_ENGINE", t.engine);
        rd("COCONS_UPD_DYNAMIC", t.upd_dynamic);
        rd("COCONS_DAG", t.dag);
        rd("COCONS_DAG_LEAD", t.dag_lead);
        rd("COCONS_DAG_LEAD2", t.dag_lead2);
        rd("COCONS_DAG_LEAD3", t.dag_lead3);
        rd("COCONS_DAG_MIN_TILES", t.dag_min_tiles);
        rd("COCONS_DAG_SPLIT", t.dag_split);
        rd("COCONS_DAG_XCC_QUOTA", t.dag_xcc_quota);
        rd("COCONS_DAG_XCD", t.dag_xcd);
        rd("COCONS_DAG_ORDER", t.dag_order);
        rd("COCONS_DAG_BW", t.dag_bw);
        rd("COCONS_DAG_BH", t.dag_bh);
        rd("COCONS_ENGINE_PAIR", t.engine_pair);
        rd("COCONS_ENGINE_BLOCK0", t.engine_block0);
        rd("COCONS_PANEL_FUSED", t.panel_fused);
        rd("COCONS_POTRF_FOLLOW", t.potrf_follow);
        rd("COCONS_PANEL_FOLLOW", t.panel_follow);
        rd("COCONS_PANEL_DIAG", t.panel_diag);
        rd("COCONS_PANEL_SPLIT", t.panel_split);
        rd("COCONS_DEBUG_HOST_DELAY_US", t.host_delay_us);
        rd("COCONS_DEBUG_HOST_DELAY_TILE", t.host_delay_tile);
        t.init = true;
    }
    return t;
}

static bool follow_on(cocons_fit *f)
{
    return tun().potrf_follow != 0 && !f->follow_off && f->dmbox != nullptr &&
           f->smb_off >= ((size_t)f->nt + 2) * ENGINE_MBOX_DOUBLES;
}

extern "C" int cocons_debug_tune(const char *name, int value)
{
    if (!name) return fail(-1, "cocons_debug_tune: null name");
    Tunables &t = tun();
    std::string k(name);
    if (k == "engine") t.engine = value;
    else if (k == "upd_dynamic") t.upd_dynamic = value;
    else if (k == "dag") t.dag = value;
    else if (k == "dag_lead") t.dag_lead = value;
    else if (k == "dag_lead2") t.dag_lead2 = value;
    else if (k == "dag_lead3") t.dag_lead3 = value;
    else if (k == "dag_min_tiles") t.dag_min_tiles = value;
    else if (k == "dag_split") t.dag_split = value;
    else if (k == "dag_xcc_quota") t.dag_xcc_quota = value;
    else if (k == "dag_xcd") t.dag_xcd = value;
    else if (k == "dag_order") t.dag_order = value;
    else if (k == "dag_xcd_min_quota") t.dag_xcd_min_quota = value;
    else if (k == "dag_bw") t.dag_bw = value < 1 ? 1 : value;
    else if (k == "dag_bh") t.dag_bh = value < 1 ? 1 : value;
    else if (k == "dag_trace") t.dag_trace = value;
    else if (k == "engine_pair") t.engine_pair = value;
    else if (k == "engine_block0") t.engine_block0 = value;
    else if (k == "panel_fused") t.panel_fused = value;
    else if (k == "potrf_follow") t.potrf_follow = value;
    else if (k == "panel_follow") t.panel_follow = value;
    else if (k == "panel_diag") t.panel_diag = value;
    else if (k == "panel_split") t.panel_split = value;
    else if (k == "gate_sabotage") t.gate_sabotage = value;
    else if (k == "host_delay_us") t.host_delay_us = value;
    else if (k == "host_delay_tile") t.host_delay_tile = value;
    else if (k == "engine_in_wait_ms") t.engine_in_wait_ms = value;
    else if (k == "upd_waves") set_update_waves(value);
    else if (k == "w8_max_tiles") set_update_w8_max_tiles(value);
    else if (k == "c_wt") set_update_c_wt(value);
    else return fail(-1, "cocons_debug_tune: unknown switch %s", name);
    return 0;
}

// Workgroups of the persistent launch that may take part on the XCD that also hosts the engine.
// Measured there: an XCD that runs kernels of two queues does not hold eight of these workgroups on every CU -- 227 instead of
// 248 beside the engine's CU: seven on most, eight on some -- and the set that fits changes when the driver's save / restore
// moves the engine.  The quota keeps the launch below what fits in ANY placement: seven per CU on the XCD's other CUs, less
// one CU's worth and one: (CUs per XCD - 1) x 7 - 9 = 208 for the 32 CUs per XCD of MI355X (the value of round 4's soak runs:
// 0 time-outs in 40 000 evaluations), from hipDeviceProp instead of a constant; COCONS_DAG_XCC_QUOTA overrides.
static int device_cus()
{
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
}

static int dag_xcc_quota()
{
    if (tun().dag_xcc_quota >= 0) return tun().dag_xcc_quota;
    static int derived = -1;
    if (derived < 0) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int xcds = 8;                                 // gfx950: eight accelerator dies (no HIP attribute reports it)
        const int cpx = cus % xcds == 0 ? cus / xcds : 32;
        derived = (cpx - 1) * 7 - 9;
        if (derived < 8) derived = 8;
    }
    return derived;
}

// COCONS_ENGINE: 1 (default) = diagonal tiles are factored by the resident engine while the trailing
// update runs; 0 = every kernel in order on one stream
static bool engine_enabled() { return tun().engine != 0; }

// one trailing-update launch (tile columns [t0, t1) of the trapezoid below (t0, t0)), optionally
// bracketed by timing events (profile runs): appended as (start, stop)
static void timed_update(cocons_fit *f, const FactorView &v, int k, int kw, int t0, int t1, hipStream_t s,
                         std::vector<hipEvent_t> *ev_upd, unsigned *sig, int sig_tile, unsigned *queue = nullptr,
                         int skip_tiles = 0)      // > 0: the diagonal block at t0 (that many tiles) was updated by the panel's launch
{
    const int mt = v.mt;
    if (t1 <= t0) return;
    const int hb = band_hi(v, k);                       // band-limited: tile columns and rows [t0, hb), plus the rows [nt, mt)
    if (hb >= 0 && hb < t1) t1 = hb;
    if (t1 <= t0) return;
    unsigned *abort_word = sig ? (unsigned *)(f->dinfo + 1) : nullptr;   // engine schedule: see update_kernel
    hipEvent_t a = nullptr, b = nullptr;
    if (ev_upd) {
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s);
    }
    // the first panel's leading columns are the unit vectors of the front padding (zero below the diagonal): they add
    // nothing to the trailing matrix, so the update starts behind them (whole 16-column chunks; bit-identical)
    const int kskip = (k == 0 && !v.hi) ? (f->pad0 / 16) * 16 : 0;
    launch_update(v.A, v.lda, k * TILE + kskip, kw * TILE - kskip, t0, mt, t0, t1, true, s, sig, sig_tile, nullptr,
                  abort_word, queue, hb, v.nt, 0, v.trim, skip_tiles > 0 ? 2 * t0 : 0, skip_tiles > 0 ? 2 * (t0 + skip_tiles) : 0);
    if (ev_upd) {
        hipEventRecord(b, s);
        ev_upd->push_back(a); ev_upd->push_back(b);
    }
}

// algorithmic flops of the trailing update of block k (tile columns [t0, nt)): lower triangle of the
// trailing block of order m (real columns only) times K, plus the rhs rows:  K m (m+1) + 2 K r m
static void count_update_flops(cocons_fit *f, int kw, int t0)
{
    double m = (double)f->n_user - ((double)t0 * TILE - (double)f->pad0);      // the caller's rows and columns from t0 on
    if (m > (double)f->n_user) m = (double)f->n_user;
    if (m < 0) m = 0;
    const double K = (double)kw * TILE - (t0 == kw ? (double)f->pad0 : 0.0);     // the first panel holds pad0 placeholder columns
    f->upd_flops += K * m * (m + 1.0) + 2.0 * K * (double)f->nrhs_cur * m;
}

// Reset the hand-off words and launch the diagonal-block engine (see factorize) for a factorisation of view
// v on the second stream, ordered behind the reset.  The engine's 8 waves take every VGPR of a CU, so it can
// only be placed on an EMPTY one: enqueue_eval calls this between the covariance assembly and the short
// right-hand-side kernel, and factorize holds the main stream behind a one-lane gate kernel until the engine
// reports itself resident.
// (Launched between the assembly and the first trailing update it could lose that race and then wait for
// a whole update to drain; with workgroups that wait for the engine on every CU it would never be placed.)
static bool engine_wanted(cocons_fit *f, const FactorView &v)
{
    return engine_enabled() && f->engine_ok && f->stream2 != nullptr && f->engine_skip == 0 && v.nt > 4;
}

// the hand-off words and tile counters of one factorisation with nt tiles, zeroed on the main stream
static int flags_reset(cocons_fit *f, int nt)
{
    if (f->flags_cap < nt) {
        if (f->stream2) HIPCHK(hipStreamSynchronize(f->stream2));
        if (f->dflags) { HIPCHK(hipFree(f->dflags)); f->dflags = nullptr; }
        f->flags_cap = round_up(nt + 8, 64);
        HIPCHK(hipMalloc(&f->dflags, (4 * (size_t)f->flags_cap + 64) * sizeof(unsigned)));
    }
    HIPCHK(hipMemsetAsync(f->dflags, 0, (4 * (size_t)f->flags_cap + 64) * sizeof(unsigned), f->stream));
    return 0;
}

// the tiles' mailboxes (the engine's pair mode, the panel kernel, potrf_solve's followers) filled with the pattern that means "not
// written yet" (every byte 0xff; potrf_tile_body: mbox) on the main stream: 88 KB each, 7.1 MB at n = 10^4
static int mbox_reset(cocons_fit *f, int nt, bool engine_schedule = true)
{
    // one allocation, one fill: the tiles' mailboxes | the strip mailboxes (0.5 MB per diagonal block) | the exchange mailboxes of
    // the split panel (64 KB per 64-row strip of the matrix and the rows under it)
    const size_t tiles = ((size_t)nt + 2) * ENGINE_MBOX_DOUBLES;
    const bool panel = tun().panel_follow && tun().panel_fused && tun().engine_pair;
    const size_t smb = panel && tun().panel_diag ? ((size_t)nt / 2 + 2) * PANEL_SMBOX_DOUBLES : 0;
    const size_t xmb = panel && tun().panel_split ? (2 * ((size_t)nt + 2) + 4) * PANEL_XMBOX_DOUBLES : 0;
    const size_t need = tiles + smb + xmb;
    if (f->dmbox_elems < need) {
        HIPCHK(hipStreamSynchronize(f->stream));
        if (f->stream2) HIPCHK(hipStreamSynchronize(f->stream2));
        if (f->dmbox) { HIPCHK(hipFree(f->dmbox)); f->dmbox = nullptr; f->dmbox_elems = 0; }
        HIPCHK(hipMalloc(&f->dmbox, need * sizeof(double)));
        f->dmbox_elems = need;
    }
    f->smb_off = tiles; f->smb_elems = smb;
    f->xmb_off = tiles + smb; f->xmb_elems = xmb;
    // (the plain schedule uses the tiles' mailboxes only)
    HIPCHK(hipMemsetAsync(f->dmbox, 0xff, (engine_schedule ? need : tiles) * sizeof(double), f->stream));
    return 0;
}

// one tile counter per trailing update: see update_kernel's dynamic tile order (COCONS_UPD_DYNAMIC=0: static)
static unsigned *tile_queue(cocons_fit *f, int k)
{
    const int dyn = tun().upd_dynamic;
    return dyn ? f->dflags + 3 * (size_t)f->flags_cap + 64 + k / 2 : nullptr;
}

// Warm-up of the engine's stream at handle creation: ONE launch of the engine kernel that raises its alive word and
// leaves (t0 >= nt), with the launch configuration of the real thing (512 threads, the dynamic LDS, the function
// attribute).  The first dispatch of a kernel on a stream that has never run anything makes the runtime set up the
// hardware queue behind it -- and, for a kernel with a private segment, scratch memory: round 3's engine had 88 B per
// lane, and the driver's box recorded a 5 ms gate time-out on the FIRST engine-schedule operation of a fresh process
// (DESIGN.md section 4a).  The kernel is scratch-free now, and what remains of the first-dispatch cost is paid here,
// outside any bounded wait.
static int engine_warm(cocons_fit *f)
{
    if (!engine_enabled() || f->nt <= 4 || !f->engine_ok) { f->engine_ok = f->engine_ok && f->stream2 != nullptr; return 0; }
    if (!f->stream2) HIPCHK(hipStreamCreateWithFlags(&f->stream2, hipStreamNonBlocking));
    if (int rc = flags_reset(f, f->nt)) return rc;
    HIPCHK(hipStreamSynchronize(f->stream));
    // The engine's stream must not share a hardware queue with the main stream (HIP multiplexes streams onto a few queues;
    // which one a stream gets depends on every stream the PROCESS has created): test it, and draw another stream if it
    // does -- the losers stay alive until a winner is found, so that the next draw lands elsewhere.  No luck: this handle
    // stays on the plain schedule.
    {
        std::vector<hipStream_t> losers;
        int ok = 0;
        for (int attempt = 0; attempt < 8; ++attempt) {
            ok = streams_run_concurrently(f->stream2, f->stream, f->dflags + 3 * (size_t)f->flags_cap + 8);
            if (ok != 0) break;
            losers.push_back(f->stream2);
            f->stream2 = nullptr;
            if (hipStreamCreateWithFlags(&f->stream2, hipStreamNonBlocking) != hipSuccess) { ok = -1; break; }
        }
        // ... and across handles: this handle's engine beside every other live handle's main stream, and every other live
        // handle's engine beside this handle's main stream.  (The own-pair test above says nothing about those: with a few
        // more streams in the process -- a batch slot, a second handle of the caller, torch's -- A.main can share a queue
        // with B.engine and B.main with A.engine: each engine then blocks the kernels the OTHER evaluation waits for, and only
        // the bounded waits end it -- correct values, seconds lost.)  Streams of handles that are busy right now are not
        // probed (the probe needs idle streams); a collision redraws THIS handle's stream and repeats all tests.  Only the
        // four most recent other handles are looked at, and a handle that finds no clash-free draw KEEPS its engine: four
        // hardware queues cannot keep a dozen live handles apart, and a clash only matters between handles that work at
        // the same moment (a test that holds twelve idle handles must not lose the engine on four of them).
        // Threads (round 6, the advisor's finding): the probe launches kernels on ANOTHER handle's streams.  It takes that handle's
        // operation lock first -- try_lock, under the registry's lock: a handle some thread is working on is busy and not probed, a
        // handle being destroyed has left the registry, and a handle that IS probed can neither be used nor destroyed until the
        // probe is over (cocons_fit_destroy waits for the lock) --, and it leaves streams the library does not own alone
        // (cocons_fit_set_stream: a caller's stream may carry the caller's own work).  This handle is not in the registry yet and
        // not in anybody's hands: redrawing ITS streams is safe here, and nowhere later.
        unsigned *words = f->dflags + 3 * (size_t)f->flags_cap + 8;
        for (int round = 0; ok == 1 && round < 8; ++round) {
            std::vector<cocons_fit *> others;
            {
                std::lock_guard<std::mutex> lk(g_reg_mutex);
                for (size_t i = g_registry.size(); i-- > 0 && others.size() < 4;) {      // the four most recent ones
                    cocons_fit *o = g_registry[i];
                    if (o == f || o->pid != f->pid || o->device != f->device || !o->own_stream || !o->stream || !o->stream2 ||
                        !o->engine_ok || !o->op_mu->try_lock())
                        continue;
                    if (hipStreamQuery(o->stream) == hipSuccess && hipStreamQuery(o->stream2) == hipSuccess) others.push_back(o);
                    else o->op_mu->unlock();
                }
            }
            (void)hipGetLastError();
            int clash = 0;                 // 1: this engine stream beside another main stream; 2: this main stream beside another engine
            for (cocons_fit *o : others) {
                int a = streams_run_concurrently(f->stream2, o->stream, words);
                if (a < 0) { ok = -1; break; }
                if (a == 0) { clash = 1; break; }
                if (f->own_stream) {
                    int b = streams_run_concurrently(o->stream2, f->stream, words);
                    if (b < 0) { ok = -1; break; }
                    if (b == 0) { clash = 2; break; }
                }
            }
            for (cocons_fit *o : others) o->op_mu->unlock();      // (streams_run_concurrently drains both streams before it returns)
            if (ok != 1 || clash == 0) break;
            hipStream_t &mine = clash == 1 ? f->stream2 : f->stream;
            losers.push_back(mine);
            mine = nullptr;
            if (hipStreamCreateWithFlags(&mine, hipStreamNonBlocking) != hipSuccess) { ok = -1; break; }
            // the redrawn stream must still pair with this handle's other stream
            int again = streams_run_concurrently(f->stream2, f->stream, words);
            if (again < 0) { ok = -1; break; }
            if (again == 0) {
                // (the redrawn stream shares a queue with this handle's other stream: draw again next round -- the own pair
                // is what must never share)
                hipStream_t &other = clash == 1 ? f->stream2 : f->stream;
                (void)other;
                bool fixed = false;
                for (int t2 = 0; t2 < 4 && !fixed; ++t2) {
                    losers.push_back(mine);
                    mine = nullptr;
                    if (hipStreamCreateWithFlags(&mine, hipStreamNonBlocking) != hipSuccess) { ok = -1; break; }
                    const int r2 = streams_run_concurrently(f->stream2, f->stream, words);
                    if (r2 < 0) { ok = -1; break; }
                    fixed = r2 == 1;
                }
                if (ok != 1) break;
                if (!fixed) { ok = 0; break; }
            }
        }
        for (hipStream_t l : losers) hipStreamDestroy(l);
        if (ok < 0) { (void)hipGetLastError(); return fail(-100, "engine_warm: stream self-test failed"); }
        if (ok == 0) { f->engine_ok = false; return 0; }
        HIPCHK(hipMemsetAsync(f->dflags, 0, (4 * (size_t)f->flags_cap + 64) * sizeof(unsigned), f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
    }
    launch_potrf_engine(nullptr, 0, 0, 0, f->dinv, f->dinfo, f->dflags, f->dflags, f->dflags, (unsigned *)(f->dinfo + 1),
                        f->dflags + 3 * (size_t)f->flags_cap, f->stream2);
    if (tun().dag)         // the other instantiation of the engine (never dereferences its buffers when t0 >= nt)
        launch_potrf_engine(nullptr, 0, 0, 0, f->dinv, f->dinfo, f->dflags, f->dflags, f->dflags, (unsigned *)(f->dinfo + 1),
                            f->dflags + 3 * (size_t)f->flags_cap, f->stream2, f->dinv, f->dinv, 0);
    HIPCHK(hipGetLastError());
    if (f->stream2) HIPCHK(hipStreamSynchronize(f->stream2));
    return 0;
}

// ---- the dependency-driven schedule (chol.hip: dag_kernel) ---------------------------------------------------------
static bool dag_wanted(cocons_fit *f, const FactorView &v)
{
    return tun().dag != 0 && v.dag_ok && !v.hi && !v.skew && v.nt > 4 && f->world == 1;
}

// buffers, zeroed task words and the step table of the factorisation of view v (on the main stream, before the engine starts)
static int dag_prepare(cocons_fit *f, const FactorView &v)
{
    const size_t elems = v.lda * (size_t)f->npad;
    if (f->dP_elems != elems) {
        HIPCHK(hipStreamSynchronize(f->stream));
        if (f->stream2) HIPCHK(hipStreamSynchronize(f->stream2));
        if (f->dP) { HIPCHK(hipFree(f->dP)); f->dP = nullptr; f->dP_elems = 0; }
        HIPCHK(hipMalloc(&f->dP, elems * sizeof(double)));
        HIPCHK(hipMemsetAsync(f->dP, 0, elems * sizeof(double), f->stream));
        f->dP_elems = elems;
    }
    if (!f->dpart) HIPCHK(hipMalloc(&f->dpart, (size_t)2 * 16 * 64 * 64 * sizeof(double)));
    if (f->dWt_tiles < v.nt) {
        HIPCHK(hipStreamSynchronize(f->stream));
        if (f->stream2) HIPCHK(hipStreamSynchronize(f->stream2));
        if (f->dWt) { HIPCHK(hipFree(f->dWt)); f->dWt = nullptr; }
        HIPCHK(hipMalloc(&f->dWt, (size_t)v.nt * TILE * TILE * sizeof(double)));
        HIPCHK(hipMemsetAsync(f->dWt, 0, (size_t)v.nt * TILE * TILE * sizeof(double), f->stream));   // zero above the diagonals, for good
        f->dWt_tiles = v.nt;
    }
    const int kskip = (f->pad0 / 16) * 16;
    // (the XCD-aware deal assumes the eight XCDs of the whole chip -- a partitioned device keeps the one counter -- and XCDs that
    // contribute comparable numbers of workgroups: a class of list positions whose XCD holds almost none is carried by the
    // others only while they are free to draw, and with everybody waiting at that class's tasks the launch crawls at the pace
    // of its few workgroups until a bounded wait ends it (tools/diag/quota_stress.py) -- a quota below half an XCD's share, which
    // only the tests set, keeps the one counter too)
    const int q_all = dag_xcc_quota();
    const int xcd_g = (tun().dag_xcd && device_cus() == 256 && (q_all == 0 || q_all >= tun().dag_xcd_min_quota)) ? 5 : 0;
    const int key[13] = {v.nt, v.mt, v.trim, kskip, tun().dag_lead, tun().dag_min_tiles, tun().dag_split, tun().dag_lead2, tun().dag_lead3,
                         tun().dag_order, xcd_g, tun().dag_bw, tun().dag_bh};
    if (memcmp(key, f->dag_key, sizeof key) != 0 || !f->ddag_steps) {
        std::vector<DagStepHost> steps;
        std::vector<unsigned> ftab;
        const bool want_tab = tun().dag_order != 0 || xcd_g > 0;
        const unsigned ntasks = dag_build_steps(v.nt, v.mt, v.trim, kskip, tun().dag_lead, tun().dag_min_tiles, tun().dag_split, steps,
                                                tun().dag_lead2, tun().dag_lead3, want_tab ? &ftab : nullptr, xcd_g,
                                                tun().dag_order ? tun().dag_bw : 0, tun().dag_bh);
        HIPCHK(hipStreamSynchronize(f->stream));
        if (f->stream2) HIPCHK(hipStreamSynchronize(f->stream2));
        if (f->ddag_steps) { HIPCHK(hipFree(f->ddag_steps)); f->ddag_steps = nullptr; }
        HIPCHK(hipMalloc(&f->ddag_steps, (steps.size() + 1) * sizeof(DagStepHost)));
        // (on the handle's own stream: the library never touches the NULL stream -- a synchronous hipMemcpy here gave it a
        // hardware queue of its own and shifted every later stream's assignment)
        HIPCHK(hipMemcpyAsync(f->ddag_steps, steps.data(), steps.size() * sizeof(DagStepHost), hipMemcpyHostToDevice, f->stream));
        if (f->ddag_ftab_words < ftab.size()) {
            if (f->ddag_ftab) { HIPCHK(hipFree(f->ddag_ftab)); f->ddag_ftab = nullptr; f->ddag_ftab_words = 0; }
            HIPCHK(hipMalloc(&f->ddag_ftab, ftab.size() * sizeof(unsigned)));
            f->ddag_ftab_words = ftab.size();
        }
        if (!ftab.empty())
            HIPCHK(hipMemcpyAsync(f->ddag_ftab, ftab.data(), ftab.size() * sizeof(unsigned), hipMemcpyHostToDevice, f->stream));
        f->dag_xcd_g = want_tab ? xcd_g : 0;
        f->dag_have_ftab = want_tab && !ftab.empty();
        HIPCHK(hipStreamSynchronize(f->stream));
        f->dag_nsteps = (int)steps.size(); f->dag_ntasks = ntasks;
        memcpy(f->dag_key, key, sizeof key);
        const size_t T64 = 2 * (size_t)v.mt;
        size_t words = 64 + T64 * (T64 + 1) / 2 + (steps.size() + 2) * T64 + steps.size() + 64 + 16 * (steps.size() + 2);
        words = (words + 31) / 32 * 32;
        f->ddag_xcnt_off = words;              // the XCDs' own task counters: eight cache lines behind everything else
        words += 8 * 32;
        if (f->ddag_words < words) {
            if (f->ddag) { HIPCHK(hipFree(f->ddag)); f->ddag = nullptr; }
            HIPCHK(hipMalloc(&f->ddag, words * sizeof(unsigned)));
            f->ddag_words = words;
        }
    }
    HIPCHK(hipMemsetAsync(f->ddag, 0, f->ddag_words * sizeof(unsigned), f->stream));
    // trace buffer: 4 stamps + one word of hw_where() pairs per task, 8 stamps per tile pair of the engine -- sized by BOTH
    // the task count and the tile count of THIS step table (a later table with fewer tasks and more tiles must not run past it)
    const size_t trace_elems = (size_t)f->dag_ntasks * 5 + 8 * (size_t)(v.nt + 2);
    if (tun().dag_trace && f->dag_trace_elems < trace_elems) {
        HIPCHK(hipStreamSynchronize(f->stream));
        if (f->stream2) HIPCHK(hipStreamSynchronize(f->stream2));
        if (f->ddag_trace) { HIPCHK(hipFree(f->ddag_trace)); f->ddag_trace = nullptr; f->dag_trace_elems = 0; }
        HIPCHK(hipMalloc(&f->ddag_trace, trace_elems * sizeof(unsigned long long)));
        f->dag_trace_elems = trace_elems;
    }
    f->dag_trace_tasks = (tun().dag_trace && f->ddag_trace) ? f->dag_ntasks : 0;      // 0: the buffer does not describe this table
    if (f->dag_trace_tasks)
        HIPCHK(hipMemsetAsync(f->ddag_trace, 0, trace_elems * sizeof(unsigned long long), f->stream));
    return 0;
}

// diagnostics: the step table and the per-task stamps of the last DAG factorisation of the handle (dag_trace = 1).
// steps_out: nsteps x 16 ints (DagStepHost); stamps_out: ntasks x 4 ticks of the 100 MHz clock.  Returns ntasks (or < 0);
// with null outputs only the sizes: *nsteps_out.  engine_out (may be null): 8 stamps per tile pair, (nt + 2) / 2 pairs ... room
// for 8 * (nt + 2) values (see EngineArgs::trace).
extern "C" long long cocons_debug_dag_trace(cocons_fit *f, int *nsteps_out, int *steps_out, unsigned long long *stamps_out,
                                            unsigned long long *engine_out)
{
    FIT_ENTER(f);
    if (!f->ddag_steps) return fail(-1, "cocons_debug_dag_trace: no DAG factorisation on this handle yet");
    if (nsteps_out) *nsteps_out = f->dag_nsteps;
    HIPCHK(hipStreamSynchronize(f->stream));
    // (asynchronous copies on the handle's own stream: a synchronous hipMemcpy runs on the NULL stream, gives it a hardware
    // queue and shifts every later stream's assignment -- the tool would perturb what it observes)
    if (steps_out)
        HIPCHK(hipMemcpyAsync(steps_out, f->ddag_steps, (size_t)f->dag_nsteps * sizeof(DagStepHost), hipMemcpyDeviceToHost, f->stream));
    if (stamps_out) {
        if (!f->ddag_trace || f->dag_trace_tasks != f->dag_ntasks)
            return fail(-1, "cocons_debug_dag_trace: tracing was off (cocons_debug_tune(\"dag_trace\", 1))");
        HIPCHK(hipMemcpyAsync(stamps_out, f->ddag_trace, (size_t)f->dag_ntasks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                              f->stream));
        if (engine_out)
            HIPCHK(hipMemcpyAsync(engine_out, f->ddag_trace + 4 * (size_t)f->dag_ntasks, 8 * (size_t)(f->nt + 2) * sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, f->stream));
    }
    HIPCHK(hipStreamSynchronize(f->stream));
    return (long long)f->dag_ntasks;
}

// (diagnostics) the first `count` task words of the handle's last dependency-driven factorisation: [0] the one task counter,
// [8..14] the record of a wait that ran out, [16..23] workgroups that took part per XCD, [32..39] the XCDs' own task counters
extern "C" int cocons_debug_dag_words(cocons_fit *f, int count, unsigned *out)
{
    FIT_ENTER(f);
    if (!f->ddag || !out || count < 1 || (size_t)count > f->ddag_words) return fail(-1, "cocons_debug_dag_words: bad argument or no DAG factorisation yet");
    HIPCHK(hipStreamSynchronize(f->stream));
    HIPCHK(hipMemcpyAsync(out, f->ddag, (size_t)count * sizeof(unsigned), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    return 0;
}

static int engine_start(cocons_fit *f, const FactorView &v)
{
    if (f->engine_live) return 0;
    const int nt = v.nt;
    hipStream_t M = f->stream;
    if (int rc = flags_reset(f, nt)) return rc;
    unsigned *in = f->dflags, *out = f->dflags + f->flags_cap, *xr = f->dflags + 2 * (size_t)f->flags_cap;
    f->dag_next = dag_wanted(f, v);
    if (f->dag_next) {
        if (int rc = dag_prepare(f, v)) return rc;
        if (f->dag_nsteps < 2) f->dag_next = false;          // too small a problem for a head worth the launch: classic throughout
    }
    f->engine_pair_live = tun().engine_pair != 0 ? 1 : 0;
    if (f->engine_pair_live || (tun().potrf_follow && !f->follow_off))
        if (int rc = mbox_reset(f, nt > f->nt ? nt : f->nt)) return rc;
    HIPCHK(hipEventRecord(f->ev_eng, M));                    // (behind the resets of the flag and task words, and of W / P when new)
    HIPCHK(hipStreamWaitEvent(f->stream2, f->ev_eng, 0));
    unsigned *alive_w = f->dflags + 3 * (size_t)f->flags_cap;
    f->engine_t0 = (tun().engine_block0 && !v.hi) ? 0 : 2;
    launch_potrf_engine(v.A, v.lda, f->engine_t0, nt, f->dinv, f->dinfo, in, out, xr, (unsigned *)(f->dinfo + 1),
                        alive_w, f->stream2, f->dag_next ? f->dWt : nullptr,
                        f->dag_next ? f->dP : nullptr, f->dag_next ? 2 * f->dag_nsteps : 0,
                        (f->dag_next && f->dag_trace_tasks) ? f->ddag_trace + 4 * (size_t)f->dag_ntasks : nullptr,
                        f->engine_pair_live ? f->dmbox : nullptr, tun().engine_in_wait_ms);
    f->engine_live = true;
    return 0;
}

// Bordered right-looking factorisation, outer block = 2 tiles (256 columns).
//
// Plain schedule (COCONS_ENGINE=0, or fewer than 5 tiles), everything on the main stream:
//   potrf(t) | trsm(t) | in-panel update of tile column t+1 | potrf(t+1) | trsm(t+1) | trailing update
//
// Engine schedule: the 256 x 256 diagonal blocks from tile 2 on are factored by ONE resident workgroup
// (potrf_engine_kernel, launched once per factorisation on the second stream, on a CU of its own) as
// soon as the trailing update has published them; the main stream runs, per block k with t = k + 2:
//   U(k)      : trailing update with panel k, tile column t first; its workgroups inside the diagonal
//               block raise in[t] / in[t+1]      -> the engine factors tile t, forms X = A(t+1,t) L(t)^-T,
//                                                   updates and factors tile t+1 (all while U(k) runs)
//   trsm(t)   : rows below the diagonal block; waits for out[t]
//   in-panel update of tile column t+1 (rows below the block) with column t; waits for xr[t]
//   trsm(t+1) : waits for out[t+1]
// So the serial part of every panel -- two 30 us single-workgroup factorisations and the tile between
// them -- is off the critical path as long as U(k) lasts ~90 us; the main stream needs no events and
// issues fewer launches than the plain schedule.
static int factorize(cocons_fit *f, const FactorView &v, std::vector<hipEvent_t> *ev_upd)
{
    const int nt = v.nt, mt = v.mt;
    hipStream_t M = f->stream;
    // the placeholder observations in front (fit_create_impl): whatever the assembly kernels put into their columns --
    // covariances, right-hand sides, cross-covariance rows -- is replaced by unit vectors, in every view whose first
    // indices are the handle's observations
    launch_front_identity(v.A, v.lda, f->pad0, mt * TILE, M);
    f->follow_used = false;
    if (!engine_wanted(f, v)) {
        f->engine_used = false;
        f->dag_used = false;
        if (int rc = flags_reset(f, nt)) return rc;
        if (tun().potrf_follow && !f->follow_off)
            if (int rc = mbox_reset(f, nt > f->nt ? nt : f->nt, false)) return rc;
        if (v.hi) {
            // band-limited: one tile column per step (factor, solve, update with K = 128) -- inside a narrow envelope
            // the in-panel update of the two-tile block costs more than the second, cheaper trailing update
            // (4.71 -> 4.53 ms at n = 10^4)
            // (row bound = the column's OWN envelope hi[k] -- per 256-block, monotone, >= k + 1 --, not band_hi(): for an
            // odd k that is the NEXT block's bound, beyond what band_zero_kernel clears in column k)
            // (packed band buffer: the kernels that stay inside tile column k address it by global indices through a
            // shifted base, where the rows under the matrix start at row (k + skew) * 128 -- kernels.h band_base)
            for (int k = 0; k < nt; ++k) {
                const int hb = v.hi[k] < nt ? v.hi[k] : nt;
                double *q = f->dinv + (size_t)(k & 1) * 2048;
                double *Ak = band_base(v.A, k, v.skew);
                const int e0 = v.skew ? k + v.skew : nt, e1 = e0 + (mt - nt);      // tile rows under the matrix
                potrf_solve(f, Ak, v.lda, k, (k + 1) * TILE, e1 * TILE - 64 * v.trim, q, M, hb * TILE, e0 * TILE);
                if (k + 1 < nt)
                    launch_update(v.A, v.lda, k * TILE, TILE, k + 1, mt, k + 1, hb < nt ? hb : nt, true, M, nullptr, -1,
                                  nullptr, nullptr, nullptr, hb, nt, v.skew, v.trim);
            }
            return 0;
        }
        for (int k = 0; k < nt; k += 2) {
            panel_ops(f, v, k, M);
            if (k + 2 < nt) {
                if (ev_upd) count_update_flops(f, 2, k + 2);
                timed_update(f, v, k, 2, k + 2, nt, M, ev_upd, nullptr, -1, tile_queue(f, k));
            }
        }
        return 0;
    }
    if (int rc = engine_start(f, v)) return rc;          // no-op when enqueue_eval started it before the assembly
    f->engine_live = false;
    f->engine_used = true;
    unsigned *in = f->dflags, *out = f->dflags + f->flags_cap, *xr = f->dflags + 2 * (size_t)f->flags_cap;
    unsigned *abort_word = (unsigned *)(f->dinfo + 1);
    unsigned *alive = f->dflags + 3 * (size_t)f->flags_cap;
    if (tun().gate_sabotage > 0) { --tun().gate_sabotage; alive += 1; }      // (tests: a word that stays zero)
    const bool block0 = f->engine_t0 == 0;           // the engine factors the first diagonal block too: the gate raises its input words
    launch_engine_gate(alive, abort_word, M, false, f->engine_ops++ == 0,
                       f->engine_pair_live,       // (the pair partner counts itself)
                       (block0 && tun().gate_sabotage == 0 && alive == f->dflags + 3 * (size_t)f->flags_cap) ? in : nullptr);
    const int rend = mt * TILE - 64 * v.trim;         // one past the last row any panel kernel touches
    // the panel of the block at tile t (behind the engine's factorisation of it): one launch or three; returns the tiles of the NEXT
    // diagonal block that the launch has updated (the update launch behind it then leaves them alone)
    auto panel_for = [&](int t, bool allow_diag) -> int {
        const bool two = t + 1 < nt;                 // the block has a second tile
        const int r0 = two ? t + 2 : t + 1;          // first tile row below the diagonal block
        const int hb = band_hi(v, t);                // rows of block t's panel: [r0, hb) and the rows under the matrix
        const int br = hb >= 0 ? hb * TILE : -1, er = nt * TILE;
        if (two && hb < 0 && tun().panel_fused) {
            const bool fol = f->engine_pair_live && tun().panel_follow && f->dmbox != nullptr;
            // the next diagonal block (tiles t + 2, t + 3), when there is one, is updated inside this launch
            const int next_tiles = t + 2 < nt ? (t + 3 < nt ? 2 : 1) : 0;
            const bool dg = allow_diag && fol && tun().panel_diag && next_tiles > 0 &&
                            ((size_t)(t >> 1) + 1) * PANEL_SMBOX_DOUBLES <= f->smb_elems;
            const int nstrips = (rend - r0 * TILE) / 64;
            const bool sp = fol && tun().panel_split > 0 && nstrips >= tun().panel_split &&
                            (size_t)nstrips * PANEL_XMBOX_DOUBLES <= f->xmb_elems;
            launch_panel_pair(v.A, v.lda, t * TILE, r0 * TILE, rend, f->dinv + (size_t)(t & 1) * 2048,
                              f->dinv + (size_t)((t + 1) & 1) * 2048, out + t, xr + t, out + t + 1, abort_word, M,
                              fol ? f->dmbox + (size_t)t * ENGINE_MBOX_DOUBLES : nullptr,
                              fol ? f->dmbox + (size_t)(t + 1) * ENGINE_MBOX_DOUBLES : nullptr,
                              dg ? f->dmbox + f->smb_off + (size_t)(t >> 1) * PANEL_SMBOX_DOUBLES : nullptr,
                              dg ? (next_tiles == 2 ? 10 : 3) : 0, in, t + 2, sp ? f->dmbox + f->xmb_off : nullptr);
            return (dg && nstrips >= (next_tiles == 2 ? 4 : 2)) ? next_tiles : 0;
        }
        launch_trsm_tile(v.A, v.lda, t * TILE, r0 * TILE, rend, f->dinv + (size_t)(t & 1) * 2048, M,
                         out + t, abort_word, br, er);
        if (two) {
            launch_update(v.A, v.lda, t * TILE, TILE, r0, mt, t + 1, t + 2, false, M, nullptr, -1, xr + t, abort_word,
                          nullptr, hb, nt, 0, v.trim);
            launch_trsm_tile(v.A, v.lda, (t + 1) * TILE, r0 * TILE, rend,
                             f->dinv + (size_t)((t + 1) & 1) * 2048, M, out + t + 1, abort_word, br, er);
        }
        return 0;
    };
    int diag_done = 0;                                // tiles of the diagonal block at t that the previous panel's launch has updated
    if (block0) diag_done = panel_for(0, !f->dag_next);          // (the persistent launch updates its first diagonal block itself)
    else panel_ops(f, v, 0, M);
    f->dag_used = f->dag_next;
    int k_first = 0;                 // first block step the classic loop below runs in full
    if (f->dag_next) {
        // the head of the factorisation -- the steps whose update fills the chip several times over -- is ONE persistent
        // launch (dag_kernel): tiles of update k + 1 start as soon as the strips of panel k + 1 they need and their own tile of
        // update k are done, and the panels between them are tile tasks of the same launch (the engine publishes the tile
        // inverses they multiply with).  Behind the head a step is bound by its dependency chain, and there the classic
        // sequence below has the shorter one: it takes over with the panel behind the last DAG step.
        const size_t T64 = 2 * (size_t)mt;
        unsigned *queue = f->ddag, *tdone = f->ddag + 64, *pdone = tdone + T64 * (T64 + 1) / 2;
        unsigned *pall = pdone + ((size_t)f->dag_nsteps + 2) * T64;
        unsigned *dcount = pall + (size_t)f->dag_nsteps + 64;
        hipEvent_t ea = nullptr, eb = nullptr;
        if (ev_upd) {
            const double before = f->upd_flops;
            for (int s = 0; s < f->dag_nsteps; ++s) count_update_flops(f, 2, 2 * s + 2);
            f->dag_flops = f->upd_flops - before;
            hipEventCreate(&ea); hipEventCreate(&eb);
            hipEventRecord(ea, M);
        }
        launch_dag(v.A, v.lda, f->dP, f->dWt, (const DagStepHost *)f->ddag_steps, f->dag_nsteps, f->dag_ntasks, queue, tdone,
                   pdone, (int)T64, pall, f->dpart, dcount, in, out, xr, abort_word, M, f->dag_trace_tasks ? f->ddag_trace : nullptr,
                   alive, dag_xcc_quota(), f->dag_trace_tasks ? (unsigned *)(f->ddag_trace + 4 * (size_t)f->dag_ntasks + 8 * (size_t)(v.nt + 2)) : nullptr,
                   f->dag_have_ftab ? f->ddag_ftab : nullptr, f->dag_xcd_g, f->ddag + f->ddag_xcnt_off);
        if (ev_upd) { hipEventRecord(eb, M); ev_upd->push_back(ea); ev_upd->push_back(eb); f->dag_events = 1; }
        k_first = 2 * f->dag_nsteps;
    }
    // (Running the panel kernels on a stream of their own behind near-tile flags, so that they start in the tail of the
    // update that feeds them, was built and measured in round 3: slower -- a 90 KB-LDS solve is not placed beside eight
    // update workgroups per CU, the event back to the main stream costs 12 us -- and removed; so were three forms of the
    // panel as products with explicit inverses published by the engine (round 3's COCONS_PANEL_MODE 1-3: +-1 %, deleted
    // in round 4))
    for (int k = k_first > 0 ? k_first - 2 : 0; k + 2 < nt; k += 2) {
        const int t = k + 2;
        // (tests: a late host -- whatever raises in[host_delay_tile], the panel launch of block t or the update launch behind it,
        // is enqueued host_delay_us late, while the engine has everything it needs to get there and wait)
        if (tun().host_delay_us > 0 && t + 2 == tun().host_delay_tile) usleep((useconds_t)tun().host_delay_us);
        if (k >= k_first) {                          // (the update with the last DAG step's panel was that launch's)
            if (ev_upd) count_update_flops(f, 2, t);
            timed_update(f, v, k, 2, t, nt, M, ev_upd, in, t, tile_queue(f, k), diag_done);
        }
        diag_done = panel_for(t, k + 4 < nt);
    }
    // no rows under the matrix (right-hand sides in the slots of the last tile): nothing on the main stream has waited for
    // the engine's last tile yet -- what follows (the reductions) must
    if (mt == nt) launch_engine_gate(out + (nt - 1), abort_word, M, true);
    return 0;
}

static int reset_info(cocons_fit *f)
{
    HIPCHK(hipMemcpyAsync(f->dinfo, f->hinfo_init, 2 * sizeof(int), hipMemcpyHostToDevice, f->stream));
    return 0;
}

// enqueue one full evaluation with nrhs right-hand-side rows; results land in hout/hinfo
static int enqueue_eval_impl(cocons_fit *f, const double *theta, const double *mean, bool use_trend,
                             const double *xb, int nxb, std::vector<hipEvent_t> *ev_upd, bool stage_events);
static int enqueue_eval(cocons_fit *f, const double *theta, const double *mean, bool use_trend,
                        const double *xb, int nxb, std::vector<hipEvent_t> *ev_upd, bool stage_events)
{
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int rc = enqueue_eval_impl(f, theta, mean, use_trend, xb, nxb, ev_upd, stage_events);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    f->enq_host_us += (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
    f->enq_calls++;
    return rc;
}

// (diagnostics) out[0] = mean host microseconds per enqueued evaluation of this handle and its batch slots, out[1] = evaluations
extern "C" int cocons_debug_host_enqueue(cocons_fit *f, double *out)
{
    if (!f || !out) return fail(-1, "cocons_debug_host_enqueue: null argument");
    double us = f->enq_host_us; long long n = f->enq_calls;
    if (f->slots) for (cocons_fit *c : *f->slots) { us += c->enq_host_us; n += c->enq_calls; }
    out[0] = n ? us / (double)n : 0.0; out[1] = (double)n;
    return 0;
}

static int enqueue_eval_impl(cocons_fit *f, const double *theta, const double *mean, bool use_trend,
                             const double *xb, int nxb, std::vector<hipEvent_t> *ev_upd, bool stage_events)
{
    const int nrhs = f->r + nxb;
    f->nrhs_cur = nrhs;
    // the padding rows under the right-hand sides (a 128-row tile holds them; r + q of its rows are used) only have to
    // be zeroed when they may hold something else: zero rows stay exactly zero through a factorisation that succeeds
    // (10 MB of strided stores, 22 us on the critical path of every evaluation at n = 10^4)
    const int was_clean = f->border_clean;
    const double *was_A = f->dA;
    if (int rc = fit_alloc_matrix(f, nrhs)) return rc;
    const bool zero_rest = !(was_clean == nrhs && was_A == f->dA);
    f->border_pending = nrhs;
    if (stage_events) hipEventRecord(f->ev[0], f->stream);
    if (int rc = reset_info(f)) return rc;
    if (f->taper_nnz > 0) { if (int rc = assemble_sigma_taper(f, theta)) return rc; }
    else assemble_sigma(f, theta, 0, 0, f->npad);
    // the engine becomes resident while the (short) right-hand-side kernel runs: not earlier -- a second
    // queue with a resident kernel cuts the workgroup dispatch rate of every other launch to a quarter
    // (tools/diag/occupancy_probe.hip), which costs the 12,000-workgroup assembly 6 % -- and not later, see engine_start
    const bool slots = f->nslot >= nrhs && f->nslot > 0;      // the right-hand sides ride in the matrix's last tile
    FactorView fv = main_view(f);
    if (slots) fv.mt = fv.nt;                         // no rows under the matrix
    else fv.trim = (f->rhs_act - nrhs >= 64) ? 1 : 0; // at most 64 of the 128 rows under the matrix are in use
    fv.dag_ok = true;                                 // (the reductions below read the factor from both buffers)
    if (engine_wanted(f, fv))
        if (int rc = engine_start(f, fv)) return rc;
    assemble_rhs(f, mean, use_trend, xb, nxb, 0, f->npad, zero_rest, slots);
    if (stage_events) hipEventRecord(f->ev[1], f->stream);
    if (int rc = factorize(f, fv, ev_upd)) return rc;
    if (stage_events) hipEventRecord(f->ev[2], f->stream);
    launch_finalize(f->dA, f->lda, f->n, slots ? f->n : f->npad, nrhs, f->dout, f->stream, f->skew, f->npad,
                    f->dag_used ? f->dP : nullptr, f->dag_used ? 2 * TILE * f->dag_nsteps : 0);
    HIPCHK(hipMemcpyAsync(f->hinfo, f->dinfo, (size_t)(2 + nrhs * nrhs) * sizeof(double),       // info words + outputs
                          hipMemcpyDeviceToHost, f->stream));
    if (stage_events) hipEventRecord(f->ev[3], f->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

static int info_status(cocons_fit *f)
{
    // COCONS_DEBUG_ABORT=1: say which wait gave up (0x1tt / 0x2tt engine waiting for tile tt, 0x3tt panel solve
    // waiting for the engine's tile tt, 0x5.. in-panel update, 0x600 the gate waiting for the engine to be resident,
    // 0x900 the reductions waiting for the engine's last tile: kernels.h, abort_code / abort_class)
    if (f->hinfo[1] != 0 && getenv("COCONS_DEBUG_ABORT")) {
        fprintf(stderr, "cocons: hand-off time-out, code 0x%x\n", f->hinfo[1]);
        if (f->dag_used && f->ddag && (f->hinfo[1] & 0xf00) >= 0xa00) {
            // a wait of the DAG launch: what it waited for (dag_wait's record) and what the word holds NOW
            unsigned rec[7] = {0, 0, 0, 0, 0, 0, 0}, now = 0, qn = 0;
            hipMemcpyAsync(rec, f->ddag + 8, sizeof rec, hipMemcpyDeviceToHost, f->stream);
            hipStreamSynchronize(f->stream);
            if (rec[2] < f->ddag_words) hipMemcpyAsync(&now, f->ddag + rec[2], sizeof now, hipMemcpyDeviceToHost, f->stream);
            hipMemcpyAsync(&qn, f->ddag, sizeof qn, hipMemcpyDeviceToHost, f->stream);
            hipStreamSynchronize(f->stream);
            fprintf(stderr, "cocons: DAG wait: task %u (of %u, counter now %u) code 0x%x waited for word %u >= %u, saw %u, holds %u now; "
                    "%.1f ms, %u polls\n", rec[0], f->dag_ntasks, qn, rec[1], rec[2], rec[3], rec[4], now, rec[5] * 1e-5, rec[6]);
            if (const char *dump = getenv("COCONS_DEBUG_ABORT_DUMP")) {
                // everything an offline look needs (tools/dag_abort.py): header, record, step table, all task words, the engine's
                // flag words, and with tracing on the stamps of every task and of the engine
                static int ndump = 0;
                char path[512];
                snprintf(path, sizeof path, "%s.%d", dump, ndump++);
                if (FILE *fp = fopen(path, "wb")) {
                    const unsigned ntr = f->dag_trace_tasks ? 1u : 0u;
                    unsigned hdr[16] = {0xDA6D0001u, (unsigned)f->nt, (unsigned)f->dag_nsteps, f->dag_ntasks, (unsigned)f->ddag_words,
                                        (unsigned)f->flags_cap, ntr, (unsigned)f->hinfo[1], qn, now, 0, 0, 0, 0, 0, 0};
                    fwrite(hdr, sizeof hdr, 1, fp);
                    fwrite(rec, sizeof rec, 1, fp);
                    std::vector<DagStepHost> sh((size_t)f->dag_nsteps);
                    hipMemcpyAsync(sh.data(), f->ddag_steps, sh.size() * sizeof(DagStepHost), hipMemcpyDeviceToHost, f->stream);
                    std::vector<unsigned> w((size_t)f->ddag_words), fl(4 * (size_t)f->flags_cap + 64);
                    hipMemcpyAsync(w.data(), f->ddag, w.size() * sizeof(unsigned), hipMemcpyDeviceToHost, f->stream);
                    hipMemcpyAsync(fl.data(), f->dflags, fl.size() * sizeof(unsigned), hipMemcpyDeviceToHost, f->stream);
                    hipStreamSynchronize(f->stream);
                    fwrite(sh.data(), sizeof(DagStepHost), sh.size(), fp);
                    fwrite(w.data(), sizeof(unsigned), w.size(), fp);
                    fwrite(fl.data(), sizeof(unsigned), fl.size(), fp);
                    if (ntr) {
                        std::vector<unsigned long long> st((size_t)f->dag_ntasks * 5 + 8 * (size_t)(f->nt + 2));
                        hipMemcpyAsync(st.data(), f->ddag_trace, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, f->stream);
                        hipStreamSynchronize(f->stream);
                        fwrite(st.data(), sizeof(unsigned long long), st.size(), fp);
                    }
                    fclose(fp);
                    fprintf(stderr, "cocons: state written to %s\n", path);
                }
            }
            if (f->dag_trace_tasks && !getenv("COCONS_DEBUG_ABORT_DUMP")) {
                // which tasks were drawn and never finished (stamps: drawn, inputs complete, product done, stored)
                std::vector<unsigned long long> st((size_t)f->dag_ntasks * 4);
                hipMemcpyAsync(st.data(), f->ddag_trace, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, f->stream);
                hipStreamSynchronize(f->stream);
                unsigned long long tmin = ~0ull;
                for (size_t i = 0; i < st.size(); i += 4) if (st[i] && st[i] < tmin) tmin = st[i];
                int shown = 0;
                for (unsigned L = 0; L < f->dag_ntasks && shown < 40; ++L) {
                    const unsigned long long *q = &st[(size_t)L * 4];
                    if (q[0] && !q[3]) {
                        fprintf(stderr, "   unfinished task %u: drawn %.1f us, inputs %s, product %s\n", L, (q[0] - tmin) * 0.01,
                                q[1] ? "complete" : "WAITING", q[2] ? "done" : "-");
                        ++shown;
                    }
                }
            }
        }
    }
    if (f->hinfo[1] != 0)
        return fail(ENGINE_ABORT, "hand-off between the diagonal-tile engine and the main stream timed out");
    // the operation ran to its end on the schedule factorize chose: book-keeping of the engine's back-off
    f->engine_active_last = f->engine_used;
    if (f->engine_used) f->engine_fails = 0;
    else if (f->engine_skip > 0) --f->engine_skip;
    int info = f->hinfo[0];
    if (info != 0x7f7f7f7f) {
        info -= f->pad0;                // (minors are counted in the handle's internal order, behind its front padding)
        if (info < 1) info = 1;
        if (info > f->n_user) info = f->n_user;   // failure reported inside the identity padding cannot happen; clamp anyway
        g_err = "leading minor not positive";
        return info;
    }
    f->border_clean = f->border_pending;      // every pivot positive and finite: zero rows are still zero
    f->border_pending = -1;
    return 0;
}

// The engine could not be scheduled in time, or one of its partners could not (another process or stream kept
// every CU busy, or a profiler serialises kernels): the caller repeats THIS operation once on the plain schedule;
// the handle stays on it for a few more operations (2, 4, ... 64 with consecutive time-outs) and then tries the
// engine again.  Every time-out is counted (cocons_fit_engine_state).
static bool engine_retry(cocons_fit *f, int st)
{
    if (st != ENGINE_ABORT || !(f->engine_used || f->follow_used)) return false;
    f->engine_retries++;
    f->engine_last_abort = f->hinfo[1];
    if (!f->engine_used || abort_class((unsigned)f->hinfo[1]) == ABORT_FOLLOW) f->follow_off = true;      // a follower of potrf_follow_kernel gave up: two launches from now on
    if (f->engine_fails < 6) f->engine_fails++;
    f->engine_skip = 1 << f->engine_fails;
    f->engine_live = false;
    f->engine_used = false;
    if (f->stream2) hipStreamSynchronize(f->stream2);
    return true;
}

// out[0] = 1 if the last completed operation of the handle ran on the engine schedule, out[1] = hand-off time-outs
// so far (each was followed by a repeat on the plain schedule), out[2] = abort code of the last one (0 = none)
extern "C" int cocons_fit_engine_state(cocons_fit *f, int *out)
{
    if (!f || !out) return fail(-1, "cocons_fit_engine_state: null argument");
    out[0] = f->engine_active_last ? 1 : 0;
    out[1] = f->engine_retries;
    out[2] = f->engine_last_abort;
    // (the slots of cocons_neg2loglik_batch are handles of their own: their time-outs count for this handle)
    if (f->slots)
        for (cocons_fit *c : *f->slots) {
            out[1] += c->engine_retries;
            if (!out[2]) out[2] = c->engine_last_abort;
        }
    return 0;
}

static const double LOG_2PI = 1.8378770664093454835606594728112;
static void dense_collect(cocons_fit *f, double *sum_logliks, double *parts);
static int sharded_eval(cocons_fit *f, const double *theta, const double *mean, double *sum_logliks, double *parts);

extern "C" int cocons_neg2loglik_dense(cocons_fit *f, const double *theta, const double *mean,
                                       double *sum_logliks, double *parts)
{
    FIT_ENTER(f);
    if (!theta || !mean || !sum_logliks) return fail(-1, "cocons_neg2loglik_dense: null argument");
    if (f->r < 1) return fail(-1, "cocons_neg2loglik_dense: fit has no z");
    if (f->coll_kind < 0) return fail(-7, "cocons_neg2loglik_dense: the communicator of this fit was aborted after an error");
    if (f->coll_kind) return sharded_eval(f, theta, mean, sum_logliks, parts);    // (also with one rank: the caller asked for it)
    for (;;) {
        if (int rc = enqueue_eval(f, theta, mean, true, nullptr, 0, nullptr, false)) return rc;
        HIPCHK(hipStreamSynchronize(f->stream));
        int st = info_status(f);
        if (engine_retry(f, st)) continue;
        if (st) return st;
        break;
    }
    dense_collect(f, sum_logliks, parts);
    return 0;
}

static void dense_collect(cocons_fit *f, double *sum_logliks, double *parts)
{
    const int nr = f->r;
    double logdet = f->hout[0], total = 0.0;
    for (int k = 0; k < nr; ++k) {                                   // R/neg2loglikelihood.R:212-218
        double quad = f->hout[1 + k * nr + k];
        total += f->n_user * LOG_2PI + 2 * logdet + quad;
        if (parts) parts[1 + k] = quad;
    }
    if (parts) parts[0] = logdet;
    *sum_logliks = total;
}

// Batch of independent evaluations (the 2P finite-difference points of one L-BFGS-B gradient,
// R/optim.R:237-259 + R/profile.R:11-12, or getHessian's 3 P (P+1)/2 points,
// R/getFunctions.R:979-1016): evaluation i runs on slot i mod S, each slot a clone of the fit
// with its own factorisation buffer and stream, so the latency-bound panel chain of one
// evaluation overlaps the MFMA-bound updates and the VALU-bound assembly of the others.
// thetas: nb x (6 p) row-major tables; means: nb x p; values[nb]; status[nb] (0 / k>0 like the
// single call).  Returns 0 unless a HIP / argument error occurred.
// A second handle over the same data with its own factorisation buffer and streams: one slot of
// cocons_neg2loglik_batch.  A taper handle's clone shares nothing on the device (pattern and taper entries are
// copied device to device) and keeps the order and the envelope of its original.
// A slot's main stream must run BESIDE the main streams of the handle and of its other slots (kernels of streams that share
// a hardware queue run one after the other): tested like the engine's stream, redrawn on a clash.
static void slot_stream_apart(cocons_fit *f, cocons_fit *c)
{
    if (!c->own_stream || !c->dflags) return;
    unsigned *words = c->dflags + 3 * (size_t)c->flags_cap + 8;
    std::vector<hipStream_t> losers;
    for (int attempt = 0; attempt < 8; ++attempt) {
        bool clash = false;
        std::vector<cocons_fit *> peers{f};
        if (f->slots) for (cocons_fit *o : *f->slots) if (o != c) peers.push_back(o);
        for (cocons_fit *o : peers) {
            if (hipStreamQuery(o->stream) != hipSuccess) { (void)hipGetLastError(); continue; }      // (busy: not probed)
            const int r = streams_run_concurrently(o->stream, c->stream, words);
            if (r == 0) { clash = true; break; }
            if (r < 0) { (void)hipGetLastError(); break; }
        }
        if (!clash) break;
        losers.push_back(c->stream);
        c->stream = nullptr;
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { c->stream = losers.back(); losers.pop_back(); break; }
    }
    for (hipStream_t l : losers) hipStreamDestroy(l);
    if (c->stream2 && streams_run_concurrently(c->stream2, c->stream, words) == 0) c->engine_ok = false;     // (its own pair again)
    (void)hipMemsetAsync(c->dflags, 0, (4 * (size_t)c->flags_cap + 64) * sizeof(unsigned), c->stream);
    (void)hipStreamSynchronize(c->stream);
}

static cocons_fit *clone_for_slot(cocons_fit *f, bool want_engine)
{
    // (a clone enters the registry with its operation lock held -- fit_create_impl(..., return_locked) -- and keeps it until it
    // is complete: its main stream may still be redrawn below, which no other thread's stream self-test may see half done)
    if (f->taper_nnz <= 0) {
        cocons_fit *c = fit_create_impl(f->n_user, f->p, f->r, 0, f->h_locs->data(), f->h_X->data(), f->h_z->data(), nullptr,
                                        f->smooth_limits, f->device, true, false, want_engine, true);
        if (c) {
            if (!c->dflags && flags_reset(c, c->nt) != 0) { c->op_mu->unlock(); cocons_fit_destroy(c); return nullptr; }
            hipStreamSynchronize(c->stream);
            slot_stream_apart(f, c);
            c->engine_ok = c->engine_ok && c->stream2 != nullptr;
            c->op_mu->unlock();
        }
        return c;
    }
    cocons_fit *c = fit_create_impl(f->n_user, f->p, f->r, 0, f->h_locs->data(), f->h_X->data(), f->h_z->data(), nullptr,
                                    f->smooth_limits, f->device, false, true, false, true);      // h_* of a taper handle are in ITS order;
    if (!c) return nullptr;                                                                  // band-limited: never an engine
    struct Unlock { cocons_fit *c; ~Unlock() { if (c) c->op_mu->unlock(); } } unlock{c};
    auto drop = [&]() { unlock.c = nullptr; c->op_mu->unlock(); cocons_fit_destroy(c); return (cocons_fit *)nullptr; };
    if (!c->dflags && flags_reset(c, c->nt) != 0) return drop();
    hipStreamSynchronize(c->stream);
    slot_stream_apart(f, c);
    c->skew = f->skew;                        // the same (packed) buffer layout as the original
    if (fit_alloc_matrix(c, f->r + f->p) != 0) return drop();
    const size_t nnz = (size_t)f->taper_nnz;
    bool ok = hipMalloc(&c->d_tci, nnz * sizeof(int)) == hipSuccess &&
              hipMalloc(&c->d_trp, (size_t)(f->n + 1) * sizeof(int)) == hipSuccess &&
              hipMalloc(&c->d_tval, nnz * sizeof(double)) == hipSuccess &&
              // (asynchronous on the clone's own stream: the library makes no call on the NULL stream)
              hipMemcpyAsync(c->d_tci, f->d_tci, nnz * sizeof(int), hipMemcpyDeviceToDevice, c->stream) == hipSuccess &&
              hipMemcpyAsync(c->d_trp, f->d_trp, (size_t)(f->n + 1) * sizeof(int), hipMemcpyDeviceToDevice, c->stream) == hipSuccess &&
              hipMemcpyAsync(c->d_tval, f->d_tval, nnz * sizeof(double), hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
    c->taper_hi = new std::vector<int>(*f->taper_hi);
    c->taper_inv = new std::vector<int>(*f->taper_inv);
    c->taper_maxband = f->taper_maxband;
    if (ok && f->d_thi)
        ok = hipMalloc(&c->d_thi, (size_t)f->nt * sizeof(int)) == hipSuccess &&
             hipMemcpyAsync(c->d_thi, f->d_thi, (size_t)f->nt * sizeof(int), hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
    if (ok) ok = hipStreamSynchronize(c->stream) == hipSuccess;
    if (!ok) return drop();
    c->taper_nnz = f->taper_nnz;
    // a band-limited handle never uses the engine, nor do its clones; a taper handle WITHOUT an envelope (COCONS_TAPER_BAND=0)
    // may -- but this clone was created without an engine stream (want_engine = false), and an engine launched on a null
    // stream would land on the NULL stream the library never touches (round 5's regression, the advisor's finding)
    c->engine_ok = f->engine_ok && c->stream2 != nullptr;
    return c;
}

extern "C" int cocons_neg2loglik_batch(cocons_fit *f, int nb, const double *thetas, const double *means,
                                       double *values, int *status)
{
    FIT_ENTER(f);
    if (nb < 0 || (nb > 0 && (!thetas || !means || !values || !status)))
        return fail(-1, "cocons_neg2loglik_batch: bad argument");
    if (f->r < 1) return fail(-1, "cocons_neg2loglik_batch: fit has no z");
    // How many evaluations in flight, and on which schedule (round 5, from the kernel trace of a batch,
    // tools/diag/batch_trace.py): a process's streams share FOUR hardware queues, and kernels of streams that share one run one
    // after the other.  A resident engine holds its queue for the whole evaluation, so two engine-schedule evaluations -- 2
    // main + 2 engine streams -- are all that fits; a third runs behind one of them (n = 4096: 862 evaluations/s with two
    // slots, 695 with three).  On the plain schedule a slot needs ONE queue -- once the slots no longer create engine streams
    // they never use, which had put four slots' main streams on two queues (one queue busy 88 % of the time in the trace) --
    // and four evaluations in flight beat two with engines at every size measured: n = 4096 1067 against 862 evaluations/s,
    // n = 10^4 134 against 128 (33-point gradient).  More hardware queues (GPU_MAX_HW_QUEUES = 8, 12) make it WORSE (642 ... 996
    // at n = 4096): four is what the chip runs side by side.  COCONS_BATCH_ENGINE=1 (two engine slots) and COCONS_BATCH_SLOTS override.
    static int nslots_env = -1, batch_engine = -2;
    if (nslots_env < 0) {
        const char *e = getenv("COCONS_BATCH_SLOTS");
        nslots_env = e ? atoi(e) : 0;
        if (nslots_env < 0) nslots_env = 0;
        if (nslots_env > 8) nslots_env = 8;
        const char *e2 = getenv("COCONS_BATCH_ENGINE");
        batch_engine = e2 ? (atoi(e2) ? 1 : 0) : -1;
    }
    const bool eng_mode = batch_engine >= 0 ? batch_engine == 1 : false;
    const int nslots = nslots_env > 0 ? nslots_env : (eng_mode ? 2 : 4);
    int S = nb < nslots ? (nb > 0 ? nb : 1) : nslots;
    if (!f->slots) f->slots = new std::vector<cocons_fit *>();
    // every extra slot is a clone of the handle with its own n x n factorisation buffer
    // (lda * npad * 8 bytes: 0.83 GB at n = 10^4); if one cannot be created (out of memory) the
    // batch runs on the slots that exist -- slot 0 is the fit itself, so it always completes
    // (the slots are handles of their own in the registry: held for the whole call, like the handle itself, so that no other
    // thread's stream self-test launches probe kernels between their evaluations)
    std::vector<std::unique_lock<std::recursive_mutex>> slot_locks;
    for (cocons_fit *c : *f->slots) slot_locks.emplace_back(*c->op_mu);
    while ((int)f->slots->size() < S - 1) {
        cocons_fit *c = clone_for_slot(f, eng_mode);
        if (!c) { (void)hipGetLastError(); break; }
        slot_locks.emplace_back(*c->op_mu);
        f->slots->push_back(c);
    }
    const bool engine_saved = f->engine_ok;
    if (!eng_mode) {
        if (S > 1) f->engine_ok = false;
        for (auto c : *f->slots) c->engine_ok = false;
    }
    if (S > (int)f->slots->size() + 1) S = (int)f->slots->size() + 1;
    for (int i = 0; i < nb; ++i) { values[i] = NAN; status[i] = -1; }   // never left unwritten
    std::vector<int> pending(S, -1);
    const int tp = 6 * f->p;
    int rc_all = 0;
    auto slot_of = [&](int s) { return s == 0 ? f : (*f->slots)[s - 1]; };
    auto collect = [&](int s) -> int {
        cocons_fit *c = slot_of(s);
        int i = pending[s];
        if (i < 0) return 0;
        pending[s] = -1;
        HIPCHK(hipStreamSynchronize(c->stream));
        int st = info_status(c);
        if (engine_retry(c, st)) {           // hand-off time-out: this evaluation again, on the plain schedule
            if (int rc = enqueue_eval(c, thetas + (size_t)i * tp, means + (size_t)i * f->p, true, nullptr, 0, nullptr, false))
                return rc;
            HIPCHK(hipStreamSynchronize(c->stream));
            st = info_status(c);
        }
        status[i] = st;
        if (st == 0) dense_collect(c, &values[i], nullptr);
        else values[i] = NAN;
        return 0;
    };
    for (int i = 0; i < nb; ++i) {
        int s = i % S;
        if (int rc = collect(s)) { rc_all = rc; break; }
        cocons_fit *c = slot_of(s);
        if (int rc = enqueue_eval(c, thetas + (size_t)i * tp, means + (size_t)i * f->p, true, nullptr, 0, nullptr, false)) {
            rc_all = rc;
            break;
        }
        pending[s] = i;
    }
    for (int s = 0; s < S; ++s)
        if (int rc = collect(s)) rc_all = rc_all ? rc_all : rc;
    f->engine_ok = engine_saved;
    return rc_all;
}

// small dense SPD solve on the host (q x q, q <= COCONS_P_MAX): W = C C^T, returns
// sum(log(diag(C))) and solves W x = b in place for nb right-hand sides.
static int host_spd_solve(int q, std::vector<double> &W, int nb, std::vector<double> &B, double *logdet_half)
{
    double ld = 0;
    for (int j = 0; j < q; ++j) {
        double d = W[j + j * q];
        for (int k = 0; k < j; ++k) d -= W[j + k * q] * W[j + k * q];
        if (!(d > 0)) return j + 1;
        d = std::sqrt(d);
        W[j + j * q] = d;
        ld += std::log(d);
        for (int i = j + 1; i < q; ++i) {
            double s = W[i + j * q];
            for (int k = 0; k < j; ++k) s -= W[i + k * q] * W[j + k * q];
            W[i + j * q] = s / d;
        }
    }
    for (int c = 0; c < nb; ++c) {
        double *b = &B[(size_t)c * q];
        for (int i = 0; i < q; ++i) {
            double s = b[i];
            for (int k = 0; k < i; ++k) s -= W[i + k * q] * b[k];
            b[i] = s / W[i + i * q];
        }
        for (int i = q - 1; i >= 0; --i) {
            double s = b[i];
            for (int k = i + 1; k < q; ++k) s -= W[k + i * q] * b[k];
            b[i] = s / W[i + i * q];
        }
    }
    *logdet_half = ld;
    return 0;
}

// shared tail of Profile / REML: Gram matrix G of [y_1..y_r, Y] (Y = L^-1 Xb) ->
// quad_k = G_kk - g_k' W^-1 g_k with W = Y'Y, g_k = Y'y_k.
static int profile_tail(cocons_fit *f, int nxb, double n_eff, bool reml, double *sum_logliks, double *parts)
{
    const int r = f->r, nr = r + nxb;
    const double *G = f->hout + 1;
    std::vector<double> W((size_t)nxb * nxb), Bv((size_t)nxb * r);
    for (int a = 0; a < nxb; ++a)
        for (int b = 0; b < nxb; ++b) W[a + (size_t)b * nxb] = G[(r + a) * nr + (r + b)];
    for (int k = 0; k < r; ++k)
        for (int a = 0; a < nxb; ++a) Bv[(size_t)k * nxb + a] = G[k * nr + (r + a)];
    std::vector<double> g = Bv;
    double ldW = 0;
    if (host_spd_solve(nxb, W, r, Bv, &ldW)) return fail(-4, "X' Sigma^-1 X is not positive definite");
    double logdet = f->hout[0], total = 0.0;
    for (int k = 0; k < r; ++k) {
        double corr = 0;
        for (int a = 0; a < nxb; ++a) corr += g[(size_t)k * nxb + a] * Bv[(size_t)k * nxb + a];
        double quad = G[k * nr + k] - corr;
        total += n_eff * LOG_2PI + 2 * logdet + (reml ? 2 * ldW : 0.0) + quad;
        if (parts) parts[2 + k] = quad;
    }
    if (parts) {
        parts[0] = logdet; parts[1] = ldW;
        // generalised-least-squares coefficients W^-1 Xb' Sigma^-1 zbar, zbar = rowSums(z)/r -- the
        // "Compute Betas" block of cocoOptim's pml/reml branch (R/optim.R:329-341)
        for (int a = 0; a < nxb; ++a) {
            double s = 0;
            for (int k = 0; k < r; ++k) s += Bv[(size_t)k * nxb + a];
            parts[2 + r + a] = s / r;
        }
    }
    *sum_logliks = total;
    return 0;
}

extern "C" int cocons_neg2loglik_profile(cocons_fit *f, const double *theta, double *sum_logliks, double *parts)
{
    FIT_ENTER(f);
    if (int rc = no_taper(f, "cocons_neg2loglik_profile")) return rc;
    if (!theta || !sum_logliks) return fail(-1, "cocons_neg2loglik_profile: null argument");
    if (f->r < 1 || f->q < 1) return fail(-1, "cocons_neg2loglik_profile: fit needs z and x_betas");
    for (;;) {
        if (int rc = enqueue_eval(f, theta, nullptr, false, f->dxb, f->q, nullptr, false)) return rc;
        HIPCHK(hipStreamSynchronize(f->stream));
        int st = info_status(f);
        if (engine_retry(f, st)) continue;
        if (st) return st;
        break;
    }
    return profile_tail(f, f->q, (double)f->n_user, false, sum_logliks, parts);   // R/neg2loglikelihood.R:155-160
}

extern "C" int cocons_neg2loglik_reml(cocons_fit *f, const double *theta, int rank, double *sum_logliks, double *parts)
{
    FIT_ENTER(f);
    if (int rc = no_taper(f, "cocons_neg2loglik_reml")) return rc;
    if (!theta || !sum_logliks) return fail(-1, "cocons_neg2loglik_reml: null argument");
    if (f->r < 1) return fail(-1, "cocons_neg2loglik_reml: fit has no z");
    for (;;) {
        if (int rc = enqueue_eval(f, theta, nullptr, false, f->dX, f->p, nullptr, false)) return rc;
        HIPCHK(hipStreamSynchronize(f->stream));
        int st = info_status(f);
        if (engine_retry(f, st)) continue;
        if (st) return st;
        break;
    }
    return profile_tail(f, f->p, (double)(f->n_user - rank), true, sum_logliks, parts);   // :283-287
}

// ---------------------------------------------------------------------------
extern "C" int cocons_fit_profile(cocons_fit *f, const double *theta, const double *mean, int reps, double *ms)
{
    FIT_ENTER(f);
    if (int rc = no_taper(f, "cocons_fit_profile")) return rc;
    if (!theta || !mean || !ms || reps < 1) return fail(-1, "cocons_fit_profile: bad argument");
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    double dag_ms = 0.0;
    for (int it = 0; it < reps; ++it) {
        std::vector<hipEvent_t> ev;
        f->upd_flops = 0.0;
        f->dag_flops = 0.0; f->dag_events = 0;
        if (int rc = enqueue_eval(f, theta, mean, true, nullptr, 0, &ev, true)) return rc;
        HIPCHK(hipStreamSynchronize(f->stream));
        float t01, t12, t23, t03;
        HIPCHK(hipEventElapsedTime(&t01, f->ev[0], f->ev[1]));
        HIPCHK(hipEventElapsedTime(&t12, f->ev[1], f->ev[2]));
        HIPCHK(hipEventElapsedTime(&t23, f->ev[2], f->ev[3]));
        HIPCHK(hipEventElapsedTime(&t03, f->ev[0], f->ev[3]));
        acc[0] += t01; acc[1] += t12; acc[2] += t23; acc[3] += t03;
        double sum = 0;
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            float t;
            HIPCHK(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
            sum += t;
            if (i == 0 && f->dag_events) dag_ms += t;
        }
        acc[6] += sum;
        acc[5] = (double)(ev.size() / 2);
        for (auto e : ev) hipEventDestroy(e);
    }
    for (int i = 0; i < 4; ++i) ms[i] = acc[i] / reps;
    ms[5] = acc[5];
    ms[6] = acc[6] / reps;
    ms[4] = acc[5] > 0 ? ms[6] / acc[5] : 0.0;
    ms[7] = f->upd_flops;
    ms[8] = dag_ms / reps;
    ms[9] = f->dag_flops;
    int st = info_status(f);
    if (engine_retry(f, st)) return cocons_fit_profile(f, theta, mean, reps, ms);
    return st;
}

// (diagnostics) the covariance assembly of an evaluation ALONE, `reps` times back to back on the handle's stream between two
// events: ms_out[0] = mean milliseconds per assembly.  tools/diag/overlap_probe.py runs it on one handle while another thread
// evaluates on a second handle: what the assembly (fp64 vector work) and the factorisation (fp64 matrix work) cost each other
// when they share the chip -- the measurement behind DESIGN.md section 8 "Assembly beside the factorisation".
extern "C" int cocons_debug_assembly_loop(cocons_fit *f, const double *theta, int reps, double *ms_out)
{
    FIT_ENTER(f);
    if (int rc = no_taper(f, "cocons_debug_assembly_loop")) return rc;
    if (!theta || !ms_out || reps < 1) return fail(-1, "cocons_debug_assembly_loop: bad argument");
    if (int rc = fit_alloc_matrix(f, f->r > 0 ? f->r : 1)) return rc;
    hipEventRecord(f->ev[0], f->stream);
    for (int i = 0; i < reps; ++i) assemble_sigma(f, theta, 0, 0, f->npad);
    hipEventRecord(f->ev[1], f->stream);
    HIPCHK(hipStreamSynchronize(f->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, f->ev[0], f->ev[1]));
    ms_out[0] = (double)ms / reps;
    return 0;
}

// ---------------------------------------------------------------------------
// stateless covariance entry points
static int cov_common(int which, int n, int m, int p, const double *theta, const double *locs,
                      const double *locs_pred, const double *X, const double *X_pred,
                      const double *smooth_limits, double *out)
{
    if (n <= 0 || p <= 0 || p > COCONS_P_MAX || !theta || !locs || !X || !out || (which != 1 && !smooth_limits) ||
        (which == 2 && (m <= 0 || !locs_pred || !X_pred)))
        return fail(-1, "cov_rns*: bad argument");
    double sl_dummy[2] = {0.0, 0.0};
    const double *sl = smooth_limits ? smooth_limits : sl_dummy;
    ThetaVecs tv;
    make_theta_vecs(theta, p, tv);
    ModeSel ms = select_mode(theta, p, sl, which);
    hipStream_t s = nullptr;     // own non-blocking stream (the library never launches on the NULL stream)
    double *dX = nullptr, *dl = nullptr, *dloc = nullptr, *dXp = nullptr, *dlp = nullptr, *dlocp = nullptr, *dout = nullptr;
    const size_t rows = which == 2 ? (size_t)m : (size_t)n;
    int rc = 0;
#define CKG(expr)                                                                 \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) {                                                  \
            rc = fail(-100 - (int)e__, "cov_rns*: %s", hipGetErrorString(e__));   \
            goto done;                                                            \
        }                                                                         \
    } while (0)
    CKG(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CKG(hipMalloc(&dX, (size_t)n * p * sizeof(double)));
    CKG(hipMalloc(&dl, (size_t)n * 2 * sizeof(double)));
    CKG(hipMalloc(&dloc, (size_t)LOCP_FIELDS * n * sizeof(double)));
    CKG(hipMalloc(&dout, rows * (size_t)n * sizeof(double)));
    CKG(hipMemcpyAsync(dX, X, (size_t)n * p * sizeof(double), hipMemcpyHostToDevice, s));
    CKG(hipMemcpyAsync(dl, locs, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    {
        LocArgs la;
        la.n = n; la.p = p; la.X = dX; la.ldx = n; la.locs = dl; la.ldl = n;
        la.out = dloc; la.stride = n; la.smooth_kind = ms.smooth_kind;
        la.smooth_min = sl[0]; la.smooth_max = sl[1]; la.th = tv;
        launch_loc_params(la, s);
        PairArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.n = n; pa.cols = dloc; pa.stride = n; pa.out = dout; pa.gr = ms.gr; pa.nu_fixed = ms.nu_fixed;
        if (which == 2) {
            CKG(hipMalloc(&dXp, (size_t)m * p * sizeof(double)));
            CKG(hipMalloc(&dlp, (size_t)m * 2 * sizeof(double)));
            CKG(hipMalloc(&dlocp, (size_t)LOCP_FIELDS * m * sizeof(double)));
            CKG(hipMemcpyAsync(dXp, X_pred, (size_t)m * p * sizeof(double), hipMemcpyHostToDevice, s));
            CKG(hipMemcpyAsync(dlp, locs_pred, (size_t)m * 2 * sizeof(double), hipMemcpyHostToDevice, s));
            LocArgs lp = la;
            lp.n = m; lp.X = dXp; lp.ldx = m; lp.locs = dlp; lp.ldl = m; lp.out = dlocp; lp.stride = m;
            launch_loc_params(lp, s);
            pa.m = m; pa.rows = dlocp; pa.stride_rows = m; pa.ld = m; pa.nrows_out = m; pa.ncols_out = n;
            launch_pair_rect(ms.mode, pa, s);
        } else {
            pa.m = n; pa.rows = dloc; pa.stride_rows = n; pa.ld = n; pa.nrows_out = n; pa.ncols_out = n;
            launch_pair_sym(ms.mode, true, pa, s);
        }
    }
    CKG(hipGetLastError());
    CKG(hipMemcpyAsync(out, dout, rows * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    CKG(hipStreamSynchronize(s));
done:
    if (s) { hipStreamSynchronize(s); hipStreamDestroy(s); }
    hipFree(dX); hipFree(dl); hipFree(dloc); hipFree(dXp); hipFree(dlp); hipFree(dlocp); hipFree(dout);
#undef CKG
    return rc;
}

extern "C" int cocons_cov_rns(int n, int p, const double *theta, const double *locs, const double *X,
                              const double *smooth_limits, double *out)
{
    return cov_common(0, n, 0, p, theta, locs, nullptr, X, nullptr, smooth_limits, out);
}

extern "C" int cocons_cov_rns_classic(int n, int p, const double *theta, const double *locs, const double *X, double *out)
{
    return cov_common(1, n, 0, p, theta, locs, nullptr, X, nullptr, nullptr, out);
}

extern "C" int cocons_cov_rns_pred(int n, int m, int p, const double *theta, const double *locs,
                                   const double *locs_pred, const double *X, const double *X_pred,
                                   const double *smooth_limits, double *out)
{
    return cov_common(2, n, m, p, theta, locs, locs_pred, X, X_pred, smooth_limits, out);
}

// ---------------------------------------------------------------------------
// sparse/taper covariance entries (first slice): the assembly only -- the spam
// Cholesky behind R/neg2loglikelihood.R:20-108 stays with the caller.
static int taper_common(bool pred, int n, int m, int p, const double *theta, const double *locs,
                        const double *locs_pred, const double *X, const double *X_pred, const double *smooth_limits,
                        int nnz, const int *colindices, const int *rowpointers, double *out)
{
    if (n <= 0 || p <= 0 || p > COCONS_P_MAX || !theta || !locs || !X || !smooth_limits || nnz < 0 || !colindices ||
        !rowpointers || (nnz > 0 && !out) || (pred && (m <= 0 || !locs_pred || !X_pred)))
        return fail(-1, "cov_rns_taper*: bad argument");
    const int nrows = pred ? m : n;
    if (rowpointers[0] != 1 || rowpointers[nrows] != nnz + 1) return fail(-1, "cov_rns_taper*: rowpointers do not match nnz (1-based CSR expected)");
    for (int w = 0; w < nnz; ++w)
        if (colindices[w] < 1 || colindices[w] > n) return fail(-1, "cov_rns_taper*: column index out of range");
    ThetaVecs tv;
    make_theta_vecs(theta, p, tv);
    for (int i = 0; i < p; ++i) tv.two_scale_je[i] = 2 * theta[TH_SCALE * p + i];   // FULL scale vector (cocons_taper.cpp:207)
    // smoothness dispatch of cov_rns_taper (:183-201); the prediction variant always takes the Bessel branch
    ModeSel ms = select_mode(theta, p, smooth_limits, pred ? 2 : 0);
    hipStream_t s = nullptr;
    double *dX = nullptr, *dl = nullptr, *dloc = nullptr, *dXp = nullptr, *dlp = nullptr, *dlocp = nullptr, *dout = nullptr;
    int *dci = nullptr, *drp = nullptr;
    int rc = 0;
#define CKT(expr)                                                                 \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) {                                                  \
            rc = fail(-100 - (int)e__, "cov_rns_taper*: %s", hipGetErrorString(e__)); \
            goto done;                                                            \
        }                                                                         \
    } while (0)
    CKT(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CKT(hipMalloc(&dX, (size_t)n * p * sizeof(double)));
    CKT(hipMalloc(&dl, (size_t)n * 2 * sizeof(double)));
    CKT(hipMalloc(&dloc, (size_t)LOCP_FIELDS * n * sizeof(double)));
    CKT(hipMalloc(&dout, (size_t)(nnz > 0 ? nnz : 1) * sizeof(double)));
    CKT(hipMalloc(&dci, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int)));
    CKT(hipMalloc(&drp, (size_t)(nrows + 1) * sizeof(int)));
    CKT(hipMemcpyAsync(dX, X, (size_t)n * p * sizeof(double), hipMemcpyHostToDevice, s));
    CKT(hipMemcpyAsync(dl, locs, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    CKT(hipMemcpyAsync(dci, colindices, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, s));
    CKT(hipMemcpyAsync(drp, rowpointers, (size_t)(nrows + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    {
        LocArgs la;
        la.n = n; la.p = p; la.X = dX; la.ldx = n; la.locs = dl; la.ldl = n;
        la.out = dloc; la.stride = n; la.smooth_kind = ms.smooth_kind;
        la.smooth_min = smooth_limits[0]; la.smooth_max = smooth_limits[1]; la.th = tv;
        launch_loc_params(la, s);
        if (pred) {
            CKT(hipMalloc(&dXp, (size_t)m * p * sizeof(double)));
            CKT(hipMalloc(&dlp, (size_t)m * 2 * sizeof(double)));
            CKT(hipMalloc(&dlocp, (size_t)LOCP_FIELDS * m * sizeof(double)));
            CKT(hipMemcpyAsync(dXp, X_pred, (size_t)m * p * sizeof(double), hipMemcpyHostToDevice, s));
            CKT(hipMemcpyAsync(dlp, locs_pred, (size_t)m * 2 * sizeof(double), hipMemcpyHostToDevice, s));
            LocArgs lp = la;
            lp.n = m; lp.X = dXp; lp.ldx = m; lp.locs = dlp; lp.ldl = m; lp.out = dlocp; lp.stride = m;
            launch_loc_params(lp, s);
            launch_taper(MODE_GEOM, true, m, nnz, dci, drp, dlocp, m, dloc, n, 0.0, dout, s);
        } else {
            launch_taper(ms.mode, false, n, nnz, dci, drp, dloc, n, dloc, n, ms.nu_fixed, dout, s);
        }
    }
    CKT(hipGetLastError());
    if (nnz > 0) CKT(hipMemcpyAsync(out, dout, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost, s));
    CKT(hipStreamSynchronize(s));
done:
    if (s) { hipStreamSynchronize(s); hipStreamDestroy(s); }
    hipFree(dX); hipFree(dl); hipFree(dloc); hipFree(dXp); hipFree(dlp); hipFree(dlocp); hipFree(dout);
    hipFree(dci); hipFree(drp);
#undef CKT
    return rc;
}

extern "C" int cocons_cov_rns_taper(int n, int p, const double *theta, const double *locs, const double *X,
                                    const double *smooth_limits, int nnz, const int *colindices,
                                    const int *rowpointers, double *entries)
{
    return taper_common(false, n, 0, p, theta, locs, nullptr, X, nullptr, smooth_limits, nnz, colindices, rowpointers, entries);
}

extern "C" int cocons_cov_rns_taper_pred(int n, int m, int p, const double *theta, const double *locs,
                                         const double *locs_pred, const double *X, const double *X_pred,
                                         const double *smooth_limits, int nnz, const int *colindices,
                                         const int *rowpointers, double *entries)
{
    return taper_common(true, n, m, p, theta, locs, locs_pred, X, X_pred, smooth_limits, nnz, colindices, rowpointers, entries);
}

// ---------------------------------------------------------------------------
// Rows of the dense covariance / correlation matrix of a fit, without the n x n matrix.
// Works on the fit's ORIGINAL observation order (the reference's orientation rule "ii = the smaller index"
// and its u <= eps rule depend on it), from the host copies the handle keeps: O(n p) bytes go down,
// nidx * n doubles come back.
extern "C" int cocons_cov_rows(cocons_fit *f, const double *theta, int classic, int nidx, const int *idx, int cor,
                               double *out)
{
    FIT_ENTER(f);
    if (int rc = no_taper(f, "cocons_cov_rows")) return rc;
    if (!theta || nidx <= 0 || !idx || !out) return fail(-1, "cocons_cov_rows: bad argument");
    const int n = f->n_user, p = f->p;          // (works on the host copies: the caller's observations in the caller's order)
    for (int b = 0; b < nidx; ++b)
        if (idx[b] < 0 || idx[b] >= n) return fail(-1, "cocons_cov_rows: row index out of range (0-based)");
    ThetaVecs tv;
    make_theta_vecs(theta, p, tv);
    ModeSel ms = select_mode(theta, p, f->smooth_limits, classic ? 1 : 0);
    hipStream_t s = f->stream;
    double *dX = nullptr, *dl = nullptr, *dloc = nullptr, *dout = nullptr;
    int *didx = nullptr;
    int rc = 0;
    do {
        hipError_t e;
#define CKR(expr) if ((e = (expr)) != hipSuccess) { rc = fail(-100 - (int)e, "cocons_cov_rows: %s", hipGetErrorString(e)); break; }
        CKR(hipMalloc(&dX, (size_t)n * p * sizeof(double)));
        CKR(hipMalloc(&dl, (size_t)n * 2 * sizeof(double)));
        CKR(hipMalloc(&dloc, (size_t)LOCP_FIELDS * n * sizeof(double)));
        CKR(hipMalloc(&dout, (size_t)nidx * n * sizeof(double)));
        CKR(hipMalloc(&didx, (size_t)nidx * sizeof(int)));
        CKR(hipMemcpyAsync(dX, f->h_X->data(), (size_t)n * p * sizeof(double), hipMemcpyHostToDevice, s));
        CKR(hipMemcpyAsync(dl, f->h_locs->data(), (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, s));
        CKR(hipMemcpyAsync(didx, idx, (size_t)nidx * sizeof(int), hipMemcpyHostToDevice, s));
        LocArgs la;
        la.n = n; la.p = p; la.X = dX; la.ldx = n; la.locs = dl; la.ldl = n;
        la.out = dloc; la.stride = n; la.smooth_kind = ms.smooth_kind;
        la.smooth_min = f->smooth_limits[0]; la.smooth_max = f->smooth_limits[1]; la.th = tv;
        launch_loc_params(la, s);
        launch_cov_rows(ms.mode, n, nidx, didx, dloc, n, ms.gr, ms.nu_fixed, cor, dout, s);
        CKR(hipGetLastError());
        CKR(hipMemcpyAsync(out, dout, (size_t)nidx * n * sizeof(double), hipMemcpyDeviceToHost, s));
        CKR(hipStreamSynchronize(s));
#undef CKR
    } while (0);
    hipStreamSynchronize(s);
    hipFree(dX); hipFree(dl); hipFree(dloc); hipFree(dout); hipFree(didx);
    return rc;
}

// ---------------------------------------------------------------------------
// kriging core: rows under the matrix = [ (z - X mean)' ; cov_rns_pred (m x n) ]
extern "C" int cocons_predict_dense(cocons_fit *f, const double *theta, const double *mean, int z_col,
                                    int m, const double *locs_pred, const double *X_pred,
                                    double *stochastic, double *quadform)
{
    FIT_ENTER(f);
    if (int rc = no_taper(f, "cocons_predict_dense")) return rc;
    if (!theta || !mean || m <= 0 || !locs_pred || !X_pred || !stochastic || !quadform || z_col < 0 || z_col >= f->r)
        return fail(-1, "cocons_predict_dense: bad argument");
    const int p = f->p, n = f->n;
    if (m > f->pred_cap) {
        hipFree(f->dlocp); hipFree(f->dXp); hipFree(f->dlocsp); hipFree(f->dstoch); hipFree(f->dquad); hipFree(f->dred);
        f->dlocp = f->dXp = f->dlocsp = f->dstoch = f->dquad = f->dred = nullptr;
        f->pred_cap = 0;
        HIPCHK(hipMalloc(&f->dlocp, (size_t)LOCP_FIELDS * m * sizeof(double)));
        HIPCHK(hipMalloc(&f->dXp, (size_t)m * p * sizeof(double)));
        HIPCHK(hipMalloc(&f->dlocsp, (size_t)m * 2 * sizeof(double)));
        HIPCHK(hipMalloc(&f->dstoch, (size_t)m * sizeof(double)));
        HIPCHK(hipMalloc(&f->dquad, (size_t)m * sizeof(double)));
        HIPCHK(hipMalloc(&f->dred, row_reduce_scratch_doubles(n, m) * sizeof(double)));
        f->pred_cap = m;
    }
    if (int rc = fit_alloc_matrix(f, m + 1)) return rc;
    hipStream_t s = f->stream;
    HIPCHK(hipMemcpyAsync(f->dXp, X_pred, (size_t)m * p * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(f->dlocsp, locs_pred, (size_t)m * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    if (int rc = reset_info(f)) return rc;
    assemble_sigma(f, theta, 0, 0, f->npad);
    // row npad: residual of realization z_col; rows npad+1 .. npad+m: cross-covariance
    {
        RhsArgs ra;
        memset(&ra, 0, sizeof ra);
        ra.n = n; ra.p = p; ra.X = f->dX; ra.ldx = n; ra.use_trend = 1;
        for (int i = 0; i < p; ++i) ra.mean[i] = mean[i];
        ra.src = f->dz + (size_t)z_col * n; ra.lds = n;
        ra.out = f->dA; ra.ld = f->lda; ra.row0 = f->npad; ra.nrows = 1;
        ra.nrows_zero = f->rhs_act - 1;      // also clears padding rows and columns >= n
        ra.col0 = 0; ra.ncols_out = f->npad;
        launch_rhs_rows(ra, s);
        ThetaVecs tv;
        make_theta_vecs(theta, p, tv);
        ModeSel ms = select_mode(theta, p, f->smooth_limits, 2);
        LocArgs lp;
        lp.n = m; lp.p = p; lp.X = f->dXp; lp.ldx = m; lp.locs = f->dlocsp; lp.ldl = m;
        lp.out = f->dlocp; lp.stride = m; lp.smooth_kind = ms.smooth_kind;
        lp.smooth_min = f->smooth_limits[0]; lp.smooth_max = f->smooth_limits[1]; lp.th = tv;
        launch_loc_params(lp, s);
        // the observation-side SoA must use the pred-branch smoothness (always logistic+sqrt, :381)
        LocArgs lo = lp;
        lo.n = n; lo.X = f->dX; lo.ldx = n; lo.locs = f->dlocs; lo.ldl = n; lo.out = f->dloc; lo.stride = f->npad;
        PairArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.n = n; pa.m = m; pa.rows = f->dlocp; pa.stride_rows = m; pa.cols = f->dloc; pa.stride = f->npad;
        pa.out = f->dA + f->npad + 1; pa.ld = f->lda; pa.nrows_out = m; pa.ncols_out = n;
        pa.gr = ms.gr; pa.nu_fixed = 0.0;
        // Sigma was assembled from dloc above (stream order); rebuild dloc only if cov_rns used a
        // different smoothness vector (fixed-nu branch) than cov_rns_pred does.
        ModeSel ms0 = select_mode(theta, p, f->smooth_limits, 0);
        if (ms0.smooth_kind != ms.smooth_kind) launch_loc_params(lo, s);
        launch_pair_rect(MODE_GEOM, pa, s);
    }
    // (the dependency-driven schedule may take the head of this factorisation too -- round 6: the row reductions below read the
    // factor from both buffers like the objectives' do; with m rows under the matrix every step is a long one)
    FactorView pv = main_view(f);
    pv.dag_ok = true;
    if (int rc = factorize(f, pv, nullptr)) return rc;
    launch_row_reduce(f->dA, f->lda, n, f->npad, f->npad + 1, m, f->dstoch, f->dquad, f->dred, s, 0, 0,
                      f->dag_used ? f->dP : nullptr, f->dag_used ? 2 * TILE * f->dag_nsteps : 0);
    HIPCHK(hipMemcpyAsync(stochastic, f->dstoch, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(quadform, f->dquad, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(f->hinfo, f->dinfo, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    int st = info_status(f);
    if (engine_retry(f, st)) return cocons_predict_dense(f, theta, mean, z_col, m, locs_pred, X_pred, stochastic, quadform);
    return st;
}

// Kriging core of the sparse branch of cocoPredict (R/predict.R:216-283) on a taper handle: S = taper o
// cov_rns_taper(theta) as in the objective, C = pred_taper o cov_rns_taper_pred(theta) (m x n, its own pattern);
// one bordered DENSE factorisation replaces  inv_cov <- spam::solve(S, t(C))  ("memory intensive", :244) and gives
//   stochastic[i] = C[i,] S^-1 resid    (:252)      quadform[i] = C[i,] S^-1 C[i,]'    (:267)
extern "C" int cocons_predict_taper(cocons_fit *f, const double *theta, const double *mean, int z_col, int m,
                                    const double *locs_pred, const double *X_pred, int nnz_pred,
                                    const int *colindices_pred, const int *rowpointers_pred,
                                    const double *taper_entries_pred, double *stochastic, double *quadform)
{
    FIT_ENTER(f);
    if (f->taper_nnz <= 0) return fail(-1, "cocons_predict_taper: not a taper fit");
    if (!theta || !mean || m <= 0 || !locs_pred || !X_pred || !stochastic || !quadform || z_col < 0 || z_col >= f->r ||
        nnz_pred < 0 || !rowpointers_pred || (nnz_pred > 0 && (!colindices_pred || !taper_entries_pred)))
        return fail(-1, "cocons_predict_taper: bad argument");
    const int p = f->p, n = f->n;
    if (rowpointers_pred[0] != 1 || rowpointers_pred[m] != nnz_pred + 1)
        return fail(-1, "cocons_predict_taper: rowpointers do not match nnz (1-based CSR expected)");
    for (int w = 0; w < nnz_pred; ++w)
        if (colindices_pred[w] < 1 || colindices_pred[w] > n) return fail(-1, "cocons_predict_taper: column index out of range");
    if (m > f->pred_cap) {
        hipFree(f->dlocp); hipFree(f->dXp); hipFree(f->dlocsp); hipFree(f->dstoch); hipFree(f->dquad); hipFree(f->dred);
        f->dlocp = f->dXp = f->dlocsp = f->dstoch = f->dquad = f->dred = nullptr;
        f->pred_cap = 0;
        HIPCHK(hipMalloc(&f->dlocp, (size_t)LOCP_FIELDS * m * sizeof(double)));
        HIPCHK(hipMalloc(&f->dXp, (size_t)m * p * sizeof(double)));
        HIPCHK(hipMalloc(&f->dlocsp, (size_t)m * 2 * sizeof(double)));
        HIPCHK(hipMalloc(&f->dstoch, (size_t)m * sizeof(double)));
        HIPCHK(hipMalloc(&f->dquad, (size_t)m * sizeof(double)));
        HIPCHK(hipMalloc(&f->dred, row_reduce_scratch_doubles(n, m) * sizeof(double)));
        f->pred_cap = m;
    }
    if (int rc = fit_alloc_matrix(f, m + 1)) return rc;
    hipStream_t s = f->stream;
    int *dci = nullptr, *drp = nullptr;
    double *dtv = nullptr;
    const size_t nz = nnz_pred > 0 ? (size_t)nnz_pred : 1;
    int st = 0;
    do {
#define CKP(expr) { hipError_t e__ = (expr); if (e__ != hipSuccess) { st = fail(-100 - (int)e__, "cocons_predict_taper: %s", hipGetErrorString(e__)); break; } }
        CKP(hipMalloc(&dci, nz * sizeof(int)));
        CKP(hipMalloc(&drp, (size_t)(m + 1) * sizeof(int)));
        CKP(hipMalloc(&dtv, nz * sizeof(double)));
        CKP(hipMemcpyAsync(f->dXp, X_pred, (size_t)m * p * sizeof(double), hipMemcpyHostToDevice, s));
        CKP(hipMemcpyAsync(f->dlocsp, locs_pred, (size_t)m * 2 * sizeof(double), hipMemcpyHostToDevice, s));
        CKP(hipMemcpyAsync(drp, rowpointers_pred, (size_t)(m + 1) * sizeof(int), hipMemcpyHostToDevice, s));
        std::vector<int> mapped;                    // the pattern's columns in the handle's order of the observations
        if (nnz_pred > 0) {
            mapped.resize(nnz_pred);
            for (int w = 0; w < nnz_pred; ++w) mapped[w] = (*f->taper_inv)[colindices_pred[w] - 1] + 1;
            CKP(hipMemcpy(dci, mapped.data(), (size_t)nnz_pred * sizeof(int), hipMemcpyHostToDevice));
            CKP(hipMemcpyAsync(dtv, taper_entries_pred, (size_t)nnz_pred * sizeof(double), hipMemcpyHostToDevice, s));
        }
#undef CKP
        for (;;) {
            if ((st = reset_info(f))) break;
            if ((st = assemble_sigma_taper(f, theta))) break;          // zeroes the whole buffer, border rows included
            RhsArgs ra;
            memset(&ra, 0, sizeof ra);
            ra.n = n; ra.p = p; ra.X = f->dX; ra.ldx = n; ra.use_trend = 1;
            for (int i = 0; i < p; ++i) ra.mean[i] = mean[i];
            ra.src = f->dz + (size_t)z_col * n; ra.lds = n;
            ra.out = f->dA; ra.ld = f->lda; ra.row0 = f->npad; ra.nrows = 1;
            ra.skew = f->skew; ra.npad = f->npad;
            ra.nrows_zero = f->rhs_act - 1;
            ra.col0 = 0; ra.ncols_out = f->npad;
            launch_rhs_rows(ra, s);
            // parameters as cocons_cov_rns_taper_pred prepares them: FULL scale vector, prediction-branch smoothness
            ThetaVecs tv;
            make_theta_vecs(theta, p, tv);
            for (int i = 0; i < p; ++i) tv.two_scale_je[i] = 2 * theta[TH_SCALE * p + i];
            ModeSel ms = select_mode(theta, p, f->smooth_limits, 2);
            LocArgs lp;
            lp.n = m; lp.p = p; lp.X = f->dXp; lp.ldx = m; lp.locs = f->dlocsp; lp.ldl = m;
            lp.out = f->dlocp; lp.stride = m; lp.smooth_kind = ms.smooth_kind;
            lp.smooth_min = f->smooth_limits[0]; lp.smooth_max = f->smooth_limits[1]; lp.th = tv;
            launch_loc_params(lp, s);
            LocArgs lo = lp;
            lo.n = n; lo.X = f->dX; lo.ldx = n; lo.locs = f->dlocs; lo.ldl = n; lo.out = f->dloc; lo.stride = f->npad;
            launch_loc_params(lo, s);                                   // (after the entries of S were computed: stream order)
            launch_taper(MODE_GEOM, true, m, nnz_pred, dci, drp, f->dlocp, m, f->dloc, f->npad, 0.0, nullptr, s,
                         dtv, f->dA, f->lda, f->npad + 1, f->skew, f->npad);
            factorize(f, main_view(f), nullptr);
            launch_row_reduce(f->dA, f->lda, n, f->npad, f->npad + 1, m, f->dstoch, f->dquad, f->dred, s, f->skew, f->npad);
            hipMemcpyAsync(stochastic, f->dstoch, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s);
            hipMemcpyAsync(quadform, f->dquad, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s);
            hipMemcpyAsync(f->hinfo, f->dinfo, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { st = fail(-100 - (int)e, "cocons_predict_taper: %s", hipGetErrorString(e)); break; }
            st = info_status(f);
            if (engine_retry(f, st)) continue;
            break;
        }
    } while (0);
    hipStreamSynchronize(s);
    hipFree(dci); hipFree(drp); hipFree(dtv);
    return st;
}

// ---------------------------------------------------------------------------
// marginal simulation core: replaces R/sim.R:147-172
//   covmat <- cov_rns[_classic](...); cholS <- chol(covmat); t(sweep(t(iiderrors) %*% cholS, 2, X %*% mean, "+"))
extern "C" int cocons_sim_dense(cocons_fit *f, const double *theta, const double *mean, int classic,
                                int nsim, const double *iiderrors, double *out)
{
    FIT_ENTER(f);
    if (int rc = no_taper(f, "cocons_sim_dense")) return rc;
    if (!theta || !mean || nsim <= 0 || !iiderrors || !out) return fail(-1, "cocons_sim_dense: bad argument");
    if (f->sorted) {
        // L E depends on the ORDER of the observations (the factor of a permuted matrix is not the
        // permuted factor): the field for given draws is only reproduced in the caller's order
        if (!f->unsorted) {
            f->unsorted = fit_create_impl(f->n_user, f->p, f->r, 0, f->h_locs->data(), f->h_X->data(),
                                          f->r > 0 ? f->h_z->data() : nullptr, nullptr, f->smooth_limits,
                                          f->device, false);
            if (!f->unsorted) return -1;
        }
        return cocons_sim_dense(f->unsorted, theta, mean, classic, nsim, iiderrors, out);
    }
    const int n = f->n, p = f->p;
    double *dE = nullptr, *dY = nullptr, *dtr = nullptr;
    int rc = 0;
    hipStream_t s = f->stream;
    do {
        if ((rc = fit_alloc_matrix(f, 1))) break;
        hipError_t e;
#define CKS(expr) if ((e = (expr)) != hipSuccess) { rc = fail(-100 - (int)e, "cocons_sim_dense: %s", hipGetErrorString(e)); break; }
        CKS(hipMalloc(&dE, (size_t)n * nsim * sizeof(double)));
        CKS(hipMalloc(&dY, (size_t)n * nsim * sizeof(double)));
        CKS(hipMalloc(&dtr, (size_t)n * sizeof(double)));
        CKS(hipMemcpyAsync(dE, iiderrors, (size_t)n * nsim * sizeof(double), hipMemcpyHostToDevice, s));
        {   // trend = X %*% mean on the host (O(n p)), as the reference does (:170)
            std::vector<double> tr(n, 0.0);
            for (int j = 0; j < p; ++j)
                for (int i = 0; i < n; ++i) tr[i] += (*f->h_X)[(size_t)i + (size_t)j * n] * mean[j];
            CKS(hipMemcpyAsync(dtr, tr.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
            CKS(hipStreamSynchronize(s));
        }
        if ((rc = reset_info(f))) break;
        f->nrhs_cur = 0;
        assemble_sigma(f, theta, classic ? 1 : 0, 0, f->npad);
        {   // no right-hand sides: clear the rows under the matrix
            RhsArgs ra;
            memset(&ra, 0, sizeof ra);
            ra.n = n; ra.p = p; ra.X = f->dX; ra.ldx = n; ra.src = f->dX; ra.lds = n;
            ra.out = f->dA; ra.ld = f->lda; ra.row0 = f->npad; ra.nrows = 0; ra.nrows_zero = f->rhs_act;
            ra.col0 = 0; ra.ncols_out = f->npad;
            launch_rhs_rows(ra, s);
        }
        factorize(f, main_view(f), nullptr);
        launch_trmm_lower(f->dA, f->lda, n, dE, n, nsim, dtr, dY, n, s);
        CKS(hipMemcpyAsync(out, dY, (size_t)n * nsim * sizeof(double), hipMemcpyDeviceToHost, s));
        CKS(hipMemcpyAsync(f->hinfo, f->dinfo, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        CKS(hipGetLastError());
        CKS(hipStreamSynchronize(s));
#undef CKS
        rc = info_status(f);
    } while (0);
    hipFree(dE); hipFree(dY); hipFree(dtr);
    if (engine_retry(f, rc)) return cocons_sim_dense(f, theta, mean, classic, nsim, iiderrors, out);
    return rc;
}

// ---------------------------------------------------------------------------
// conditional simulation core: replaces R/sim.R:84-127
//   covmat, covmat_pred, covmat_unobs; L <- chol(covmat_unobs - covmat_pred solve(covmat) t(covmat_pred));
//   t(sweep(t(iiderrors) %*% L, 2, systematic + stochastic, "+"))
// One Cholesky of the JOINT covariance of (observed, new) locations: its lower-right block is
// the factor of the Schur complement, and the kriging mean falls out of the border row, so the
// LU solve, the m x n x m product and the second chol of the reference are all this one pass.
extern "C" int cocons_sim_cond_dense(cocons_fit *f, const double *theta, const double *mean, int z_col,
                                     int m, const double *locs_pred, const double *X_pred,
                                     const double *locs_unobs, int nsim, const double *iiderrors, double *out)
{
    FIT_ENTER(f);
    if (int rc = no_taper(f, "cocons_sim_cond_dense")) return rc;
    if (!theta || !mean || m <= 0 || !locs_pred || !X_pred || !locs_unobs || nsim <= 0 || !iiderrors || !out ||
        z_col < 0 || z_col >= f->r)
        return fail(-1, "cocons_sim_cond_dense: bad argument");
    const int n = f->n, p = f->p, npad = f->npad;
    const int mpad = round_up(m, TILE), N = npad + mpad;
    const size_t ldj = (size_t)N + TILE;
    double *dJ = nullptr, *dXp = nullptr, *dlp = nullptr, *dlu = nullptr, *dlocp = nullptr, *dlocu = nullptr;
    double *dE = nullptr, *dY = nullptr, *dmu = nullptr, *dst = nullptr, *dq = nullptr, *dred = nullptr;
    hipStream_t s = f->stream;
    int rc = 0;
    do {
        hipError_t e;
#define CKS(expr) if ((e = (expr)) != hipSuccess) { rc = fail(-100 - (int)e, "cocons_sim_cond_dense: %s", hipGetErrorString(e)); break; }
        CKS(hipMalloc(&dJ, ldj * (size_t)N * sizeof(double)));
        CKS(hipMalloc(&dXp, (size_t)m * p * sizeof(double)));
        CKS(hipMalloc(&dlp, (size_t)m * 2 * sizeof(double)));
        CKS(hipMalloc(&dlu, (size_t)m * 2 * sizeof(double)));
        CKS(hipMalloc(&dlocp, (size_t)LOCP_FIELDS * mpad * sizeof(double)));
        CKS(hipMalloc(&dlocu, (size_t)LOCP_FIELDS * mpad * sizeof(double)));
        CKS(hipMalloc(&dE, (size_t)m * nsim * sizeof(double)));
        CKS(hipMalloc(&dY, (size_t)m * nsim * sizeof(double)));
        CKS(hipMalloc(&dmu, (size_t)m * sizeof(double)));
        CKS(hipMalloc(&dst, (size_t)m * sizeof(double)));
        CKS(hipMalloc(&dq, (size_t)m * sizeof(double)));
        CKS(hipMalloc(&dred, row_reduce_scratch_doubles(n, m) * sizeof(double)));
        CKS(hipMemcpyAsync(dXp, X_pred, (size_t)m * p * sizeof(double), hipMemcpyHostToDevice, s));
        CKS(hipMemcpyAsync(dlp, locs_pred, (size_t)m * 2 * sizeof(double), hipMemcpyHostToDevice, s));
        CKS(hipMemcpyAsync(dlu, locs_unobs, (size_t)m * 2 * sizeof(double), hipMemcpyHostToDevice, s));
        CKS(hipMemcpyAsync(dE, iiderrors, (size_t)m * nsim * sizeof(double), hipMemcpyHostToDevice, s));
        if ((rc = reset_info(f))) break;
        f->nrhs_cur = 1;
        ThetaVecs tv;
        make_theta_vecs(theta, p, tv);
        const ModeSel ms0 = select_mode(theta, p, f->smooth_limits, 0);   // cov_rns semantics
        const ModeSel msp = select_mode(theta, p, f->smooth_limits, 2);   // cov_rns_pred semantics
        LocArgs lo;   // observed side
        lo.n = n; lo.p = p; lo.X = f->dX; lo.ldx = n; lo.locs = f->dlocs; lo.ldl = n;
        lo.out = f->dloc; lo.stride = npad; lo.smooth_kind = ms0.smooth_kind;
        lo.smooth_min = f->smooth_limits[0]; lo.smooth_max = f->smooth_limits[1]; lo.th = tv;
        LocArgs lu = lo;   // new locations with the coordinates handed to cov_rns (covmat_unobs)
        lu.n = m; lu.X = dXp; lu.ldx = m; lu.locs = dlu; lu.ldl = m; lu.out = dlocu; lu.stride = mpad;
        LocArgs lp = lu;   // new locations with newlocs (cov_rns_pred), always logistic + sqrt
        lp.locs = dlp; lp.out = dlocp; lp.smooth_kind = msp.smooth_kind;
        launch_loc_params(lo, s);
        launch_loc_params(lu, s);
        launch_loc_params(lp, s);
        PairArgs pa;
        // Sigma_oo  (rows/cols [0, npad))
        memset(&pa, 0, sizeof pa);
        pa.n = n; pa.m = n; pa.rows = f->dloc; pa.cols = f->dloc; pa.stride = npad; pa.stride_rows = npad;
        pa.out = dJ; pa.ld = ldj; pa.nrows_out = npad; pa.ncols_out = npad; pa.gr = ms0.gr; pa.nu_fixed = ms0.nu_fixed;
        launch_pair_sym(ms0.mode, false, pa, s);
        // Sigma_uu  (rows/cols [npad, N))
        memset(&pa, 0, sizeof pa);
        pa.n = m; pa.m = m; pa.rows = dlocu; pa.cols = dlocu; pa.stride = mpad; pa.stride_rows = mpad;
        pa.out = dJ + (size_t)npad + (size_t)npad * ldj; pa.ld = ldj; pa.nrows_out = mpad; pa.ncols_out = mpad;
        pa.gr = ms0.gr; pa.nu_fixed = ms0.nu_fixed;
        launch_pair_sym(ms0.mode, false, pa, s);
        // cross block (rows [npad, N) x cols [0, npad)): observed side needs the pred-branch smoothness
        if (ms0.smooth_kind != msp.smooth_kind) { lo.smooth_kind = msp.smooth_kind; launch_loc_params(lo, s); }
        memset(&pa, 0, sizeof pa);
        pa.n = n; pa.m = m; pa.rows = dlocp; pa.stride_rows = mpad; pa.cols = f->dloc; pa.stride = npad;
        pa.out = dJ + npad; pa.ld = ldj; pa.nrows_out = mpad; pa.ncols_out = npad; pa.gr = msp.gr;
        launch_pair_rect(MODE_GEOM, pa, s);
        {   // border row N: residual of realization z_col over the observed columns, zero elsewhere
            RhsArgs ra;
            memset(&ra, 0, sizeof ra);
            ra.n = n; ra.p = p; ra.X = f->dX; ra.ldx = n; ra.use_trend = 1;
            for (int i = 0; i < p; ++i) ra.mean[i] = mean[i];
            ra.src = f->dz + (size_t)z_col * n; ra.lds = n;
            ra.out = dJ; ra.ld = ldj; ra.row0 = N; ra.nrows = 1; ra.nrows_zero = TILE - 1;
            ra.col0 = 0; ra.ncols_out = N;
            launch_rhs_rows(ra, s);
        }
        FactorView v;
        v.A = dJ; v.lda = ldj; v.nt = N / TILE; v.mt = N / TILE + 1;
        factorize(f, v, nullptr);
        // kriging mean: stochastic_i = sum_{c<n} J(npad+i, c) J(N, c);  tmp_mu = X_pred mean + stochastic
        launch_row_reduce(dJ, ldj, n, N, npad, m, dst, dq, dred, s);
        std::vector<double> mu(m), stv(m);
        CKS(hipMemcpyAsync(stv.data(), dst, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s));
        CKS(hipStreamSynchronize(s));
        for (int i = 0; i < m; ++i) {
            double sys = 0;
            for (int j = 0; j < p; ++j) sys += X_pred[(size_t)i + (size_t)j * m] * mean[j];
            mu[i] = sys + stv[i];
        }
        CKS(hipMemcpyAsync(dmu, mu.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice, s));
        // fields = L_S E + tmp_mu with L_S = lower-right block of the joint factor
        launch_trmm_lower(dJ + (size_t)npad + (size_t)npad * ldj, ldj, m, dE, m, nsim, dmu, dY, m, s);
        CKS(hipMemcpyAsync(out, dY, (size_t)m * nsim * sizeof(double), hipMemcpyDeviceToHost, s));
        CKS(hipMemcpyAsync(f->hinfo, f->dinfo, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        CKS(hipGetLastError());
        CKS(hipStreamSynchronize(s));
#undef CKS
        rc = info_status(f);
        if (rc > n) rc = n;    // a failure inside the Schur block is still "Cholesky error"
    } while (0);
    hipFree(dJ); hipFree(dXp); hipFree(dlp); hipFree(dlu); hipFree(dlocp); hipFree(dlocu);
    hipFree(dE); hipFree(dY); hipFree(dmu); hipFree(dst); hipFree(dq); hipFree(dred);
    if (engine_retry(f, rc))
        return cocons_sim_cond_dense(f, theta, mean, z_col, m, locs_pred, X_pred, locs_unobs, nsim, iiderrors, out);
    return rc;
}

// ---------------------------------------------------------------------------
extern "C" int cocons_chol_solve(int n, const double *Ain, int nrhs, const double *rhs,
                                 double *L, double *Y, double *logdet_half)
{
    if (n <= 0 || !Ain || nrhs < 0 || (nrhs > 0 && !rhs)) return fail(-1, "cocons_chol_solve: bad argument");
    // reuse the fit machinery with a dummy 1-column design
    std::vector<double> locs((size_t)2 * n, 0.0), X((size_t)n, 1.0);
    double sl[2] = {0.5, 0.5};
    cocons_fit *f = fit_create_impl(n, 1, 0, 0, locs.data(), X.data(), nullptr, nullptr, sl, -1, false);   // caller's order,
    if (!f) return -1;                                                                                       // padding behind
    f->engine_ok = false;     // one-shot handle whose input is uploaded once: plain schedule
    int rc = 0;
    do {
        if ((rc = fit_alloc_matrix(f, nrhs > 0 ? nrhs : 1))) break;
        hipStream_t s = f->stream;
        // identity everywhere in the padded square, zero rhs rows, then copy A and rhs^T in
        std::vector<double> hostA(f->lda * (size_t)f->npad, 0.0);
        for (int c = 0; c < f->npad; ++c) hostA[(size_t)c + (size_t)c * f->lda] = 1.0;
        for (int c = 0; c < n; ++c) {
            for (int r_ = c; r_ < n; ++r_) hostA[(size_t)r_ + (size_t)c * f->lda] = Ain[(size_t)r_ + (size_t)c * n];
            for (int k = 0; k < nrhs; ++k) hostA[(size_t)(f->npad + k) + (size_t)c * f->lda] = rhs[(size_t)c + (size_t)k * n];
        }
        hipError_t e = hipMemcpyAsync(f->dA, hostA.data(), hostA.size() * sizeof(double), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) { rc = fail(-100, "cocons_chol_solve: %s", hipGetErrorString(e)); break; }
        if ((rc = reset_info(f))) break;
        factorize(f, main_view(f), nullptr);
        launch_finalize(f->dA, f->lda, n, f->npad, 0, f->dout, s);
        e = hipMemcpyAsync(hostA.data(), f->dA, hostA.size() * sizeof(double), hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) { rc = fail(-100, "cocons_chol_solve: %s", hipGetErrorString(e)); break; }
        hipMemcpyAsync(f->hout, f->dout, sizeof(double), hipMemcpyDeviceToHost, s);
        hipMemcpyAsync(f->hinfo, f->dinfo, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) { rc = fail(-100, "cocons_chol_solve: %s", hipGetErrorString(e)); break; }
        if ((rc = info_status(f))) break;
        if (logdet_half) *logdet_half = f->hout[0];
        if (L)
            for (int c = 0; c < n; ++c)
                for (int r_ = 0; r_ < n; ++r_)
                    L[(size_t)r_ + (size_t)c * n] = (r_ >= c) ? hostA[(size_t)r_ + (size_t)c * f->lda] : 0.0;
        if (Y)
            for (int k = 0; k < nrhs; ++k)
                for (int c = 0; c < n; ++c) Y[(size_t)c + (size_t)k * n] = hostA[(size_t)(f->npad + k) + (size_t)c * f->lda];
    } while (0);
    cocons_fit_destroy(f);
    return rc;
}

// ---------------------------------------------------------------------------
// Sharded evaluation: Sigma ROW-BLOCK partitioned over the ranks.  Block b = the 256 rows of the
// tiles 2b, 2b+1; blocks are dealt in GROUPS of G consecutive blocks, owner(b) = (b / G) mod world (COCONS_SHARD_GROUP,
// default 4).  A rank assembles, solves and updates ITS rows of every column; per 256-column block k:
//   owner(k)     factors the 256 x 256 diagonal block (all earlier updates of its rows are local)      [0.5 MB]
//   broadcast    L_kk (+ the 4x4-inverse operands of its sixteen 16 x 16 diagonal blocks) from owner(k)
//   every rank   solves ITS rows of the panel, X = B L_kk^-T, in place (solve | in-panel update | solve, row-filtered)
//   owner(k+1)   updates its diagonal block (k+1,k+1) with its own rows of X -- local data -- factors it and starts the
//                broadcast of L_(k+1,k+1): the chain diagonal block -> diagonal block never waits for the bulk exchange
//   all-gather   of the solved rows, packed by owner (every rank contributes B_k / world; on point-to-point links every
//                link then carries B_k / world instead of the whole panel a broadcast would push through each)
//   every rank   updates ITS rows of the trailing matrix with the gathered panel (the column side of a tile belongs to
//                another rank in general: both operands come from the gathered buffer through a per-tile offset table)
// The right-hand sides are rows under the matrix = part of the last block's row block: their owner ends up with
// L^-1 (z - X beta) in place and every rank with every L_kk, so the reductions need no data exchange beyond the usual
// all-reduce of (1 + r^2) doubles and the failing minor.
static const int PT = 2;     // tiles per block

static int shard_group()
{
    static const int g = [] {
        const char *e = getenv("COCONS_SHARD_GROUP");
        int v = e ? atoi(e) : 4;
        return v < 1 ? 1 : v;
    }();
    return g;
}
static inline int shard_owner(int b, int world) { return (b / shard_group()) % world; }

extern "C" int cocons_shard_block_owner(int b, int world) { return (b < 0 || world < 1) ? -1 : shard_owner(b, world); }
extern "C" int cocons_shard_num_blocks(cocons_fit *f) { return f ? (f->nt + PT - 1) / PT : -1; }

static const size_t LKK_DOUBLES = (size_t)PT * TILE * PT * TILE + 2 * 2048;     // diagonal block + Q operands of its two tiles

// host-side plan of one evaluation's exchange: per block k the rows below it, dealt to their owners and packed
struct ShardPlan {
    int nt = 0, mt = 0, world = 0, group = 0;
    std::vector<int> tlo;            // per block: first 64-row tile below the block
    std::vector<int> ncols;          // per block: its columns (256, or 128 for a last block of one tile)
    std::vector<long long> srows;    // per block: rows per slot S_k (64 x the largest number of tiles any rank owns below)
    std::vector<int> pmap;           // nb x T64: element offset of 64-row tile ti in the gathered buffer of block k (-1: above)
    std::vector<int> cnt;            // nb x world: tiles rank w owns below block k
    size_t max_elems = 0;            // largest gathered buffer
};

static void shard_make_plan(ShardPlan &P, int nt, int mt, int world, int group)
{
    P.nt = nt; P.mt = mt; P.world = world; P.group = group;
    const int nb = (nt + PT - 1) / PT, T64 = 2 * mt;
    P.tlo.assign(nb, 0); P.ncols.assign(nb, 0); P.srows.assign(nb, 0);
    P.pmap.assign((size_t)nb * T64, -1); P.cnt.assign((size_t)nb * world, 0);
    P.max_elems = 0;
    for (int k = 0; k < nb; ++k) {
        const int w = (nt - k * PT) < PT ? (nt - k * PT) : PT;
        P.ncols[k] = w * TILE;
        P.tlo[k] = 2 * (k * PT + w);
        int *c = &P.cnt[(size_t)k * world];
        for (int ti = P.tlo[k]; ti < T64; ++ti) c[((ti / 4) / group) % world]++;
        int mx = 0;
        for (int r = 0; r < world; ++r) mx = c[r] > mx ? c[r] : mx;
        P.srows[k] = 64LL * mx;
        std::vector<int> pos(world, 0);
        for (int ti = P.tlo[k]; ti < T64; ++ti) {
            const int o = ((ti / 4) / group) % world;
            P.pmap[(size_t)k * T64 + ti] = (int)((long long)o * P.srows[k] * P.ncols[k] + 64LL * pos[o]++);
        }
        const size_t el = (size_t)world * (size_t)P.srows[k] * (size_t)P.ncols[k];
        if (el > P.max_elems) P.max_elems = el;
    }
}

struct ShardState {
    ShardPlan plan;
    int *d_pmap = nullptr;
    double *lkk[2] = {nullptr, nullptr};
    hipEvent_t ev_main_L = nullptr, ev_comm_L[2] = {nullptr, nullptr}, ev_main_X[2] = {nullptr, nullptr},
               ev_comm_X[2] = {nullptr, nullptr};
    hipEvent_t ev_main_U[2] = {nullptr, nullptr};    // main stream: the received L_kk in lkk[k & 1] has been unpacked (the buffer may be
    bool unpacked[2] = {false, false};               // overwritten by the broadcast of L_(k+2)); unpacked[b]: recorded this evaluation
};

static void shard_state_free(ShardState *S)
{
    if (!S) return;
    hipFree(S->d_pmap); hipFree(S->lkk[0]); hipFree(S->lkk[1]);
    if (S->ev_main_L) hipEventDestroy(S->ev_main_L);
    for (int b = 0; b < 2; ++b) {
        if (S->ev_comm_L[b]) hipEventDestroy(S->ev_comm_L[b]);
        if (S->ev_main_X[b]) hipEventDestroy(S->ev_main_X[b]);
        if (S->ev_comm_X[b]) hipEventDestroy(S->ev_comm_X[b]);
        if (S->ev_main_U[b]) hipEventDestroy(S->ev_main_U[b]);
    }
    delete S;
}

extern "C" int cocons_fit_world(cocons_fit *f) { return (f && f->coll_kind) ? f->coll_world : 1; }

// buffers, plan and events of the sharded evaluation on this handle (world ranks)
static int shard_prepare(cocons_fit *f, int rank, int world)
{
    f->rank = rank; f->world = world; f->nrhs_cur = f->r;
    if (int rc = fit_alloc_matrix(f, f->r)) return rc;
    const int mt = f->nt + f->rhs_act / TILE;
    if (!f->shard) f->shard = new ShardState();
    ShardState *S = f->shard;
    if (S->plan.nt != f->nt || S->plan.mt != mt || S->plan.world != world || S->plan.group != shard_group()) {
        HIPCHK(hipStreamSynchronize(f->stream));
        if (f->cstream) HIPCHK(hipStreamSynchronize(f->cstream));
        shard_make_plan(S->plan, f->nt, mt, world, shard_group());
        if (S->d_pmap) { HIPCHK(hipFree(S->d_pmap)); S->d_pmap = nullptr; }
        HIPCHK(hipMalloc(&S->d_pmap, S->plan.pmap.size() * sizeof(int)));
        HIPCHK(hipMemcpyAsync(S->d_pmap, S->plan.pmap.data(), S->plan.pmap.size() * sizeof(int), hipMemcpyHostToDevice, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
        const size_t bytes = S->plan.max_elems * sizeof(double);
        if (f->xbuf_bytes < bytes) {
            if (f->xbuf_own) { hipFree(f->xbuf[0]); hipFree(f->xbuf[1]); }
            f->xbuf[0] = f->xbuf[1] = nullptr;
            HIPCHK(hipMalloc(&f->xbuf[0], bytes));
            HIPCHK(hipMalloc(&f->xbuf[1], bytes));
            HIPCHK(hipMemsetAsync(f->xbuf[0], 0, bytes, f->stream));      // (slot padding is exchanged too: no NaN patterns)
            HIPCHK(hipMemsetAsync(f->xbuf[1], 0, bytes, f->stream));
            f->xbuf_bytes = bytes; f->xbuf_own = true;
        }
    }
    for (int b = 0; b < 2; ++b) {
        if (!S->lkk[b]) HIPCHK(hipMalloc(&S->lkk[b], LKK_DOUBLES * sizeof(double)));
        if (!S->ev_comm_L[b]) HIPCHK(hipEventCreateWithFlags(&S->ev_comm_L[b], hipEventDisableTiming));
        if (!S->ev_main_X[b]) HIPCHK(hipEventCreateWithFlags(&S->ev_main_X[b], hipEventDisableTiming));
        if (!S->ev_comm_X[b]) HIPCHK(hipEventCreateWithFlags(&S->ev_comm_X[b], hipEventDisableTiming));
        if (!S->ev_main_U[b]) HIPCHK(hipEventCreateWithFlags(&S->ev_main_U[b], hipEventDisableTiming));
        S->unpacked[b] = false;
    }
    if (!S->ev_main_L) HIPCHK(hipEventCreateWithFlags(&S->ev_main_L, hipEventDisableTiming));
    return 0;
}

// assemble this rank's rows of Sigma (lower triangle) and, on their owner, the right-hand-side rows under the matrix
static int shard_begin(cocons_fit *f, const double *theta, const double *mean, int rank, int world)
{
    if (int rc = no_taper(f, "sharded evaluation")) return rc;
    if (f->r < 1) return fail(-1, "sharded evaluation: fit has no z");
    if (int rc = shard_prepare(f, rank, world)) return rc;
    if (int rc = reset_info(f)) return rc;
    if (engine_enabled()) {
        if (int rc = flags_reset(f, f->nt)) return rc;     // (the words shard_factor_diag's engine launches read and raise)
        if (tun().engine_pair)
            if (int rc = mbox_reset(f, f->nt)) return rc;  // (... and the mailboxes of their pair mode)
    }
    ThetaVecs tv;
    make_theta_vecs(theta, f->p, tv);
    ModeSel ms = select_mode(theta, f->p, f->smooth_limits, 0);
    LocArgs la;
    la.n = f->n; la.p = f->p; la.X = f->dX; la.ldx = f->n; la.locs = f->dlocs; la.ldl = f->n;
    la.out = f->dloc; la.stride = f->npad; la.smooth_kind = ms.smooth_kind;
    la.smooth_min = f->smooth_limits[0]; la.smooth_max = f->smooth_limits[1];
    la.th = tv;
    launch_loc_params(la, f->stream);                  // (replicated: O(n p))
    PairArgs pa;
    pa.n = f->n; pa.m = f->n; pa.rows = f->dloc; pa.cols = f->dloc;
    pa.stride = f->npad; pa.stride_rows = f->npad; pa.out = f->dA; pa.ld = f->lda;
    pa.nrows_out = f->npad; pa.ncols_out = f->npad;
    pa.bj0 = f->pad0 / 64; pa.H = 0; pa.blocked = 0;
    pa.gr = ms.gr; pa.nu_fixed = ms.nu_fixed;
    pa.pad_diag = f->nslot > 0 ? 1e300 : 1.0;
    pa.own_world = world; pa.own_rank = rank; pa.own_group = shard_group();
    launch_pair_sym(ms.mode, false, pa, f->stream);
    const int mt = f->nt + f->rhs_act / TILE;
    if (shard_owner(f->nt / PT, world) == rank)         // the rows under the matrix belong to the block of tile row nt
        assemble_rhs(f, mean, true, nullptr, 0, 0, f->npad);
    launch_front_identity(f->dA, f->lda, f->pad0, mt * TILE, f->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

// the owner factors the diagonal block of block k in place and packs it (with the Q operands) for the broadcast
static int shard_factor_diag(cocons_fit *f, int k)
{
    ShardState *S = f->shard;
    const int t = k * PT, w = S->plan.ncols[k] / TILE;
    hipStream_t s = f->stream;
    double *A = f->dA, *q0 = f->dinv, *q1 = f->dinv + 2048;
    if (engine_enabled() && f->flags_cap >= t + w) {
        // the whole block in ONE launch of the diagonal-block engine (tile, strip solve, tile update, tile: what the four
        // launches below do, without their three boundaries -- this block is the chain every rank waits for, section 5): its
        // input words are raised beforehand, so it never waits, and it leaves behind the block of its second tile
        unsigned *in = f->dflags, *out = f->dflags + f->flags_cap, *xr = f->dflags + 2 * (size_t)f->flags_cap;
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(in + t), 7, (size_t)w, s));
        // (pair mode: its two workgroups side by side -- the second tile's factorisation starts ~6 us behind the first's end
        // instead of behind the strip solve and the tile update: 78 -> ~56 us for the block)
        const bool pair = w == 2 && tun().engine_pair && f->dmbox && f->smb_off >= ((size_t)f->nt + 2) * ENGINE_MBOX_DOUBLES;
        launch_potrf_engine(A, f->lda, t, t + w, f->dinv, f->dinfo, in, out, xr, (unsigned *)(f->dinfo + 1),
                            f->dflags + 3 * (size_t)f->flags_cap, s, nullptr, nullptr, 0, nullptr,
                            pair ? f->dmbox : nullptr);
    } else {
        launch_potrf_tile(A, f->lda, t * TILE, q0, f->dinfo, s);
        if (w == 2) {
            launch_trsm_tile(A, f->lda, t * TILE, (t + 1) * TILE, (t + 2) * TILE, q0, s);
            launch_update(A, f->lda, t * TILE, TILE, t + 1, t + 2, t + 1, t + 2, true, s);
            launch_potrf_tile(A, f->lda, (t + 1) * TILE, q1, f->dinfo, s);
        }
    }
    (void)q1;
    double *L = S->lkk[k & 1];
    HIPCHK(hipMemcpy2DAsync(L, (size_t)PT * TILE * sizeof(double), A + (size_t)t * TILE + (size_t)t * TILE * f->lda,
                            f->lda * sizeof(double), (size_t)w * TILE * sizeof(double), (size_t)w * TILE,
                            hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(L + (size_t)PT * TILE * PT * TILE, f->dinv, 2 * 2048 * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipEventRecord(S->ev_main_L, s));
    return 0;
}

// ---- collectives: RCCL on the communication stream, or the caller's transport ----
// COCONS_SHARD_COMM2 (default 1): the broadcasts of the factored diagonal blocks get a stream -- and, under RCCL, a
// communicator -- of their own, so that the chain from one diagonal block to the next never queues behind the bulk exchange
// COCONS_SHARD_COMM2=1: the broadcasts of the factored diagonal blocks get a stream -- and under RCCL a communicator
// (ncclCommSplit) -- of their own.  OPT-IN since round 6 (the advisor's finding): two RCCL communicators working side by side,
// one of them with receivers that sit resident until their owner has factored, have never run with more than one rank (a GPU
// box of this pool has one GPU) -- default: one communicator, one communication stream, the broadcast still issued IN FRONT
// of the all-gather.  tests/test_gpu_configs.py::test_native_sharded_rccl_two_gpus runs both forms where two devices exist.
static bool shard_comm2()
{
    static const int v = [] { const char *e = getenv("COCONS_SHARD_COMM2"); return e ? atoi(e) : 0; }();
    return v != 0;
}

static int coll_prepare(cocons_fit *f)
{
    if (!f->cstream) HIPCHK(hipStreamCreateWithFlags(&f->cstream, hipStreamNonBlocking));
    if (!f->cstream_l) {
        // RCCL serialises the operations of ONE communicator whatever streams they are given: a second stream only helps
        // with a second communicator (split off by the caller of this function); a caller-provided transport has no such rule
        const bool own = shard_comm2() && (f->coll_kind == 2 || (f->coll_kind == 1 && f->comm_l && f->comm_l != f->comm));
        if (own) HIPCHK(hipStreamCreateWithFlags(&f->cstream_l, hipStreamNonBlocking));
        else f->cstream_l = f->cstream;
    }
    if (!f->dcoll) HIPCHK(hipMalloc(&f->dcoll, (size_t)(2 + (COCONS_P_MAX + f->r) * (COCONS_P_MAX + f->r)) * sizeof(double)));
    return 0;
}

extern "C" int cocons_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(-1, "cocons_comm_unique_id: null argument");
    RcclApi *R = rccl_api();
    if (!R) return -1;
    ncclUniqueId id;
    NCCLCHK(R->GetUniqueId(&id));
    static_assert(sizeof(ncclUniqueId) == COCONS_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof id);
    return 0;
}

extern "C" int cocons_fit_comm_init(cocons_fit *f, int nranks, int rank, const void *idp)
{
    FIT_ENTER(f);
    if (int rc = no_taper(f, "cocons_fit_comm_init")) return rc;
    if (!idp || nranks < 1 || rank < 0 || rank >= nranks) return fail(-1, "cocons_fit_comm_init: bad argument");
    if (f->coll_kind) return fail(-1, "cocons_fit_comm_init: the fit already has collectives");
    RcclApi *R = rccl_api();
    if (!R) return -1;
    ncclUniqueId id;
    memcpy(&id, idp, sizeof id);
    // RCCL polls hipGetLastError() after its own launches: a stale (already handled) error code of this
    // process must not be mistaken for a failure of the communicator set-up
    (void)hipGetLastError();
    NCCLCHK(R->CommInitRank(&f->comm, nranks, id, rank));
    f->comm_own = true;
    f->coll_kind = 1; f->coll_rank = rank; f->coll_world = nranks;
    // a second communicator over the same ranks for the small broadcasts on the chain (collective call: every rank is here).
    // Not available / refused: the broadcasts share the one communicator and its stream.
    f->comm_l = nullptr; f->comm_l_own = false;
    if (shard_comm2() && R->CommSplit) {
        ncclComm_t c2 = nullptr;
        if (R->CommSplit(f->comm, 0, rank, &c2, nullptr) == ncclSuccess && c2) { f->comm_l = c2; f->comm_l_own = true; }
        else (void)hipGetLastError();
    }
    return coll_prepare(f);
}

extern "C" int cocons_fit_set_collectives(cocons_fit *f, int rank, int world, cocons_bcast_fn bcast,
                                          cocons_allreduce_fn allreduce, void *user)
{
    FIT_ENTER(f);
    if (int rc = no_taper(f, "cocons_fit_set_collectives")) return rc;
    if (world < 1 || rank < 0 || rank >= world || !bcast || !allreduce)
        return fail(-1, "cocons_fit_set_collectives: bad argument");
    if (f->coll_kind == 1) return fail(-1, "cocons_fit_set_collectives: the fit already has an RCCL communicator");
    f->coll_kind = 2; f->coll_rank = rank; f->coll_world = world;
    f->cb_bcast = bcast; f->cb_allreduce = allreduce; f->cb_user = user;
    return coll_prepare(f);
}

extern "C" int cocons_fit_set_allgather(cocons_fit *f, cocons_allgather_fn allgather)
{
    FIT_ENTER(f);
    if (f->coll_kind != 2) return fail(-1, "cocons_fit_set_allgather: call cocons_fit_set_collectives first");
    f->cb_allgather = allgather;
    return 0;
}

// broadcast of L_kk (packed by shard_factor_diag on its owner) on the communication stream
static int coll_bcast_L(cocons_fit *f, int k, bool in_group)
{
    ShardState *S = f->shard;
    const int b = k & 1, owner = shard_owner(k, f->coll_world);
    hipStream_t cs = f->cstream_l;
    // the owner's copy is packed on its main stream; a receiver's buffer was last read by the unpack of L_(k-2) on ITS main
    // stream (the broadcasts have a stream of their own since round 5: nothing else orders the two)
    if (f->coll_rank == owner) HIPCHK(hipStreamWaitEvent(cs, S->ev_main_L, 0));
    else if (S->unpacked[b]) HIPCHK(hipStreamWaitEvent(cs, S->ev_main_U[b], 0));
    if (f->coll_kind == 1) {
        RcclApi *R = rccl_api();
        NCCLCHK(R->Broadcast(S->lkk[b], S->lkk[b], LKK_DOUBLES, ncclDouble, owner, f->comm_l ? f->comm_l : f->comm, cs));
        if (!in_group) HIPCHK(hipEventRecord(S->ev_comm_L[b], cs));
    } else {
        if (f->cb_bcast(f->cb_user, S->lkk[b], (long long)(LKK_DOUBLES * sizeof(double)), owner, (void *)cs) != 0)
            return fail(-6, "caller-provided broadcast failed");
        HIPCHK(hipEventRecord(S->ev_comm_L[b], cs));
    }
    return 0;
}

// all-gather of the solved rows of panel k: every rank's slot of the owner-packed buffer
static int coll_allgather_X(cocons_fit *f, int k, bool in_group)
{
    ShardState *S = f->shard;
    const int b = k & 1;
    const size_t cnt = (size_t)S->plan.srows[k] * (size_t)S->plan.ncols[k];
    HIPCHK(hipStreamWaitEvent(f->cstream, S->ev_main_X[b], 0));
    if (f->coll_kind == 1) {
        RcclApi *R = rccl_api();
        NCCLCHK(R->AllGather(f->xbuf[b] + (size_t)f->coll_rank * cnt, f->xbuf[b], cnt, ncclDouble, f->comm, f->cstream));
        if (!in_group) HIPCHK(hipEventRecord(S->ev_comm_X[b], f->cstream));
    } else {
        if (!f->cb_allgather) return fail(-6, "caller-provided transport has no all-gather (cocons_fit_set_allgather)");
        if (f->cb_allgather(f->cb_user, f->xbuf[b], (long long)(cnt * sizeof(double)), (void *)f->cstream) != 0)
            return fail(-6, "caller-provided all-gather failed");
        HIPCHK(hipEventRecord(S->ev_comm_X[b], f->cstream));
    }
    return 0;
}

// One rank's part of step k up to the exchange of the solved rows:
//   L_kk in place (received: unpacked) | solve the own rows below | [owner of block k+1] diagonal block k+1 updated with
//   its own rows, factored, packed | own rows packed into the gathered buffer
static int shard_step_pre(cocons_fit *f, int k, int nb)
{
    ShardState *S = f->shard;
    const ShardPlan &P = S->plan;
    const int W = f->coll_world, rank = f->coll_rank, G = shard_group();
    const int t = k * PT, w = P.ncols[k] / TILE, tn = t + w;           // tn: first tile below / right of the block
    const int mt = P.mt;
    hipStream_t s = f->stream;
    double *A = f->dA;
    if (rank != shard_owner(k, W)) {
        HIPCHK(hipStreamWaitEvent(s, S->ev_comm_L[k & 1], 0));
        const double *L = S->lkk[k & 1];
        HIPCHK(hipMemcpy2DAsync(A + (size_t)t * TILE + (size_t)t * TILE * f->lda, f->lda * sizeof(double), L,
                                (size_t)PT * TILE * sizeof(double), (size_t)w * TILE * sizeof(double), (size_t)w * TILE,
                                hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(f->dinv, L + (size_t)PT * TILE * PT * TILE, 2 * 2048 * sizeof(double), hipMemcpyDeviceToDevice, s));
        HIPCHK(hipEventRecord(S->ev_main_U[k & 1], s));
        S->unpacked[k & 1] = true;
    }
    if (tn >= mt) return 0;                                            // nothing below the block
    if (P.cnt[(size_t)k * W + rank] > 0) {
        launch_trsm_tile(A, f->lda, t * TILE, tn * TILE, mt * TILE, f->dinv, s, nullptr, nullptr, -1, 0, W, rank, G);
        if (w == 2) {
            launch_update_from(A, f->lda, A + (size_t)t * TILE * f->lda, f->lda, TILE, tn, mt, t + 1, t + 2, false, s, G, W, rank);
            launch_trsm_tile(A, f->lda, (t + 1) * TILE, tn * TILE, mt * TILE, f->dinv + 2048, s, nullptr, nullptr, -1, 0, W, rank, G);
        }
    }
    if (tn >= P.nt) return 0;                                          // last block: only right-hand-side rows below, no exchange
    if (k + 1 < nb && rank == shard_owner(k + 1, W)) {
        const int w1 = P.ncols[k + 1] / TILE;
        launch_update(A, f->lda, t * TILE, w * TILE, tn, tn + w1, tn, tn + w1, true, s);     // own rows of X: local
        if (int rc = shard_factor_diag(f, k + 1)) return rc;
    }
    const long long slot = (long long)P.srows[k] * P.ncols[k];
    launch_pack_rows(A, f->lda, t * TILE, w * TILE, f->xbuf[k & 1], (size_t)P.srows[k], S->d_pmap + (size_t)k * 2 * mt, P.tlo[k],
                     2 * mt, slot * rank, slot * (rank + 1), s);
    HIPCHK(hipEventRecord(S->ev_main_X[k & 1], s));
    HIPCHK(hipGetLastError());
    return 0;
}

// ... and behind it: the own rows of the trailing matrix updated with the gathered panel
static int shard_step_post(cocons_fit *f, int k, int nb)
{
    ShardState *S = f->shard;
    const ShardPlan &P = S->plan;
    const int W = f->coll_world, rank = f->coll_rank, G = shard_group();
    const int t = k * PT, w = P.ncols[k] / TILE, tn = t + w;
    if (tn >= P.nt) return 0;
    HIPCHK(hipStreamWaitEvent(f->stream, S->ev_comm_X[k & 1], 0));
    if (P.cnt[(size_t)k * W + rank] > 0) {
        // (the owner of block k + 1 has updated that diagonal block with its own rows already: shard_step_pre)
        const bool ahead = k + 1 < nb && rank == shard_owner(k + 1, W);
        const int skip_lo = ahead ? 2 * tn : 0, skip_hi = ahead ? 2 * (tn + P.ncols[k + 1] / TILE) : 0;
        launch_update_from(f->dA, f->lda, f->xbuf[k & 1], (size_t)P.srows[k], w * TILE, tn, P.mt, tn, P.nt, true, f->stream,
                           G, W, rank, nullptr, -1, nullptr, nullptr, nullptr, -1, 0, 0, 0, 0,
                           S->d_pmap + (size_t)k * 2 * P.mt, skip_lo, skip_hi);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// the reductions: the owner of the right-hand-side rows has L^-1 rhs in place and -- like everybody -- every L_kk
static int shard_finish(cocons_fit *f, double *partial, int *info)
{
    const int nr = f->nrhs_cur, len = 1 + nr * nr;
    for (int i = 0; i < len; ++i) partial[i] = 0.0;
    const bool mine = shard_owner(f->nt / PT, f->coll_world) == f->coll_rank;
    if (mine) {
        launch_finalize(f->dA, f->lda, f->n, f->npad, nr, f->dout, f->stream);
        HIPCHK(hipMemcpyAsync(f->hout, f->dout, (size_t)len * sizeof(double), hipMemcpyDeviceToHost, f->stream));
    }
    HIPCHK(hipMemcpyAsync(f->hinfo, f->dinfo, 2 * sizeof(int), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    if (mine) for (int i = 0; i < len; ++i) partial[i] = f->hout[i];
    if (info) *info = *f->hinfo;
    return 0;
}

static int sharded_eval_impl(cocons_fit *f, const double *theta, const double *mean, double *sum_logliks, double *parts);

// A rank that fails in the middle of the schedule (HIP or RCCL error: status < 0) must not leave its peers blocked in
// the next collective until a watchdog fires: it aborts its communicator, which makes the peers' pending RCCL calls
// fail, and the handle refuses further sharded evaluations.  (status > 0 -- Sigma not positive definite -- is an
// ordinary result that every rank reaches together.)
static int sharded_eval(cocons_fit *f, const double *theta, const double *mean, double *sum_logliks, double *parts)
{
    const int rc = sharded_eval_impl(f, theta, mean, sum_logliks, parts);
    if (rc < 0 && f->coll_kind == 1 && f->comm && f->coll_world > 1) {
        const std::string keep = g_err;
        RcclApi *R = rccl_api();
        if (R && f->comm_l && f->comm_l != f->comm) (void)R->CommAbort(f->comm_l);
        if (R) (void)R->CommAbort(f->comm);
        f->comm = nullptr; f->comm_l = nullptr;
        f->coll_kind = -1;                      // poisoned: see cocons_neg2loglik_dense
        g_err = keep + " (communicator aborted)";
    }
    return rc;
}

static int shard_collect(cocons_fit *f, std::vector<double> &part, double minfo, double *sum_logliks, double *parts)
{
    const int nr = f->r;
    if (minfo != (double)0x7f7f7f7f) {
        int st = (int)minfo;
        st -= f->pad0;
        if (st < 1) st = 1;
        if (st > f->n_user) st = f->n_user;
        g_err = "leading minor not positive";
        return st;
    }
    double total = 0.0;
    for (int c = 0; c < nr; ++c) {
        const double quad = part[1 + c * nr + c];
        total += f->n_user * LOG_2PI + 2 * part[0] + quad;
        if (parts) parts[1 + c] = quad;
    }
    if (parts) parts[0] = part[0];
    *sum_logliks = total;
    return 0;
}

static int sharded_eval_impl(cocons_fit *f, const double *theta, const double *mean, double *sum_logliks, double *parts)
{
    const int rank = f->coll_rank, world = f->coll_world;
    if (int rc = shard_begin(f, theta, mean, rank, world)) return rc;
    const int nb = cocons_shard_num_blocks(f);
    if (rank == shard_owner(0, world))
        if (int rc = shard_factor_diag(f, 0)) return rc;
    if (int rc = coll_bcast_L(f, 0, false)) return rc;
    for (int k = 0; k < nb; ++k) {
        if (int rc = shard_step_pre(f, k, nb)) return rc;
        // (the broadcast of the NEXT diagonal block is issued in FRONT of this block's bulk exchange, in the same order on every
        // rank: it depends on nothing but the owner's own rows -- collectives issued earlier --, so whether the two share a
        // stream, a hardware queue or neither, the chain diagonal block -> diagonal block never waits for an all-gather.
        // Until round 4 it was issued behind the all-gather on the one communication stream, and waited for it.)
        const bool exchange = k * PT + f->shard->plan.ncols[k] / TILE < f->nt;
        if (k + 1 < nb) if (int rc = coll_bcast_L(f, k + 1, false)) return rc;
        if (exchange) if (int rc = coll_allgather_X(f, k, false)) return rc;
        if (int rc = shard_step_post(f, k, nb)) return rc;
    }
    const int nr = f->r, len = 1 + nr * nr;
    std::vector<double> part(len + 1);
    int info = 0;
    if (int rc = shard_finish(f, part.data(), &info)) return rc;
    HIPCHK(hipStreamSynchronize(f->cstream));
    if (f->cstream_l != f->cstream) HIPCHK(hipStreamSynchronize(f->cstream_l));
    double minfo = (double)info;                       // 0x7f7f7f7f = no failing minor (exact in a double)
    if (world > 1) {
        if (f->coll_kind == 1) {
            RcclApi *R = rccl_api();
            HIPCHK(hipMemcpyAsync(f->dcoll, part.data(), (size_t)len * sizeof(double), hipMemcpyHostToDevice, f->cstream));
            HIPCHK(hipMemcpyAsync(f->dcoll + len, &minfo, sizeof(double), hipMemcpyHostToDevice, f->cstream));
            NCCLCHK(R->AllReduce(f->dcoll, f->dcoll, (size_t)len, ncclDouble, ncclSum, f->comm, f->cstream));
            NCCLCHK(R->AllReduce(f->dcoll + len, f->dcoll + len, 1, ncclDouble, ncclMin, f->comm, f->cstream));
            HIPCHK(hipMemcpyAsync(part.data(), f->dcoll, (size_t)(len + 1) * sizeof(double), hipMemcpyDeviceToHost, f->cstream));
            HIPCHK(hipStreamSynchronize(f->cstream));
            minfo = part[len];
        } else {
            if (f->cb_allreduce(f->cb_user, part.data(), len, 0) != 0 || f->cb_allreduce(f->cb_user, &minfo, 1, 1) != 0)
                return fail(-6, "caller-provided all-reduce failed");
        }
    }
    return shard_collect(f, part, minfo, sum_logliks, parts);
}

// ---- one process, several GPUs ---------------------------------------------------------------------
struct cocons_multi {
    int ndev;
    std::vector<cocons_fit *> fits;
    std::vector<ncclComm_t> comms;
};

extern "C" void cocons_multi_destroy(cocons_multi *m)
{
    if (!m) return;
    for (auto f : m->fits) cocons_fit_destroy(f);          // (communicators are not owned by the fits)
    for (auto c : m->comms) rccl_comm_destroy(c);
    delete m;
}

extern "C" cocons_multi *cocons_multi_create(int n, int p, int r, const double *locs, const double *X, const double *z,
                                             const double *smooth_limits, int ndev, const int *devices)
{
    if (ndev < 1 || !devices || r < 1) { fail(-1, "cocons_multi_create: bad argument"); return nullptr; }
    RcclApi *R = rccl_api();
    if (!R) return nullptr;
    cocons_multi *m = new cocons_multi();
    m->ndev = ndev;
    for (int d = 0; d < ndev; ++d) {
        cocons_fit *f = cocons_fit_create(n, p, r, 0, locs, X, z, nullptr, smooth_limits, devices[d]);
        if (!f) { cocons_multi_destroy(m); return nullptr; }
        m->fits.push_back(f);
    }
    // a device listed twice (tests on a one-GPU box) cannot carry an RCCL communicator: such a handle serves
    // the entry points that need no collective (cocons_multi_predict_dense) and refuses the sharded objective
    bool distinct = true;
    for (int a = 0; a < ndev; ++a)
        for (int b = a + 1; b < ndev; ++b)
            if (devices[a] == devices[b]) distinct = false;
    if (!distinct) return m;
    m->comms.assign(ndev, nullptr);
    (void)hipGetLastError();
    ncclResult_t nr = R->CommInitAll(m->comms.data(), ndev, devices);
    if (nr != ncclSuccess) {
        fail(-200, "ncclCommInitAll: %s", R->GetErrorString(nr));
        m->comms.clear();
        cocons_multi_destroy(m);
        return nullptr;
    }
    // second communicators for the chain's broadcasts (one collective call per local rank, inside a group)
    std::vector<ncclComm_t> c2(ndev, nullptr);
    bool split_ok = shard_comm2() && R->CommSplit != nullptr;
    if (split_ok) {
        split_ok = R->GroupStart() == ncclSuccess;
        for (int d = 0; d < ndev && split_ok; ++d) {
            hipSetDevice(devices[d]);
            if (R->CommSplit(m->comms[d], 0, d, &c2[d], nullptr) != ncclSuccess) split_ok = false;
        }
        if (R->GroupEnd() != ncclSuccess) split_ok = false;
        for (int d = 0; d < ndev; ++d) if (!c2[d]) split_ok = false;
        if (!split_ok) { for (auto c : c2) if (c) rccl_comm_destroy(c); (void)hipGetLastError(); }
    }
    for (int d = 0; d < ndev; ++d) {
        cocons_fit *f = m->fits[d];
        f->comm = m->comms[d]; f->comm_own = false;
        f->comm_l = split_ok ? c2[d] : nullptr; f->comm_l_own = split_ok;      // (destroyed with the fit)
        f->coll_kind = 1; f->coll_rank = d; f->coll_world = ndev;
        if (fit_check(f) != 0 || coll_prepare(f) != 0) { cocons_multi_destroy(m); return nullptr; }
    }
    return m;
}

// The calling thread drives every device: each schedule step is enqueued on all devices in turn (all calls
// are asynchronous), the per-panel broadcasts of the ranks are issued inside one RCCL group.
extern "C" int cocons_multi_neg2loglik_dense(cocons_multi *m, const double *theta, const double *mean,
                                             double *sum_logliks, double *parts)
{
    if (!m || !theta || !mean || !sum_logliks) return fail(-1, "cocons_multi_neg2loglik_dense: null argument");
    if (m->comms.empty()) return fail(-1, "cocons_multi_neg2loglik_dense: this handle has no communicator (a device is listed twice)");
    RcclApi *R = rccl_api();
    if (!R) return -1;
    const int W = m->ndev;
    std::vector<std::unique_lock<std::recursive_mutex>> op_locks;       // (this entry point drives the ranks' handles directly)
    for (int d = 0; d < W; ++d) op_locks.emplace_back(*m->fits[d]->op_mu);
    for (int d = 0; d < W; ++d) {
        if (int rc = fit_check(m->fits[d])) return rc;
        if (int rc = shard_begin(m->fits[d], theta, mean, d, W)) return rc;
    }
    const int nb = cocons_shard_num_blocks(m->fits[0]);
    auto grouped = [&](int k, bool gather) -> int {
        NCCLCHK(R->GroupStart());
        int rc_in = 0;
        std::string err_in;
        for (int d = 0; d < W && rc_in == 0; ++d) {
            rc_in = fit_check(m->fits[d]);
            if (rc_in == 0) rc_in = gather ? coll_allgather_X(m->fits[d], k, true) : coll_bcast_L(m->fits[d], k, true);
            if (rc_in != 0) err_in = g_err;
        }
        // the group is closed on EVERY path: an error between ncclGroupStart and ncclGroupEnd used to leave it open, and every
        // later RCCL call of the thread inside it
        const ncclResult_t ge = R->GroupEnd();
        if (rc_in != 0) { g_err = err_in; return rc_in; }
        NCCLCHK(ge);
        for (int d = 0; d < W; ++d) {
            if (int rc = fit_check(m->fits[d])) return rc;
            ShardState *S = m->fits[d]->shard;
            HIPCHK(hipEventRecord(gather ? S->ev_comm_X[k & 1] : S->ev_comm_L[k & 1],
                                  gather ? m->fits[d]->cstream : m->fits[d]->cstream_l));
        }
        return 0;
    };
    {
        cocons_fit *f0 = m->fits[shard_owner(0, W)];
        if (int rc = fit_check(f0)) return rc;
        if (int rc = shard_factor_diag(f0, 0)) return rc;
    }
    if (int rc = grouped(0, false)) return rc;
    for (int k = 0; k < nb; ++k) {
        for (int d = 0; d < W; ++d) {
            if (int rc = fit_check(m->fits[d])) return rc;
            if (int rc = shard_step_pre(m->fits[d], k, nb)) return rc;
        }
        const bool exchange = k * PT + m->fits[0]->shard->plan.ncols[k] / TILE < m->fits[0]->nt;
        if (k + 1 < nb) if (int rc = grouped(k + 1, false)) return rc;     // (in front of the bulk exchange: see sharded_eval_impl)
        if (exchange) if (int rc = grouped(k, true)) return rc;
        for (int d = 0; d < W; ++d) {
            if (int rc = fit_check(m->fits[d])) return rc;
            if (int rc = shard_step_post(m->fits[d], k, nb)) return rc;
        }
    }
    cocons_fit *f0 = m->fits[0];
    const int nr = f0->r, len = 1 + nr * nr;
    std::vector<double> tot(len, 0.0), part(len);
    int info_min = 0x7f7f7f7f;
    for (int d = 0; d < W; ++d) {
        int info = 0;
        if (int rc = fit_check(m->fits[d])) return rc;
        if (int rc = shard_finish(m->fits[d], part.data(), &info)) return rc;
        HIPCHK(hipStreamSynchronize(m->fits[d]->cstream));
        if (m->fits[d]->cstream_l != m->fits[d]->cstream) HIPCHK(hipStreamSynchronize(m->fits[d]->cstream_l));
        for (int i = 0; i < len; ++i) tot[i] += part[i];
        if (info < info_min) info_min = info;
    }
    return shard_collect(f0, tot, (double)info_min, sum_logliks, parts);
}

// Dense kriging with the m prediction locations split over the devices of the handle (BASELINE config C5:
// the right-hand sides shard): every device factors Sigma with its own slice of the
// cross-covariance rows as border -- no exchange at all -- and the slices are concatenated on the host.
// Outputs as cocons_predict_dense.  One host thread per device issues that device's call.
extern "C" int cocons_multi_predict_dense(cocons_multi *m, const double *theta, const double *mean, int z_col,
                                          int mp, const double *locs_pred, const double *X_pred,
                                          double *stochastic, double *quadform)
{
    if (!m || !theta || !mean || mp <= 0 || !locs_pred || !X_pred || !stochastic || !quadform)
        return fail(-1, "cocons_multi_predict_dense: bad argument");
    const int W = m->ndev, p = m->fits[0]->p;
    std::vector<int> rcs(W, 0);
    std::vector<std::string> errs(W);
    std::vector<std::thread> th;
    for (int d = 0; d < W; ++d) {
        const int lo = (int)((long long)mp * d / W), hi = (int)((long long)mp * (d + 1) / W);
        if (hi <= lo) continue;
        th.emplace_back([=, &rcs, &errs]() {
            const int k = hi - lo;
            std::vector<double> lp((size_t)2 * k), Xp((size_t)p * k);       // column-major slices
            for (int c = 0; c < 2; ++c)
                for (int i = 0; i < k; ++i) lp[(size_t)i + (size_t)c * k] = locs_pred[(size_t)(lo + i) + (size_t)c * mp];
            for (int c = 0; c < p; ++c)
                for (int i = 0; i < k; ++i) Xp[(size_t)i + (size_t)c * k] = X_pred[(size_t)(lo + i) + (size_t)c * mp];
            rcs[d] = cocons_predict_dense(m->fits[d], theta, mean, z_col, k, lp.data(), Xp.data(), stochastic + lo, quadform + lo);
            if (rcs[d] != 0) errs[d] = g_err;          // g_err is thread-local
        });
    }
    for (auto &t : th) t.join();
    for (int d = 0; d < W; ++d)
        if (rcs[d] != 0) { g_err = errs[d]; return rcs[d]; }
    return 0;
}

// Replica mode inside one process: the nb independent parameter points of one finite-difference gradient
// (R/optim.R:256-259, 1 + 2P points) or of getHessian (R/getFunctions.R:979-1016) are dealt over the devices of the
// handle -- point i goes to device i mod ndev -- and every device runs its share through cocons_neg2loglik_batch on
// its own fit (own slots, own streams), driven by one host thread per device.  No collective: the evaluations are
// independent, so the handle may list a device more than once.  thetas / means / values / status as
// cocons_neg2loglik_batch.
extern "C" int cocons_multi_neg2loglik_batch(cocons_multi *m, int nb, const double *thetas, const double *means,
                                             double *values, int *status)
{
    if (!m || nb < 0 || (nb > 0 && (!thetas || !means || !values || !status)))
        return fail(-1, "cocons_multi_neg2loglik_batch: bad argument");
    const int W = m->ndev, p = m->fits[0]->p, tp = 6 * p;
    for (int i = 0; i < nb; ++i) { values[i] = NAN; status[i] = -1; }
    std::vector<int> rcs(W, 0);
    std::vector<std::string> errs(W);
    std::vector<std::thread> th;
    for (int d = 0; d < W; ++d) {
        const int cnt = nb > d ? (nb - d + W - 1) / W : 0;
        if (cnt == 0) continue;
        th.emplace_back([=, &rcs, &errs]() {
            std::vector<double> T((size_t)cnt * tp), M((size_t)cnt * p), V(cnt);
            std::vector<int> S(cnt);
            for (int j = 0; j < cnt; ++j) {
                const int i = d + j * W;
                memcpy(&T[(size_t)j * tp], thetas + (size_t)i * tp, (size_t)tp * sizeof(double));
                memcpy(&M[(size_t)j * p], means + (size_t)i * p, (size_t)p * sizeof(double));
            }
            rcs[d] = cocons_neg2loglik_batch(m->fits[d], cnt, T.data(), M.data(), V.data(), S.data());
            if (rcs[d] != 0) errs[d] = g_err;          // g_err is thread-local
            for (int j = 0; j < cnt; ++j) { values[d + j * W] = V[j]; status[d + j * W] = S[j]; }
        });
    }
    for (auto &t : th) t.join();
    for (int d = 0; d < W; ++d)
        if (rcs[d] != 0) { g_err = errs[d]; return rcs[d]; }
    return 0;
}

// devices the communicators of a multi handle span (0: the handle has none -- a device is listed twice), and the
// size RCCL itself reports for the communicator of the handle's first device (ncclCommCount)
extern "C" int cocons_multi_comm_ranks(cocons_multi *m, int *ndev, int *rccl_count)
{
    if (!m) return fail(-1, "cocons_multi_comm_ranks: null handle");
    if (ndev) *ndev = m->ndev;
    int cnt = 0;
    if (!m->comms.empty()) {
        RcclApi *R = rccl_api();
        if (!R) return -1;
        NCCLCHK(R->CommCount(m->comms[0], &cnt));
    }
    if (rccl_count) *rccl_count = cnt;
    return 0;
}

// the same for a fit that carries a communicator of its own (cocons_fit_comm_init): what ncclCommCount and
// ncclCommUserRank / ncclCommCuDevice say -- the proof bench.py prints that RCCL saw N ranks on N devices
extern "C" int cocons_fit_comm_info(cocons_fit *f, int *count, int *user_rank, int *device)
{
    if (!f) return fail(-1, "cocons_fit_comm_info: null handle");
    int c = 0, u = -1, dv = -1;
    if (f->coll_kind == 1 && f->comm) {
        RcclApi *R = rccl_api();
        if (!R) return -1;
        NCCLCHK(R->CommCount(f->comm, &c));
        NCCLCHK(R->CommUserRank(f->comm, &u));
        NCCLCHK(R->CommCuDevice(f->comm, &dv));
    } else if (f->coll_kind == 2) {
        c = f->coll_world; u = f->coll_rank; dv = f->device;
    }
    if (count) *count = c;
    if (user_rank) *user_rank = u;
    if (device) *device = dv;
    return 0;
}

// ---------------------------------------------------------------------------
// diagnostic: the persistent launch of the dependency-driven schedule (dag_kernel) REPLAYED ALONE -- the same task list, the
// same products, the same C traffic, but nobody to wait for: what the engine would publish while the launch runs (the
// inverses W of the diagonal tiles, the strips X(t+1,t), the raised out[] / xr[] words) is put there beforehand, taken from
// a factorisation of the same matrix on the plain schedule.  This is what makes the launch countable: rocprofv3 --pmc
// serialises kernels, and the real launch waits for the engine on the other stream.
// Sequence: (1) evaluation on the plain schedule -> the complete factor L in the handle's buffer; (2) W(t) = L(t,t)^-1
// (host, 128 x 128 triangular) and L(t+1,t) for every diagonal block of the head into the W buffer / the second buffer P, and
// a copy of L for the check; (3) per repetition: Sigma assembled again, first panel by the classic kernels, task words zeroed,
// in[] / out[] / xr[] raised, dag_kernel launched between two events.  The launch leaves the diagonal blocks updated but
// unfactored (no engine), so no value comes out of a replay; the check is the panels it formed against the plain factor.
// out[0] = mean duration of the launch in ms, out[1] = its update flops (as bench.py counts them), out[2] = max |P - L| over
// the panels the launch formed relative to max |L| there, out[3] = tasks, out[4] = steps.
__global__ void __launch_bounds__(256)
panel_diff_kernel(const double *P, const double *L, size_t lda, int c0, int c1, int rend, unsigned long long *out)
{
    const int c = c0 + (int)blockIdx.x;
    if (c >= c1) return;
    const int r0 = 2 * TILE * (c / (2 * TILE) + 1);          // first row below column c's diagonal block
    double md = 0.0, ml = 0.0;
    for (int r = r0 + (int)threadIdx.x; r < rend; r += (int)blockDim.x) {
        const double l = L[(size_t)r + (size_t)c * lda], d = fabs(P[(size_t)r + (size_t)c * lda] - l);
        md = d > md || d != d ? d : md;
        ml = fabs(l) > ml ? fabs(l) : ml;
    }
    // (non-negative doubles order like their bit patterns; a NaN difference has the largest pattern of all)
    atomicMax(out, (unsigned long long)__double_as_longlong(md));
    atomicMax(out + 1, (unsigned long long)__double_as_longlong(ml));
}

extern "C" int cocons_debug_dag_replay(cocons_fit *f, const double *theta, const double *mean, int reps, double *out)
{
    FIT_ENTER(f);
    if (int rc = no_taper(f, "cocons_debug_dag_replay")) return rc;
    if (!theta || !mean || !out || reps < 1) return fail(-1, "cocons_debug_dag_replay: bad argument");
    if (f->r < 1 || f->coll_kind) return fail(-1, "cocons_debug_dag_replay: needs a plain dense fit with z");
    // (1) the factor, on the plain schedule
    const int engine_saved = tun().engine;
    tun().engine = 0;
    int st = enqueue_eval(f, theta, mean, true, nullptr, 0, nullptr, false);
    if (st == 0) { HIPCHK(hipStreamSynchronize(f->stream)); st = info_status(f); }
    tun().engine = engine_saved;
    if (st) return st;
    const int nrhs = f->r;
    const bool slots = f->nslot >= nrhs && f->nslot > 0;
    FactorView fv = main_view(f);
    if (slots) fv.mt = fv.nt; else fv.trim = (f->rhs_act - nrhs >= 64) ? 1 : 0;
    fv.dag_ok = true;
    if (!(tun().dag != 0 && !fv.hi && !fv.skew && fv.nt > 4)) return fail(-1, "cocons_debug_dag_replay: the DAG schedule does not apply to this fit");
    if (int rc = flags_reset(f, fv.nt)) return rc;
    if (int prc = dag_prepare(f, fv)) return prc;
    if (f->dag_nsteps < 2) return fail(-1, "cocons_debug_dag_replay: problem too small for a DAG head");
    hipStream_t M = f->stream;
    const size_t lda = fv.lda;
    const int nt_head = 2 * f->dag_nsteps + 2;                 // diagonal tiles 2 .. nt_head - 1 belong to the head's blocks
    // (2) what the engine would publish
    double *Lcopy = nullptr;
    HIPCHK(hipMalloc(&Lcopy, lda * (size_t)f->npad * sizeof(double)));
    int rc = 0;
    std::vector<double> tile((size_t)TILE * TILE), W((size_t)TILE * TILE);
    unsigned long long *dcmp = nullptr;
    do {
#define CKR(expr) { hipError_t e__ = (expr); if (e__ != hipSuccess) { rc = fail(-100 - (int)e__, "cocons_debug_dag_replay: %s", hipGetErrorString(e__)); break; } }
        CKR(hipMemcpyAsync(Lcopy, f->dA, lda * (size_t)f->npad * sizeof(double), hipMemcpyDeviceToDevice, M));
        bool bad = false;
        for (int t = 2; t < nt_head && t < fv.nt && !bad; ++t) {
            const double *src = f->dA + (size_t)t * TILE + (size_t)t * TILE * lda;
            if (hipMemcpy2DAsync(tile.data(), TILE * sizeof(double), src, lda * sizeof(double), TILE * sizeof(double), TILE,
                                 hipMemcpyDeviceToHost, M) != hipSuccess || hipStreamSynchronize(M) != hipSuccess) { bad = true; break; }
            std::fill(W.begin(), W.end(), 0.0);
            for (int j = 0; j < TILE; ++j)                     // column j of W = L^-1: forward substitution on e_j
                for (int i = j; i < TILE; ++i) {
                    double sacc = i == j ? 1.0 : 0.0;
                    for (int k = j; k < i; ++k) sacc -= tile[(size_t)i + (size_t)k * TILE] * W[(size_t)k + (size_t)j * TILE];
                    W[(size_t)i + (size_t)j * TILE] = sacc / tile[(size_t)i + (size_t)i * TILE];
                }
            if (hipMemcpyAsync(f->dWt + (size_t)t * TILE * TILE, W.data(), W.size() * sizeof(double), hipMemcpyHostToDevice, M) != hipSuccess ||
                hipStreamSynchronize(M) != hipSuccess) { bad = true; break; }
            if ((t & 1) == 0 && t + 1 < fv.nt) {               // X(t+1,t) = L(t+1,t): the engine's second copy, in P
                const size_t off = (size_t)(t + 1) * TILE + (size_t)t * TILE * lda;
                if (hipMemcpy2DAsync(f->dP + off, lda * sizeof(double), f->dA + off, lda * sizeof(double), TILE * sizeof(double), TILE,
                                     hipMemcpyDeviceToDevice, M) != hipSuccess) { bad = true; break; }
            }
        }
        if (bad) { rc = fail(-100, "cocons_debug_dag_replay: copying the diagonal tiles failed"); break; }
        CKR(hipMalloc(&dcmp, 2 * sizeof(unsigned long long)));
        // (3) the launch, alone
        const size_t T64 = 2 * (size_t)fv.mt;
        unsigned *in = f->dflags, *outw = f->dflags + f->flags_cap, *xr = f->dflags + 2 * (size_t)f->flags_cap;
        unsigned *abort_word = (unsigned *)(f->dinfo + 1);
        unsigned *queue = f->ddag, *tdone = f->ddag + 64, *pdone = tdone + T64 * (T64 + 1) / 2;
        unsigned *pall = pdone + ((size_t)f->dag_nsteps + 2) * T64;
        unsigned *dcount = pall + (size_t)f->dag_nsteps + 64;
        f->upd_flops = 0.0;
        f->nrhs_cur = nrhs;
        for (int s2 = 0; s2 < f->dag_nsteps; ++s2) count_update_flops(f, 2, 2 * s2 + 2);
        const double flops = f->upd_flops;
        hipEvent_t ea = nullptr, eb = nullptr;
        CKR(hipEventCreate(&ea));
        CKR(hipEventCreate(&eb));
        double ms_sum = 0.0;
        for (int it = 0; it < reps && rc == 0; ++it) {
            if ((rc = reset_info(f))) break;
            assemble_sigma(f, theta, 0, 0, f->npad);
            assemble_rhs(f, mean, true, nullptr, 0, 0, f->npad, true, slots);
            launch_front_identity(fv.A, fv.lda, f->pad0, fv.mt * TILE, M);
            panel_ops(f, fv, 0, M);
            CKR(hipMemsetAsync(f->ddag, 0, f->ddag_words * sizeof(unsigned), M));
            CKR(hipMemsetD32Async((hipDeviceptr_t)f->dflags, 0x3fffffff, 3 * (size_t)f->flags_cap, M));   // in / out / xr: all raised
            // (as many workgroups take part as in a real evaluation: the engine and its partner are entered on XCD 0 by hand)
            unsigned *alive_w = f->dflags + 3 * (size_t)f->flags_cap;
            static const unsigned pair_on_xcd0 = 2u;
            CKR(hipMemcpyAsync(alive_w + 16, &pair_on_xcd0, sizeof(unsigned), hipMemcpyHostToDevice, M));
            CKR(hipEventRecord(ea, M));
            launch_dag(fv.A, fv.lda, f->dP, f->dWt, (const DagStepHost *)f->ddag_steps, f->dag_nsteps, f->dag_ntasks, queue, tdone,
                       pdone, (int)T64, pall, f->dpart, dcount, in, outw, xr, abort_word, M, nullptr, alive_w, dag_xcc_quota(), nullptr,
                       f->dag_have_ftab ? f->ddag_ftab : nullptr, f->dag_xcd_g, f->ddag + f->ddag_xcnt_off);
            CKR(hipEventRecord(eb, M));
            CKR(hipGetLastError());
            CKR(hipStreamSynchronize(M));
            float ms = 0;
            CKR(hipEventElapsedTime(&ms, ea, eb));
            ms_sum += ms;
        }
        if (ea) hipEventDestroy(ea);
        if (eb) hipEventDestroy(eb);
        if (rc) break;
        CKR(hipMemcpyAsync(f->hinfo, f->dinfo, 2 * sizeof(int), hipMemcpyDeviceToHost, M));
        CKR(hipStreamSynchronize(M));
        if (f->hinfo[1] != 0) { rc = fail(ENGINE_ABORT, "cocons_debug_dag_replay: a wait of the replayed launch ran out"); break; }
        // the check: every panel the launch formed (blocks 1 .. nsteps - 1) against the plain factor
        CKR(hipMemsetAsync(dcmp, 0, 2 * sizeof(unsigned long long), M));
        const int c0 = 2 * TILE, c1 = 2 * TILE * f->dag_nsteps, rend = fv.mt * TILE - 64 * fv.trim;
        hipLaunchKernelGGL(panel_diff_kernel, dim3(c1 - c0), dim3(256), 0, M, (const double *)f->dP, (const double *)Lcopy, lda, c0, c1, rend, dcmp);
        unsigned long long h[2] = {0, 0};
        CKR(hipMemcpyAsync(h, dcmp, sizeof h, hipMemcpyDeviceToHost, M));
        CKR(hipStreamSynchronize(M));
        double md, ml;
        memcpy(&md, &h[0], 8); memcpy(&ml, &h[1], 8);
        out[0] = ms_sum / reps; out[1] = flops; out[2] = ml > 0 ? md / ml : NAN; out[3] = (double)f->dag_ntasks; out[4] = (double)f->dag_nsteps;
#undef CKR
    } while (0);
    hipFree(Lcopy); hipFree(dcmp);
    f->border_clean = -1; f->border_pending = -1;        // (the buffer holds a half-done factorisation)
    f->dag_used = false;
    return rc;
}

// ---------------------------------------------------------------------------
// diagnostic: the device Matern correlation 2^(1-nu)/Gamma(nu) u^nu K_nu(u) at n points (host in/out)
extern "C" int cocons_debug_matern(int n, const double *nu, const double *u, double *out)
{
    if (n <= 0 || !nu || !u || !out) return fail(-1, "cocons_debug_matern: bad argument");
    hipStream_t s = nullptr;
    double *d = nullptr;
    int rc = 0;
    do {
        hipError_t e;
#define CKD(expr) if ((e = (expr)) != hipSuccess) { rc = fail(-100 - (int)e, "cocons_debug_matern: %s", hipGetErrorString(e)); break; }
        CKD(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        CKD(hipMalloc(&d, (size_t)3 * n * sizeof(double)));
        CKD(hipMemcpyAsync(d, nu, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
        CKD(hipMemcpyAsync(d + n, u, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
        launch_matern_points(n, d, d + n, d + 2 * (size_t)n, s);
        CKD(hipGetLastError());
        CKD(hipMemcpyAsync(out, d + 2 * (size_t)n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
        CKD(hipStreamSynchronize(s));
#undef CKD
    } while (0);
    if (s) { hipStreamSynchronize(s); hipStreamDestroy(s); }
    hipFree(d);
    return rc;
}

