#!/usr/bin/env python3
"""bench.py -- headline benchmark of the dense hot path on MI355X.

Metric (BASELINE.json): -2 log-likelihood evaluations per second at n = 10 000 (100 x 100
grid, full nonstationary cov_rns: std.dev, scale, aniso, tilt, smooth ~ 1 + x + y, nu in
[0.5, 2.5], nugget), i.e. one "step" = one GetNeg2loglikelihood core evaluation =
covariance assembly + bordered Cholesky (factorisation + solve fused) + reductions, with the
fit's inputs (locs, X, z) already resident in HBM.  Cholesky fp64 TFLOP/s is reported
beside it (n^3/3 flop per evaluation).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--n 10000] [--mode shard|replica]

N = 1 : one GPU evaluates the whole thing.
N > 1 : one rank per GPU.  Started by torch.distributed.run (the ranks read RANK / LOCAL_RANK /
        WORLD_SIZE) or bare (`python bench.py --gpus N`): then this process starts the N ranks itself
        and never touches a GPU.  Default mode "shard": Sigma's 256-row blocks are dealt over the ranks
        in groups and the evaluation is the library's native sharded one (per block: broadcast of the
        factored diagonal block, every rank solves its rows, all-gather of the solved rows, every rank
        updates its rows -- RCCL on its own stream, csrc/api.hip `sharded_eval`); torch.distributed
        only carries the 128-byte RCCL unique id, the barriers and the timing reduction (gloo).  The
        line carries the critical-path model's PREDICTION for this N beside the measurement
        (`shard_prediction`, cocons_amd/shard_model.py).
        Total work is fixed -> "scaling": "strong".  If the sharded path fails the run FAILS (non-zero
        exit, no JSON value): a replica number is never reported in its place.
        Mode "replica" (explicit): every rank evaluates its own theta (what optimParallel's 1+2P
        finite-difference points are) -> "weak".

Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X dense fp64 matrix peak (= fp64 vector peak): 256 CU x 4 SIMD
#                                  x 32 flop/clk x 2.4 GHz; v_mfma_f64_16x16x4_f64 issues every 64 cycles
MFMA_UTIL_PROFILE = "r03_update_kernel_mfma_util.json"      # matrix pipe busy fraction of the update kernel (rocprofv3 --pmc)
TRAFFIC_PROFILE = "r03_update_kernel_hbm_traffic.json"
# the persistent launch (dag_kernel) replayed alone under the counters (tools/dag_replay.py, `tools/gpu_run.sh pmc_dag`): the newest
# round's pair that is committed under profiles/
def _newest_profile(suffix):
    for tag in ("r06", "r05"):
        if os.path.exists(os.path.join(ROOT, "profiles", tag + suffix)):
            return tag + suffix
    return "r05" + suffix


DAG_UTIL_PROFILE = _newest_profile("_dag_kernel_mfma_util.json")
DAG_TRAFFIC_PROFILE = _newest_profile("_dag_kernel_hbm_traffic.json")


def kernel_source_sha16():
    """sha256 (first 16 hex digits) of the source of the factorisation kernels: a stored counter profile records it
    (tools/summarize_pmc_dag.py), and the roofline block says whether the profile still describes the kernel being timed."""
    import hashlib
    with open(os.path.join(ROOT, "cocons_amd", "csrc", "chol.hip"), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def chol_flops(n):
    return n ** 3 / 3.0


def cpu_baseline(n, locs, X, th, z, want_value=True):
    """The oracle (CPU restatement of src/cocons_full.cpp + LAPACK) timed on this box's host
    cores, single-threaded like the reference's serial loop: ONE full evaluation at n."""
    from oracle import oracle as O
    from cocons_amd import workloads as wl
    O.build()
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                      # pragma: no cover
        threadpool_limits = None
    from scipy.linalg import lapack, solve_triangular
    t0 = time.perf_counter()
    S = O.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    t1 = time.perf_counter()
    # for context: the same dpotrf with every host core LAPACK will use
    S2 = S.copy(order="F")
    lapack.dpotrf(np.eye(256) * 2.0, lower=0)             # spin the BLAS thread pool up first
    t4 = time.perf_counter()
    lapack.dpotrf(S2, lower=0, clean=0, overwrite_a=1)
    t_all = time.perf_counter() - t4
    del S2
    t1b = time.perf_counter()
    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    try:
        R, info = lapack.dpotrf(S, lower=0, clean=0, overwrite_a=1)
        t2 = time.perf_counter()
        y = solve_triangular(R, z - X @ th["mean"], trans="T", lower=False, check_finite=False)
        t3 = time.perf_counter()
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    val = n * math.log(2 * math.pi) + 2 * float(np.sum(np.log(np.diag(R)))) + float(y @ y)
    t_cov, t_chol, t_solve = t1 - t0, t2 - t1b, t3 - t2
    return {
        "value": 1.0 / (t_cov + t_chol + t_solve), "unit": "evals/s", "cores": 1, "kind": "port",
        "sample": "1 full evaluation at n=%d: oracle cov_rns (gcc -O2, serial like the reference) %.2fs + "
                  "LAPACK dpotrf %.2fs + dtrtrs %.2fs, 1 thread" % (n, t_cov, t_chol, t_solve),
        "host_cpus": os.cpu_count(),
        "dpotrf_all_host_cores_s": round(t_all, 3),
    }, (val if info == 0 else float("nan"))


def other_configs(device, steps_c2=60, evals_c4=50):
    """BASELINE configs C2, C4 and C5 on this GPU, measured after the timed region of the headline (C3) -- supplementary
    numbers on the same line, never `value`.
      C2  64 x 64 grid, n = 4096, full nonstationary model: sequential -2 loglik evaluations/s, stages, Cholesky TFLOP/s.
      C4  the same problem inside L-BFGS-B as cocoOptim configures it (R/optim.R:237-259, R/profile.R:11-18: central
          differences, 1 + 2P points per gradient), ~50 evaluations: evaluations/s with the points one after the other and
          with each gradient's points through the batch entry -- and the HOST / DEVICE split: the very parameter vectors the
          optimiser visited are evaluated again (a) through the C ABI alone (cocons_neg2loglik_dense: device + one launch
          sequence + one synchronisation each) and (b) through the full host closure (getModelLists, penalty, ctypes) without
          the optimiser, so that what the loop costs beyond the device is attributed: closure plumbing, optimiser, the rest.
      C5  cocoPredict core, n = m = 8192 (128 x 64 grid, half-cell-shifted prediction grid): wall ms, TFLOP/s on
          n^3/3 + n^2 m (R/predict.R:136-183)."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from optim_loop import lbfgsb_central
    out = {}
    # ---- C2
    g = 64
    n = g * g
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    z = wl.synthetic_z(n)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS, device=device)
    for _ in range(5):
        fit.neg2loglik_core(th)
    t0 = time.perf_counter()
    for _ in range(steps_c2):
        v2 = fit.neg2loglik_core(th)[0]
    dt = (time.perf_counter() - t0) / steps_c2
    st = fit.profile_stages(th, reps=5)
    ctf = chol_flops(n) / (st["cholesky_ms"] * 1e-3) / 1e12
    out["C2"] = {"workload": "64x64 grid n=4096, p=3, full nonstationary cov_rns, dense -2loglik, sequential",
                 "evals_per_s": round(1.0 / dt, 2), "ms_per_eval": round(1e3 * dt, 4), "evals": steps_c2,
                 "stages_ms": {k: round(st[k], 4) for k in ("assembly_ms", "cholesky_ms", "reduce_ms", "eval_ms")},
                 "cholesky_tflops_fp64": round(ctf, 3), "cholesky_frac": round(ctf / FP64_MFMA_PEAK_TFLOPS, 4),
                 "host_turnaround_ms": round(1e3 * dt - st["eval_ms"], 4), "neg2loglik": v2,
                 "engine": fit.engine_state()}
    # ---- C4
    pp = wl.par_pos_full()
    x0 = wl.theta_vector_from_lists(th, pp) + 0.1
    lam = (0.0, 0.0, 0.0)
    visited = []

    def fn(t):
        visited.append(np.array(t, dtype=float))
        return ca.GetNeg2loglikelihood(t, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)

    def fnb(ts):
        return ca.GetNeg2loglikelihood_batch(ts, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)

    fn(x0)
    visited.clear()
    t0 = time.perf_counter()
    res = lbfgsb_central(fn, x0, x0 - 3, x0 + 3, max_evals=evals_c4)
    dt_loop = time.perf_counter() - t0
    pts = list(visited)
    tls = [ca.getModelLists(t, pp, "diff") for t in pts]
    # (a) the same points through the C ABI alone; Cholesky failures (a line search may leave the positive definite region)
    #     cost what a success costs and are counted
    fails = 0
    t0 = time.perf_counter()
    for tl in tls:
        try:
            fit.neg2loglik_core(tl)
        except ca.CholeskyError:
            fails += 1
    dt_abi = time.perf_counter() - t0
    # (b) through the host closure, no optimiser
    t0 = time.perf_counter()
    for t in pts:
        ca.GetNeg2loglikelihood(t, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)
    dt_clo = time.perf_counter() - t0
    # (c) device time alone at the first and the last point the optimiser visited that is positive definite
    dev_ms = []
    for tl in (tls[0], tls[-1]):
        try:
            dev_ms.append(round(fit.profile_stages(tl, reps=3)["eval_ms"], 4))
        except ca.CholeskyError:
            dev_ms.append(None)
    nev = max(len(pts), 1)
    fnb([x0, x0])
    t0 = time.perf_counter()
    resb = lbfgsb_central(fn, x0, x0 - 3, x0 + 3, max_evals=evals_c4, fn_batch=fnb)
    dt_b = time.perf_counter() - t0
    out["C4"] = {"workload": "C2's problem inside L-BFGS-B (central differences, P = %d, 1 + 2P points per gradient), ~%d evaluations"
                             % (x0.size, evals_c4),
                 "sequential": {"evals": res["nfev"], "evals_per_s": round(res["nfev"] / dt_loop, 2),
                                "ms_per_eval": round(1e3 * dt_loop / res["nfev"], 4)},
                 "batched_gradient_points": {"evals": resb["nfev"], "evals_per_s": round(resb["nfev"] / dt_b, 2),
                                             "ms_per_eval": round(1e3 * dt_b / resb["nfev"], 4)},
                 "split_ms_per_eval": {"device_eval_ms_first_last_point": dev_ms,
                                       "c_abi_call": round(1e3 * dt_abi / nev, 4),
                                       "host_closure": round(1e3 * (dt_clo - dt_abi) / nev, 4),
                                       "optimiser_and_rest": round(1e3 * (dt_loop - dt_clo) / nev, 4),
                                       "loop_total": round(1e3 * dt_loop / nev, 4)},
                 "cholesky_failures_among_visited_points": fails,
                 "f_start_end": [float(fn(x0)), float(res["fun"])], "engine": fit.engine_state()}
    fit.close()
    # ---- C5
    locs = wl.grid_locs(128, 64)
    sc = wl.design_from_locs(locs)
    X = sc["std.covs"]
    th5 = wl.theta_full()
    th5["mean"] = np.array([0.3, -0.1, 0.2])
    z = wl.synthetic_z(8192)
    lp = locs + np.array([0.5 / 127, 0.5 / 63])
    Xp = wl.design_from_locs(lp, sc["mean.vector"], sc["sd.vector"])["std.covs"]
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS, device=device)
    fit.predict_core(th5, lp, Xp)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        fit.predict_core(th5, lp, Xp)
        ts.append(time.perf_counter() - t0)
    n5 = m5 = 8192
    fl = n5 ** 3 / 3.0 + float(m5) * n5 * n5
    out["C5"] = {"workload": "cocoPredict core n_train = m_pred = 8192 (128x64 grid, half-cell-shifted prediction grid), one GPU",
                 "wall_ms_min": round(1e3 * min(ts), 3), "wall_ms_median": round(1e3 * sorted(ts)[len(ts) // 2], 3),
                 "tflops_fp64": round(fl / min(ts) / 1e12, 2), "frac": round(fl / min(ts) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                 "flops": fl, "note": "n^3/3 + n^2 m flop: bordered Cholesky with the m cross-covariance rows as border; host "
                                      "vectors in, (stochastic, quadratic form) out", "engine": fit.engine_state()}
    fit.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=10000, help="number of locations (square grid edge^2)")
    ap.add_argument("--mode", choices=["shard", "replica"], default="shard")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the supplementary C2 / C4 / C5 measurements")
    ap.add_argument("--inflight", type=int, default=2,
                    help="extra measurement at N=1: this many independent evaluations in flight on the GPU "
                         "(separate fit handles / streams, as optimParallel's workers issue them); 0 = skip")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Started without a launcher: this process never touches the GPU; it starts one fresh rank
        # per GPU (torch.distributed.run), relays rank 0's JSON line and returns the children's code.
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import cocons_amd as ca
    from cocons_amd import workloads as wl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    # rehearsal knob for boxes with a single GPU: every rank on device 0, gloo instead of RCCL
    # (RCCL refuses several ranks on one device).  Never set by the driver.
    rehearsal = os.environ.get("COCONS_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    ctl = None          # gloo group: rendezvous of the RCCL unique id, barriers, timing -- never a panel
    if world > 1:
        # A rank stuck in a collective (a peer died, a link is down) must end the run, loudly, not sit there until
        # somebody else's time-out: every rank arms a watchdog that kills it -- non-zero exit, no JSON -- when the whole
        # multi-GPU bench (normally well under a minute) has not finished in COCONS_BENCH_WATCHDOG_S seconds (default 300).
        import threading

        def _watchdog(limit=float(os.environ.get("COCONS_BENCH_WATCHDOG_S", "300"))):
            time.sleep(limit)
            sys.stderr.write("bench.py: rank %d still running after %.0f s -- a collective is stuck; giving up\n" % (rank, limit))
            sys.stderr.flush()
            os._exit(4)
        threading.Thread(target=_watchdog, daemon=True).start()
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
        ctl = dist.group.WORLD

    g = int(round(math.sqrt(args.n)))
    n = g * g
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    z = wl.synthetic_z(n)
    r = 1

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(group=ctl)
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
        return float(t.item())

    shard_mode = world > 1 and args.mode == "shard"
    if shard_mode:
        from cocons_amd.shard import ShardedFit
        fit = ShardedFit(locs, X, z, wl.SMOOTH_LIMITS, device=local_rank)
        if rehearsal:
            fit.init_host_transport(dist, rank, world, group=ctl)     # ranks share one GPU: gloo is the wire
        else:
            fit.init_rccl(dist, rank, world, group=ctl)               # RCCL communicator inside the library

        def step():
            return fit.neg2loglik_core(th)[0]
    else:
        fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS, device=local_rank)
        if world > 1:      # replica mode: every rank its own finite-difference point
            th = {k: np.array(v, dtype=np.float64) for k, v in th.items()}
            th["std.dev"][0] += 1.22e-4 * rank

        def step():
            return fit.neg2loglik_core(th)[0]

    # N > 1, sharded: what RCCL itself says about the communicator of every rank (ncclCommCount / UserRank / CuDevice)
    comm = None
    if shard_mode:
        mine = fit.comm_info()
        mine["rank_env"], mine["local_rank"], mine["pid"] = rank, local_rank, os.getpid()
        allinfo = [None] * world
        dist.all_gather_object(allinfo, mine, group=ctl)
        comm = {"transport": "gloo host transport (rehearsal)" if rehearsal else "RCCL",
                "rccl_ranks": int(allinfo[0]["count"]), "ranks": allinfo,
                "distinct_devices": len({(a["device"]) for a in allinfo})}
        if not rehearsal and (comm["rccl_ranks"] != world or comm["distinct_devices"] != world):
            sys.exit("bench.py: RCCL reports %d ranks on %d distinct devices, expected %d"
                     % (comm["rccl_ranks"], comm["distinct_devices"], world))

    val = None
    for _ in range(args.warmup):
        val = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        val = step()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    evals = args.steps * (world if (world > 1 and not shard_mode) else 1)
    evals_per_s = evals / dt
    ms_per_step = 1e3 * dt / args.steps
    # did the timed handle run on the schedule it claims?  (a hand-off time-out repeats the evaluation on the plain
    # schedule and is counted: cocons_fit_engine_state)
    es = fit.engine_state()
    engine = {"engine_active": es["active"], "engine_retries": es["retries"], "engine_last_abort": es["last_abort"],
              "switches": {k: os.environ.get(k) for k in ("COCONS_ENGINE", "COCONS_UPD_DYNAMIC", "COCONS_UPD_WAVES",
                                                         "COCONS_UPD_W8_MAX_TILES", "COCONS_DAG", "COCONS_DAG_MIN_TILES", "COCONS_DAG_SPLIT", "COCONS_DAG_XCC_QUOTA")
                           if os.environ.get(k) is not None}}

    # Stage timings and the dominant kernel's roofline (HIP events on the launch stream around every stage and every
    # trailing-update launch, cocons_fit_profile) are taken HERE, straight behind the timed steps and on the same warm
    # device: taken at the end of the run, behind the host-side set-up of the supplementary measurements below (the GPU idle
    # for half a second), the same launches measured 7 % longer while the clocks ramped up again.  For N > 1 on a plain
    # single-GPU handle on rank 0 (the kernels are the same ones).
    st = None
    if rank == 0:
        pfit = fit if not shard_mode else ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS, device=local_rank)
        if pfit is not fit:
            for _ in range(3):
                pfit.neg2loglik_core(th)
        st = pfit.profile_stages(th, reps=3)

    # extra (N>1, sharded mode): the replica mode on the same ranks -- every rank evaluates its
    # own theta with its own fit, no collective in the data path ("weak" scaling).
    replica = None
    if shard_mode:
        rfit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS, device=local_rank)
        rth = {k: np.array(v, dtype=np.float64) for k, v in th.items()}
        rth["std.dev"][0] += 1.22e-4 * rank
        for _ in range(2):
            rfit.neg2loglik_core(rth)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            rfit.neg2loglik_core(rth)
        barrier()
        dtr = max_over_ranks(time.perf_counter() - t1)
        replica = {"evals_per_s": round(args.steps * world / dtr, 4), "scaling": "weak",
                   "note": "every rank evaluates its own theta (no data-path collective)"}
        # the same with each rank's points pipelined through the batch entry
        nbatch = max(args.steps, 6)
        rths = []
        for i in range(nbatch):
            t = {k: np.array(v, dtype=np.float64) for k, v in rth.items()}
            t["std.dev"][0] += 1.22e-4 * (i + 1)
            rths.append(t)
        rfit.neg2loglik_batch_core(rths[:3])
        barrier()
        t1 = time.perf_counter()
        rfit.neg2loglik_batch_core(rths)
        barrier()
        dtb = max_over_ranks(time.perf_counter() - t1)
        replica["batched_evals_per_s"] = round(nbatch * world / dtb, 4)
        rfit.close()

    # extra (N=1 only): throughput with several independent evaluations in flight -- the call
    # pattern of optimParallel's forked workers sharing one GPU (R/optim.R:117-121).  Reported
    # beside `value`, never as `value`.
    inflight = None
    if world == 1 and args.inflight > 1:
        import threading
        fits = [fit] + [ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS, device=local_rank)
                        for _ in range(args.inflight - 1)]
        per = max(args.steps // args.inflight, 1)

        def worker(f):
            for _ in range(per):
                f.neg2loglik_core(th)

        for f in fits:
            f.neg2loglik_core(th)
        torch.cuda.synchronize()
        ts = [threading.Thread(target=worker, args=(f,)) for f in fits]
        t1 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        inflight = {"evaluations_in_flight": args.inflight, "evals_per_s": round(per * args.inflight / dt2, 4),
                    "evals": per * args.inflight}
        for f in fits[1:]:
            f.close()

    # extra (N=1 only): the library's own batch entry (cocons_neg2loglik_batch): the same
    # independent evaluations pipelined over internal slots, no host threads.
    batch = None
    if world == 1 and args.inflight > 1:
        nbatch = max(args.steps, 6)
        ths = []
        for i in range(nbatch):
            t = {k: np.array(v, dtype=np.float64) for k, v in th.items()}
            t["std.dev"][0] += 1.22e-4 * (i + 1)        # distinct finite-difference points
            ths.append(t)
        fit.neg2loglik_batch_core(ths[:3])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        bv, bs = fit.neg2loglik_batch_core(ths)
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t1
        batch = {"evals": nbatch, "evals_per_s": round(nbatch / dtb, 4), "all_ok": bool((bs == 0).all())}

    # extra (N=1, n = 10^4 only): the taper objective (GetNeg2loglikelihoodTaper, type = "sparse") of the same grid with a
    # Wendland-1 taper of range 0.06 (~104 neighbours per row) through the band-limited factorisation; a supplementary
    # number, never `value`
    taper = None
    if world == 1 and n == 10000 and args.inflight > 1:
        try:
            delta = 0.06
            cell = {}
            for i, (x, y) in enumerate(locs):
                cell.setdefault((int(x / delta), int(y / delta)), []).append(i)
            ci, rp, ent = [], [1], []
            for i, (x, y) in enumerate(locs):
                cx, cy = int(x / delta), int(y / delta)
                cand = np.array(sorted(j for a in (-1, 0, 1) for b in (-1, 0, 1) for j in cell.get((cx + a, cy + b), [])))
                d = np.sqrt(np.sum((locs[cand] - locs[i]) ** 2, axis=1))
                keep = d <= delta
                h = d[keep] / delta
                ci.extend((cand[keep] + 1).tolist())
                ent.extend(((1 - h) ** 4 * (4 * h + 1)).tolist())
                rp.append(len(ci) + 1)
            tfit = ca.CoconsTaperFit(locs, X, z, wl.SMOOTH_LIMITS, np.array(ci, dtype=np.int32),
                                     np.array(rp, dtype=np.int32), np.array(ent), device=local_rank)
            for _ in range(3):
                tv, _ = tfit.neg2loglik_core(th)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                tv, _ = tfit.neg2loglik_core(th)
            torch.cuda.synchronize()
            dtt = (time.perf_counter() - t1) / args.steps
            taper = {"workload": "GetNeg2loglikelihoodTaper, same grid, Wendland-1 taper delta=0.06", "nnz": len(ci),
                     "ms_per_eval": round(1e3 * dtt, 4), "evals_per_s": round(1.0 / dtt, 3), "neg2loglik": tv}
            tfit.close()
        except Exception as e:                          # noqa: BLE001 -- supplementary: reported, not fatal
            taper = {"error": repr(e)}

    # extra (N=1, n = 10^4 only): BASELINE's other configurations on the same GPU (C2, C4, C5) -- see other_configs()
    configs = None
    if world == 1 and n == 10000 and not args.no_configs:
        try:
            configs = other_configs(local_rank)
        except Exception as e:                          # noqa: BLE001 -- supplementary: reported, not fatal
            import traceback
            configs = {"error": repr(e), "trace": traceback.format_exc()[-800:]}

    out = None
    if rank == 0:
        # (stage timings and the dominant kernel's roofline: `st`, taken right behind the timed steps above)
        stages = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}
        roofline = None
        flops, launches = st["update_flops"], st["update_launches"]
        if launches > 0 and st["update_sum_ms"] > 0:
            dag = st.get("dag_ms", 0.0) > 0 and st.get("dag_flops", 0.0) > 0
            if dag:
                # the dominant kernel is the ONE persistent launch that runs the head of the factorisation (its trailing updates
                # AND the panels between them, dag_kernel); only the update flops are counted for it
                achieved = st["dag_flops"] / (st["dag_ms"] * 1e-3) / 1e12
                kernel = ("cocons::dag_kernel (persistent, dependency-driven: the trailing updates of the head of the factorisation "
                          "-- steps of >= %s update tiles, %.0f %% of the update flops -- and the panel tasks between them, "
                          "v_mfma_f64_16x16x4_f64); the remaining %d steps: cocons::update_kernel launches"
                          % (os.environ.get("COCONS_DAG_MIN_TILES", "2000"), 100.0 * st["dag_flops"] / flops, launches - 1))
                kflops, kms, klaunches = st["dag_flops"], st["dag_ms"], 1
            else:
                achieved = flops / (st["update_sum_ms"] * 1e-3) / 1e12
                kernel = ("cocons::update_kernel<64, 8, 0, 4> / <64, 16, 0, 8> (trailing SYRK/GEMM, v_mfma_f64_16x16x4_f64; the "
                          "8-wave form for launches of <= 3500 tiles)")
                kflops, kms, klaunches = flops, st["update_avg_ms"], launches
            roofline = {"bound": "mfma",
                        "kernel": kernel,
                        "achieved": round(achieved, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
                        "measured_in": "this run (HIP events around each launch on the launch stream)",
                        "flops_per_launch": kflops / max(klaunches, 1),
                        "launch_ms": round(kms, 4), "launches_per_eval": klaunches,
                        "all_update_launches": {"launches": launches, "sum_ms": round(st["update_sum_ms"], 4), "flops": flops,
                                                "tflops": round(flops / (st["update_sum_ms"] * 1e-3) / 1e12, 3)},
                        "pipe_busy_frac_pmc": None,
                        "pipe_busy_source": "NOT measured in this run: profiles/r03_update_kernel_mfma_util.json "
                                            "(SQ_VALU_MFMA_BUSY_CYCLES over SIMD cycles, all 39 launches; rocprofv3 --pmc "
                                            "serialises kernels, so that pass runs the plain schedule -- the kernels and their "
                                            "per-launch traffic are the same); bare-instruction probes in "
                                            "profiles/r02_mfma_f64_probe.json",
                        "traffic": None}
            mu = os.path.join(ROOT, "profiles", MFMA_UTIL_PROFILE)
            if n == 10000 and os.path.exists(mu):
                with open(mu) as fh:
                    roofline["pipe_busy_frac_pmc"] = round(json.load(fh).get("mfma_busy_over_simd_cycles_all_launches", 0.0), 3)
            tr = os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)
            if n == 10000 and os.path.exists(tr):
                with open(tr) as fh:
                    tj = json.load(fh)
                dtr = os.path.join(ROOT, "profiles", DAG_TRAFFIC_PROFILE)
                dmu = os.path.join(ROOT, "profiles", DAG_UTIL_PROFILE)
                if dag and os.path.exists(dtr) and os.path.exists(dmu):
                    # Round 5: the SAME launch replayed alone under rocprofv3 --pmc (same task list, products and C traffic; what
                    # the engine publishes prepared beforehand -- cocons_debug_dag_replay).  Per launch, separate passes.
                    with open(dtr) as fh:
                        dj = json.load(fh)
                    with open(dmu) as fh:
                        uj = json.load(fh)
                    # the stored counters describe the kernel source they were taken with: say so when that is not this tree's
                    prof_sha = dj.get("kernel_source_sha16")
                    roofline["counter_profile_matches_kernel_source"] = (prof_sha == kernel_source_sha16()) if prof_sha else None
                    roofline["traffic"] = dj.get("hbm_bytes_per_launch")
                    roofline["traffic_high"] = dj.get("hbm_bytes_per_launch_high")
                    roofline["traffic_l2_hit_rate"] = dj.get("l2_hit_rate")
                    # what the K = 256 blocking needs: every C tile of every step read and written once (16 B per element
                    # and step) + the panels once
                    roofline["traffic_algorithmic"] = round(16.0 * st["dag_flops"] / (2.0 * 256.0), 1)
                    roofline["traffic_source"] = ("NOT measured in this run: profiles/%s (FETCH_SIZE raw + WRITE_SIZE of cocons::dag_kernel "
                                                  "replayed alone, `tools/gpu_run.sh pmc_dag`, commit %s); `traffic_high` applies the guide's "
                                                  "x2 to FETCH_SIZE (calibrated for 16 B/lane reads; the C tiles are read 8 B/lane); "
                                                  "`traffic_algorithmic` = C read + write once per step at K = 256"
                                                  % (DAG_TRAFFIC_PROFILE, dj.get("commit", "?")))
                    roofline["pipe_busy_frac_pmc"] = round(uj.get("mfma_busy_over_simd_cycles", 0.0), 3)
                    roofline["pipe_busy_source"] = ("NOT measured in this run: profiles/%s (SQ_VALU_MFMA_BUSY_CYCLES over SIMD cycles of "
                                                    "the replayed launch, %.0f us under the counters, clock %.2f GHz)"
                                                    % (DAG_UTIL_PROFILE, uj.get("launch_us_under_pmc", 0.0),
                                                       uj.get("clock_GHz_from_GRBM_GUI_ACTIVE", 0.0)))
                elif dag:
                    # The persistent launch cannot be counted: rocprofv3 --pmc serialises kernels, and this launch waits for its
                    # partner on the other stream (the diagonal-block engine).  What exists is the count for the SAME tile kernel
                    # in separate launches (the plain schedule, round 3): per evaluation FETCH + WRITE for all 39 updates -- the
                    # share of this launch by flops is an estimate, and it is labelled as one; `traffic` stays null.
                    per_eval = 1e3 * (tj.get("FETCH_SIZE_KB_per_eval_raw", 0.0) + tj.get("WRITE_SIZE_KB_per_eval", 0.0))
                    roofline["traffic_estimate"] = round(per_eval * st["dag_flops"] / flops, 1) if per_eval > 0 else None
                    roofline["traffic_source"] = ("NOT measured: the persistent launch cannot run under rocprofv3 --pmc (kernels are "
                                                  "serialised there and it waits for the engine on the other stream).  "
                                                  "`traffic_estimate` = this launch's share (by flops) of FETCH_SIZE (raw) + "
                                                  "WRITE_SIZE per evaluation of the same tile kernel in 39 separate launches, "
                                                  "profiles/%s (commit %s); its C tiles move by the same L2-bypassing "
                                                  "read-modify-write, its operands through the same L2" % (TRAFFIC_PROFILE,
                                                                                                           tj.get("commit", "?")))
                else:
                    roofline["traffic"] = tj.get("hbm_bytes_per_launch")
                    roofline["traffic_source"] = "NOT measured in this run: profiles/%s (separate rocprofv3 --pmc passes of " \
                                                 "this command, commit %s); %s" % (TRAFFIC_PROFILE, tj.get("commit", "?"),
                                                                                   tj.get("note", ""))
        chol_tf = chol_flops(n) / (stages["cholesky_ms"] * 1e-3) / 1e12
        pairs = n * (n + 1) / 2.0
        assembly = {"kernel": "cocons::pair_sym_kernel<0, false> (general-nu Bessel-K branch)", "bound": "fp64 valu",
                    "ms": stages["assembly_ms"], "pairs_per_s": round(pairs / (stages["assembly_ms"] * 1e-3), 1),
                    "hbm_write_GBps": round(8.0 * pairs / (stages["assembly_ms"] * 1e-3) / 1e9, 2),
                    "note": "lower triangle only, written once into the factorisation buffer (8 B per pair)"}
        cpu = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, cpu_val = cpu_baseline(n, locs, X, th, z)
            parity = abs(val - cpu_val) / abs(cpu_val)
        # N > 1, sharded: what the critical-path model of the row-block exchange predicts for this N, beside the measurement
        # (single-GPU rate in it: this handle's own eval_ms); the first run on a multi-GPU node confirms or refutes it
        shard_pred = None
        if shard_mode:
            from cocons_amd import shard_model
            one = 1e3 / st["eval_ms"] if st and st.get("eval_ms", 0) > 0 else None
            shard_pred = shard_model.predict(n, world, one)
            shard_pred["measured_evals_per_s"] = round(evals_per_s, 4)
            shard_pred["one_gpu_evals_per_s_this_run"] = round(one, 2) if one else None
            shard_pred["note"] = ("model of DESIGN.md section 5 (cocons_amd/shard_model.py): per block max(chain, rank work); "
                                  "kernel times measured on one GPU, link figures assumed; rehearsal runs (ranks sharing one "
                                  "GPU over gloo) are not what it predicts")
        label = {10000: "C3", 4096: "C2"}.get(n, "grid")
        out = {
            "metric": "-2loglik evals/sec (dense cov_rns + Cholesky/solve/log-det, fp64) at n=%d" % n,
            "value": round(evals_per_s, 4), "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if (shard_mode or world == 1) else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %dx%d grid n=%d, p=3, full nonstationary cov_rns (Bessel-K branch), r=1, "
                                   "dense -2loglik" % (label, g, g, n),
                       "parallelism": ("Sigma row blocks sharded x%d, native RCCL broadcast + all-gather%s"
                                       % (world, " [rehearsal: gloo wire, one GPU]" if rehearsal else "")) if shard_mode
                                      else ("replica x%d" % world if world > 1 else "single GPU"),
                       "target_8gpu_evals_per_s": 100},
            "cholesky_tflops_fp64": round(chol_tf, 3),
            "cholesky_frac": round(chol_tf / FP64_MFMA_PEAK_TFLOPS, 4),
            "cholesky_note": "n^3/3 flop / cholesky_ms of ONE GPU (stages_ms); bordered factorisation incl. the solve",
            "stages_ms": stages,
            "engine": engine,
            "comm": comm,
            "shard_prediction": shard_pred,
            "neg2loglik": val,
            "throughput_inflight": inflight,
            "throughput_batch_api": batch,
            "taper_path": taper,
            "configs": configs,
            "replica_mode": replica,
            "parity_rel_err_vs_cpu": parity,
            "roofline": roofline,
            "assembly": assembly,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        try:
            dist.barrier(group=ctl)
            dist.destroy_process_group()
        except Exception:                                       # noqa: BLE001
            pass


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:                                       # noqa: BLE001
        # a failing rank must fail the run (the launcher then stops the other ranks): no JSON line, non-zero exit
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
