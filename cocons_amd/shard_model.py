"""Critical-path model of the row-block sharded evaluation (csrc/api.hip `sharded_eval`): what
`bench.py --gpus N` prints beside its measurement.  Nothing here has been measured on a multi-GPU node -- the kernel
times are single-GPU measurements of this repository (kernel traces of rounds 3-4), the link figures are assumptions
stated below; the point of printing the prediction is that the first run on real hardware confirms or refutes it.

Per 256-column block k every rank runs, in stream order,
    unpack L_kk | solve its rows | pack them | wait for the all-gather | update its rows
while the owner of block k + 1 -- between its solve and its pack -- updates its own diagonal block, factors it and starts
the broadcast of L_(k+1,k+1).  So a step lasts
    max( chain, rank ),   chain = bcast(L) + unpack + solve + diagonal update + diagonal factor      (independent of N)
                          rank  = unpack + solve + pack + allgather(B_k) + U_k / N
The chain term holds no all-gather because the code issues the broadcast of L_(k+1,k+1) IN FRONT of the all-gather of step k
(round 5; until round 4 it was queued behind it, so the chain did wait for the bulk exchange and this model was optimistic).
Since round 6 both share ONE communicator and communication stream by default -- a stream and, under RCCL, a communicator
(ncclCommSplit) of its own for the broadcasts is the opt-in COCONS_SHARD_COMM2=1, never yet run with more than one rank -- so the
broadcast can still queue behind the PREVIOUS step's all-gather where that one has not drained; the model ignores it (that
all-gather started a whole solve | diagonal update | factor earlier).
`predict(..., group_local=True)` is a WHAT-IF, not the code: inside an ownership group (COCONS_SHARD_GROUP = 4 consecutive blocks
of one owner) the next diagonal block needs nothing but its owner's own rows, so the chain of a same-owner transition could run
on the single-GPU machinery (resident engine pair + one-launch panel over the owner's rows: T_CHAIN_LOCAL) and only a group
boundary would pay broadcast + unpack + the three solve launches (not built -- the path has never run
on two GPUs, and the round went into the single-GPU kernel).
What forbids more: the chain's solve (three launches of latency-bound kernels, 45 us: a launch over 4 strips takes as long as one
over 80 -- solving the next diagonal block's 256 rows first would not shorten it) and its diagonal factor (59 us: one launch of
the engine kernel, a pair of workgroups since round 5; the same two tile factorisations that bound the single-GPU tail); at
N = 2 the rank term (one link, half the updates).
"""
from __future__ import annotations

TILE = 128

# single-GPU kernel times, microseconds (launch gaps included where a sequence is meant)
T_UNPACK = 5.0            # two device-to-device copies of 0.5 MB + 32 KB
T_SOLVE = 45.0            # solve | in-panel update | solve over a rank's rows (three launches; strips run side by side)
T_DIAG_UPDATE = 10.0      # 256 x 256 x 256 update of the next diagonal block (10 tiles)
T_DIAG_FACTOR = 66.0      # ONE launch of the diagonal-block engine, its two workgroups side by side (round 5: the second tile's
                          # factorisation starts ~6 us behind the first's end): 59 us measured at n = 10^4,
                          # profiles/r05_shard_one_rank_kernel_stats.csv (one workgroup, round 4: 78) + the fill that raises its
                          # input words + one gap (until the end of round 4: four launches, 30 | 13 | 10 | 30 + gaps = 100)
T_PACK = 10.0
T_CHAIN_LOCAL = 81.0      # what-if (group_local): diagonal block k -> k + 1 of ONE owner on the engine schedule restricted to its rows:
                          # the engine's cycle, 61 us (profiles/r05_timeline_n4096.txt), + the first strips of the panel, ~20
GROUP = 4                 # COCONS_SHARD_GROUP
T_ASSEMBLY_MS = 0.9       # covariance assembly of the whole lower triangle on one GPU (shards: / N)
T_UPDATES_MS = 6.4        # all trailing updates of one evaluation on one GPU at n = 10^4 (scaled by (n / 10^4)^3)
COLL_LATENCY = 20.0       # per collective, microseconds (assumption)
LINK_GBPS = 70.0          # one xGMI link, one direction, effective (assumption; 7 links per GPU, point to point)
L_BYTES = (2 * TILE) ** 2 * 8 + 2 * 2048 * 8


def allgather_gbps(world: int) -> float:
    """Aggregate rate at which one rank RECEIVES in an all-gather: it has world - 1 peers, each on its own link."""
    return LINK_GBPS * max(1, min(world - 1, 7))


def predict(n: int, world: int, single_gpu_evals_per_s: float | None = None, group_local: bool = False) -> dict:
    """Predicted -2 loglik evaluations per second of the sharded evaluation at order n on `world` GPUs."""
    npad = (n + TILE - 1) // TILE * TILE
    nt = npad // TILE
    nb = (nt + 1) // 2
    rows_total = npad + TILE
    upd_total_us = T_UPDATES_MS * 1e3 * (n / 1e4) ** 3
    # trailing update of step k ~ (rows below)^2
    below = [max(0, rows_total - 256 * (k + 1)) for k in range(nb)]
    wsum = sum(b * b for b in below) or 1.0
    chain = COLL_LATENCY + L_BYTES / (LINK_GBPS * 1e3) + T_UNPACK + T_SOLVE + T_DIAG_UPDATE + T_DIAG_FACTOR
    total = T_DIAG_FACTOR + T_ASSEMBLY_MS * 1e3 * (n / 1e4) ** 2 / world
    chain_bound = 0
    for k in range(nb):
        bk = below[k] * 256 * 8.0                                   # bytes of the solved panel
        ag = 0.0 if world == 1 else COLL_LATENCY + bk * (world - 1) / world / (allgather_gbps(world) * 1e3)
        rank = T_UNPACK + T_SOLVE + T_PACK + ag + upd_total_us * below[k] ** 2 / wsum / world
        ch = chain
        if group_local and (k + 1) % GROUP != 0:                     # block k + 1 has the same owner as block k
            ch = T_CHAIN_LOCAL
        step = max(ch if world > 1 else 0.0, rank)
        chain_bound += step == ch
        total += step
    out = {"evals_per_s": round(1e6 / total, 1), "ms_per_eval": round(total * 1e-3, 3), "chain_us_per_block": round(chain, 1),
           "blocks": nb, "blocks_bound_by_the_chain": int(chain_bound),
           "forbidding_term": ("chain" if chain_bound * 2 >= nb else "rank work (updates / N + all-gather over the links)") if world > 1 else None,
           "status": "unverified: no multi-GPU node has run the sharded evaluation yet",
           "assumptions": {"link_GBps_per_direction": LINK_GBPS, "collective_latency_us": COLL_LATENCY,
                           "diag_factor_us": T_DIAG_FACTOR, "solve_us": T_SOLVE, "updates_ms_one_gpu": round(upd_total_us * 1e-3, 2)}}
    if group_local:
        out["what_if"] = "same-owner transitions on the single-GPU machinery (NOT the shipped schedule)"
    if single_gpu_evals_per_s:
        out["vs_one_gpu"] = round(out["evals_per_s"] / single_gpu_evals_per_s, 2)
    return out


if __name__ == "__main__":
    for w in (1, 2, 4, 8):
        print(w, predict(10000, w, 118.0))
        if w > 1:
            print(w, "what-if", {k: v for k, v in predict(10000, w, 118.0, group_local=True).items() if k in ("evals_per_s", "vs_one_gpu", "blocks_bound_by_the_chain")})
