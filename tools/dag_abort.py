#!/usr/bin/env python3
"""Offline look at the state a timed-out DAG launch left behind (COCONS_DEBUG_ABORT=1 COCONS_DEBUG_ABORT_DUMP=path writes
path.N, see info_status in csrc/api.hip; with cocons_debug_tune("dag_trace", 1) the dump holds the stamps of every task).

    python tools/dag_abort.py gpurun_out/abort.bin.0
"""
import sys

import numpy as np


def decode(steps, L):
    """task index -> (step, kind, detail)"""
    s = int(np.searchsorted(steps[:, 0].view(np.uint32) if steps.dtype != np.uint32 else steps[:, 0], L, side="right")) - 1
    base, near, tpos, nT = [int(v) for v in steps[s].view(np.uint32)[:4]]
    H, W, tj0, k0, K, nstrip, two, need, nd_next, split = [int(v) for v in steps[s][4:14]]
    p2, p3 = [int(v) for v in steps[s].view(np.uint32)[14:16]]
    q = L - base
    per = 2 * nstrip
    nA, perBC = per + nd_next, (per if two else 0)
    kind, u, qt = None, 0, q                      # the list of a step: tiles | T1, early halves | tiles | T2 | tiles | T3 | tiles
    if q >= tpos:
        if q < tpos + nA:
            u = q - tpos
            kind = "T1" if u < per else "early"
        elif q < p2:
            qt = q - nA
        elif q < p2 + perBC:
            kind, u = "T2", q - p2
        elif q < p3:
            qt = q - nA - perBC
        elif q < p3 + perBC:
            kind, u = "T3", q - p3
        else:
            qt = q - nA - 2 * perBC
    if kind == "early":
        return s, "early", "dd %d" % (u - per)
    if kind:
        return s, kind, "strip %d h %d (row64 %d)" % (u // 2, u % 2, tj0 + (4 if two else 2) + u // 2)
    jl = 0
    while (jl + 1) * H - (jl + 1) * jl // 2 <= qt and jl + 1 < W:
        jl += 1
    tj = tj0 + jl
    ti = tj + (qt - (jl * H - jl * (jl - 1) // 2))
    t = tj0 // 2
    Ti, Tj = ti // 2 - t, tj // 2 - t
    kind = "diag" if (0 <= Ti <= 1 and 0 <= Tj <= Ti) else ("near" if q < near else "far")
    return s, kind, "tile (%d, %d)" % (ti, tj)


def main():
    raw = open(sys.argv[1], "rb").read()
    hdr = np.frombuffer(raw, np.uint32, 16, 0)
    assert hdr[0] == 0xDA6D0001, "not a dump"
    nt, nsteps, ntasks, nwords, fcap, ntr, code, qn, now = [int(v) for v in hdr[1:10]]
    off = 64
    rec = np.frombuffer(raw, np.uint32, 7, off); off += 28
    steps = np.frombuffer(raw, np.int32, nsteps * 16, off).reshape(nsteps, 16); off += nsteps * 64
    words = np.frombuffer(raw, np.uint32, nwords, off); off += 4 * nwords
    flags = np.frombuffer(raw, np.uint32, 4 * fcap + 64, off); off += 4 * (4 * fcap + 64)
    print("abort code 0x%x; nt %d, %d steps, %d tasks, counter %d" % (code, nt, nsteps, ntasks, qn))
    s, kind, det = decode(steps, int(rec[0]))
    print("first to give up: task %d = step %d %s %s; code 0x%x, word %d, needed %d, saw %d, holds %d now; %.1f ms, %d polls"
          % (rec[0], s, kind, det, rec[1], rec[2], rec[3], rec[4], now, rec[5] * 1e-5, rec[6]))
    mt = None
    for m in range(nt, nt + 3):          # the words' layout depends on mt (nt or nt + 1)
        T64 = 2 * m
        if 64 + T64 * (T64 + 1) // 2 + (nsteps + 2) * T64 + nsteps + 64 + 16 * (nsteps + 2) <= nwords:
            mt = m
    T64 = 2 * mt
    tdone = words[64: 64 + T64 * (T64 + 1) // 2]
    pdone = words[64 + len(tdone): 64 + len(tdone) + (nsteps + 2) * T64].reshape(nsteps + 2, T64)
    pall = words[64 + len(tdone) + pdone.size:][: nsteps + 64]
    dcount = words[64 + len(tdone) + pdone.size + nsteps + 64:][: 16 * (nsteps + 2)].reshape(-1, 16)
    w = int(rec[2])
    if 64 <= w < 64 + len(tdone):
        i = w - 64
        ti = int((np.sqrt(8 * i + 1) - 1) // 2)
        print("   the word is tdone of tile (%d, %d)" % (ti, i - ti * (ti + 1) // 2))
    elif w < 64 + len(tdone) + pdone.size:
        i = w - 64 - len(tdone)
        print("   the word is pdone of panel %d strip %d" % (i // T64, i % T64))
    inn, out, xr = flags[:fcap], flags[fcap: 2 * fcap], flags[2 * fcap: 3 * fcap]
    print("engine words (first %d tiles): in  %s" % (2 * nsteps + 4, inn[: 2 * nsteps + 4]))
    print("                               out %s" % out[: 2 * nsteps + 4])
    print("                               xr  %s" % xr[: 2 * nsteps + 4])
    print("alive %d" % flags[3 * fcap])
    print("pall %s" % pall[: nsteps + 2])
    for p in range(1, nsteps + 1):
        H = int(steps[p - 1][4]) - 4 if p >= 1 else 0
        row = pdone[p][: max(0, int(steps[p - 1][9]))]
        if row.size and row.min() < 6:
            print("panel %d: strips below 6: %s" % (p, {int(i): int(v) for i, v in enumerate(row) if v < 6}))
            break
    if not ntr:
        return
    st = np.frombuffer(raw, np.uint64, ntasks * 4, off).reshape(ntasks, 4); off += 32 * ntasks
    eng = np.frombuffer(raw, np.uint64, 8 * (nt + 2), off).reshape(-1, 8)
    t0 = st[:, 0][st[:, 0] > 0].min()
    us = lambda v: (float(v) - float(t0)) * 0.01
    drawn = st[:, 0] > 0
    unfin = drawn & (st[:, 3] == 0)
    print("%d tasks drawn, %d unfinished; last stamp at %.1f us" % (drawn.sum(), unfin.sum(), us(st.max())))
    summ = {}
    for L in np.nonzero(unfin)[0]:
        s, kind, det = decode(steps, int(L))
        state = "waiting-inputs" if st[L, 1] == 0 else ("in-product" if st[L, 2] == 0 else "before-store")
        summ.setdefault((s, kind, state), []).append((int(L), det, us(st[L, 0])))
    for key in sorted(summ):
        v = summ[key]
        print("  step %2d %-5s %-14s x %4d   e.g. task %d %s drawn %.1f us" % (key[0], key[1], key[2], len(v), v[0][0], v[0][1], v[0][2]))
    hw = None
    if len(raw) >= off + 8 * eng.size + 8 * ntasks:
        hw = np.frombuffer(raw, np.uint32, 2 * ntasks, off + 8 * eng.size).reshape(ntasks, 2)

    def where(w):
        w = int(w)      # XCC in bits 28..31; HW_ID: wave 0..3, simd 4..5, pipe 6..7, cu 8..11, sh 12, se 13..15
        return "xcc %d se %d sh %d cu %2d simd %d wave %d" % (w >> 28, (w >> 13) & 7, (w >> 12) & 1, (w >> 8) & 15, (w >> 4) & 3, w & 15)
    if hw is not None and hw.any():
        e0, e1, ech = int(eng[0][0]), int(eng[0][1]), int(eng[0][2])
        print("engine ran first on %s; last seen on %s (changed at pair %d)" % (where(e0), where(e1), ech))
        fin = (st[:, 3] > 0) & (hw[:, 0] != 0)
        moved = fin & ((hw[:, 0] >> 8) != (hw[:, 1] >> 8))          # CU / SE / XCC changed between draw and store
        print("finished tasks whose workgroup changed CU between draw and store: %d of %d" % (moved.sum(), fin.sum()))
        for L in np.nonzero(moved)[0][:12]:
            print("   task %6d %s: %s -> %s   drawn %.1f stored %.1f" % (L, decode(steps, int(L))[1:], where(hw[L, 0]), where(hw[L, 1]),
                                                                    us(st[L, 0]), us(st[L, 3])))
        span = (st[:, 3].astype(np.float64) - st[:, 1].astype(np.float64)) * 0.01
        frozen = np.nonzero((st[:, 3] > 0) & (span > 50000))[0]
        for L in frozen:
            print("FROZEN task %6d %s: drawn on %s, stored on %s; inputs %.1f product %.1f stored %.1f"
                  % (L, decode(steps, int(L))[1:], where(hw[L, 0]), where(hw[L, 1]), us(st[L, 1]), us(st[L, 2]), us(st[L, 3])))
            # who else was on that CU at the time
            same = np.nonzero((hw[:, 0] >> 8 == hw[L, 0] >> 8) & (st[:, 0] > 0) & (st[:, 0] <= st[L, 3]) &
                              ((st[:, 3] == 0) | (st[:, 3] >= st[L, 1])))[0]
            for M in same[:12]:
                print("      same CU: task %6d %s drawn %.1f stored %s" % (M, decode(steps, int(M))[1:], us(st[M, 0]),
                                                                         "%.1f" % us(st[M, 3]) if st[M, 3] else "-"))
    print("engine stamps per tile pair (us): 1 = first tile factored, 2 = out, 4 = xr, 6 = second tile factored, 7 = out")
    for p in range(min(nsteps + 2, eng.shape[0])):
        print("  pair %2d: %s" % (p, " ".join("%9.1f" % us(v) if v else "        -" for v in eng[p])))


if __name__ == "__main__":
    main()


def tile_history(path, ti, tj):
    """every task of tile (ti, tj) with its stamps, and the tile's tdone word"""
    raw = open(path, "rb").read()
    hdr = np.frombuffer(raw, np.uint32, 16, 0)
    nt, nsteps, ntasks, nwords, fcap, ntr = [int(v) for v in hdr[1:7]]
    off = 64 + 28
    steps = np.frombuffer(raw, np.int32, nsteps * 16, off).reshape(nsteps, 16); off += nsteps * 64
    words = np.frombuffer(raw, np.uint32, nwords, off); off += 4 * nwords + 4 * (4 * fcap + 64)
    st = np.frombuffer(raw, np.uint64, ntasks * 4, off).reshape(ntasks, 4)
    t0 = st[:, 0][st[:, 0] > 0].min()
    print("tdone(%d, %d) = %d" % (ti, tj, words[64 + ti * (ti + 1) // 2 + tj]))
    for s in range(nsteps):
        base, near, tpos, nT = [int(v) for v in steps[s].view(np.uint32)[:4]]
        H, W, tj0 = [int(v) for v in steps[s][4:7]]
        nstrip, two, nd_next = int(steps[s][9]), int(steps[s][10]), int(steps[s][12])
        p2, p3 = [int(v) for v in steps[s].view(np.uint32)[14:16]]
        jl = tj - tj0
        if jl < 0 or jl >= W or ti < tj:
            continue
        qt = jl * H - jl * (jl - 1) // 2 + (ti - tj)
        nA, perBC = 2 * nstrip + nd_next, (2 * nstrip if two else 0)
        q = qt                                   # position of update tile qt in the step's list (see decode)
        if q >= tpos:
            q += nA
            if q >= p2:
                q += perBC
                if q >= p3:
                    q += perBC
        L = base + q
        print("  step %2d task %6d (%s): %s" % (s, L, decode(steps, L)[1:], " ".join("%9.1f" % ((float(v) - float(t0)) * 0.01) if v else "        -" for v in st[L])))


if len(sys.argv) > 3:
    tile_history(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
