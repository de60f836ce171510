#!/usr/bin/env python3
"""Replay the traced shared phases (tools/wave_trace.py) under other wave-level schedules (CPU only).

A lane's task is a string of bodies: I (task / march init), F (filtered turn), S (estimate set-up, s = without
slab test), T (walk trip = one dependent table fetch), H (hit evaluation / end of an estimated turn), A (AO task
set-up incl. estimate set-up).  The product runs a phase in rounds of 64 tasks and every round in lock step
(nested loops: the wave-level count of a body is the sum over turns of the max over lanes).  A flat schedule
lets every lane sit at its own body and executes ONE body per step for the lanes that wait at it.
Reports wave-level executions per body and the VALU they stand for.
"""
import argparse
import collections
import sys

import numpy as np

COST = {"I": 150, "F": 18, "S": 260, "s": 180, "T": 62, "H": 40, "A": 330, "a": 250}


def parse(path):
    """-> phases [(ctx, [task string])], walks {ctx: [[per-lane list of skip lengths per trip] per wave-level walk]}"""
    z = np.load(path)
    hdr, ev = z["hdr"], z["ev"]
    R, NB = 32, ev.shape[2]
    phases = []
    rounds = []  # (ctx, [per-lane list of walks; a walk = list of trip bytes])
    trunc = 0
    for w in range(len(hdr) // R):
        cur = None
        prev_rem = None
        for r in range(R):
            h = int(hdr[w * R + r])
            if not h:
                break
            ctx, rem = (h >> 16) & 0x7fff, h & 0xffff
            if cur is None or cur[0] != ctx or rem >= prev_rem:
                cur = (ctx, [])
                phases.append(cur)
            prev_rem = rem
            rnd = []
            for lane in range(64):
                b = [int(x) for x in ev[w * R + r, lane]]
                if b[0] != 0xFE:
                    continue
                s, i, ok = [], 1, False
                walks = []

                def walk(i):
                    trips = []
                    if i < NB and b[i] == 0xFD:
                        i += 1
                        while i < NB:
                            trips.append(b[i])
                            i += 1
                            if trips[-1] >= 251:
                                break
                    return i, trips
                try:
                    if ctx == 2:
                        i, trips = walk(i)
                        fl, tr = b[i], b[i + 1]
                        s.append("A" if fl & 4 else "a")
                        s.extend("T" * tr)
                        if fl & 1:
                            s.append("H")
                            walks.append(trips)
                        ok = b[i + 2] == 0xFF
                    else:
                        s.append("I")
                        while i < NB:
                            if b[i] == 0xFF:
                                ok = True
                                break
                            nf, why = b[i], b[i + 1]
                            i += 2
                            s.extend("F" * nf)
                            if why != 1:
                                continue
                            i, trips = walk(i)
                            fl, tr, conv = b[i], b[i + 1], b[i + 2]
                            i += 3
                            s.append("S" if fl & 4 else "s")
                            s.extend("T" * tr)
                            s.append("H")
                            walks.append(trips if fl & 1 else None)
                except IndexError:
                    ok = False
                trunc += 0 if ok else 1
                cur[1].append("".join(s))
                rnd.append(walks)
            rounds.append((ctx, rnd))
    return phases, trunc, rounds


def lookahead(rounds, nw):
    """wave-level trips of the traced walks if one trip looked at K consecutive samples (their table fetches in
    flight together): a lane goes through its recorded trips in order; a trip whose skip is one sample lets the
    same wave-level trip take the lane's next recorded trip as well (up to K)."""
    def merged(trips, K):
        n, i = 0, 0
        while i < len(trips):
            n += 1
            k = 1
            while k < K and i < len(trips) and trips[i] == 1:
                i += 1
                k += 1
            i += 1
        return n
    for ctx, name in ((2, "AO"), (3, "shadow")):
        for K in (1, 2, 3, 4):
            tot = 0
            lane_tot = 0
            for c, rnd in rounds:
                if c != ctx:
                    continue
                depth = max((len(w) for w in rnd), default=0)
                for k in range(depth):  # the k-th estimated turn of the round's lanes runs in lock step
                    col = [w[k] for w in rnd if k < len(w) and w[k] is not None]
                    if col:
                        tot += max(merged(t, K) for t in col)
                        lane_tot += sum(merged(t, K) for t in col)
            print(f"   {name:6s} lookahead {K}: wave-level trips {tot / nw:6.2f} per wavefront, lane-level {lane_tot / nw / 64:6.2f} per sample")


def lockstep(tasks):
    """the product: rounds of 64, nested loops.  Returns Counter of wave-level body executions."""
    c = collections.Counter()
    for base in range(0, len(tasks), 64):
        rnd = tasks[base:base + 64]
        # split every task into turns: [I][F* (S T* H)?]*
        seqs = []
        for t in rnd:
            turns, i = [], 0
            head = ""
            if t and t[0] in "IAa":
                head = t[0]
                i = 1
            cur = {"F": 0, "S": "", "T": 0, "H": 0}
            out = []
            while i < len(t):
                ch = t[i]
                if ch == "F":
                    if cur["S"] or cur["T"] or cur["H"]:
                        out.append(cur)
                        cur = {"F": 0, "S": "", "T": 0, "H": 0}
                    cur["F"] += 1
                elif ch in "Ss":
                    if cur["S"]:
                        out.append(cur)
                        cur = {"F": 0, "S": "", "T": 0, "H": 0}
                    cur["S"] = ch
                elif ch == "T":
                    cur["T"] += 1
                elif ch == "H":
                    cur["H"] = 1
                    out.append(cur)
                    cur = {"F": 0, "S": "", "T": 0, "H": 0}
                i += 1
            if cur["F"] or cur["S"] or cur["T"]:
                out.append(cur)
            seqs.append((head, out))
        heads = set(h for h, _ in seqs if h)
        for h in heads:
            c["A" if h in "Aa" and "A" in heads else h] += 1 if h != "a" or "A" not in heads else 0
        k = 0
        while True:
            turn = [o[k] for _, o in seqs if k < len(o)]
            if not turn:
                break
            c["F"] += max(t["F"] for t in turn)
            ss = set(t["S"] for t in turn if t["S"])
            if ss:
                c["S" if "S" in ss else "s"] += 1
            c["T"] += max(t["T"] for t in turn)
            c["H"] += 1 if any(t["H"] for t in turn) else 0
            k += 1
    return c


def flat(tasks, policy, deal=True, thr=16):
    """one body per step for the lanes waiting at it; deal: a lane that ends takes the next task of the phase."""
    c = collections.Counter()
    nxt = 0
    lanes = [None] * 64  # [task string, position]
    def take(l):
        nonlocal nxt
        if nxt < len(tasks):
            lanes[l] = [tasks[nxt], 0]
            nxt += 1
        else:
            lanes[l] = None
    for l in range(64):
        take(l)
    if not deal:
        raise NotImplementedError
    while True:
        at = collections.Counter()
        for st in lanes:
            if st is not None:
                ch = st[0][st[1]] if st[1] < len(st[0]) else None
                if ch is None:
                    continue
                at[ch.upper() if ch in "sa" else ch] += 1
        if not at:
            break
        pick = policy(at, thr)
        # execute: the lanes at that body advance one body (F: the whole run of F, cost = the longest run)
        longest = 0
        sub = set()
        for l, st in enumerate(lanes):
            if st is None or st[1] >= len(st[0]):
                continue
            ch = st[0][st[1]]
            key = ch.upper() if ch in "sa" else ch
            if key != pick:
                continue
            sub.add(ch)
            if ch == "F":
                n = 0
                while st[1] < len(st[0]) and st[0][st[1]] == "F":
                    st[1] += 1
                    n += 1
                longest = max(longest, n)
            else:
                st[1] += 1
            if st[1] >= len(st[0]):
                take(l)
                if lanes[l] is not None and not lanes[l][0]:
                    take(l)
        if pick == "F":
            c["F"] += longest
        elif pick in "SA":
            c[pick if pick in sub else pick.lower()] += 1
        else:
            c[pick] += 1
        c["steps"] += 1
    return c


def pol_majority(at, thr):
    return max(at, key=lambda k: (at[k], k == "T"))


def pol_walk_first(at, thr):
    """walk trips whenever at least thr lanes wait for one (or nothing else can run); otherwise the fullest body"""
    if at.get("T", 0) >= thr:
        return "T"
    others = {k: v for k, v in at.items() if k != "T"}
    if not others:
        return "T"
    k = max(others, key=lambda k: others[k])
    if at.get("T", 0) > others[k]:
        return "T"
    return k


def pol_cheap_first(at, thr):
    """cheap bodies (F, H, T) run as soon as anyone waits; expensive ones (I, S, A) only when >= thr lanes wait or nothing else is left"""
    for k in ("S", "A", "I"):
        if at.get(k, 0) >= thr:
            return k
    for k in ("T", "H", "F"):
        if at.get(k, 0):
            # prefer the fuller of T / F
            cands = [x for x in ("T", "F", "H") if at.get(x, 0)]
            return max(cands, key=lambda x: at[x])
    return max(at, key=lambda k: at[k])


def valu(c):
    return sum(COST.get(k, 0) * v for k, v in c.items())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--thr", type=int, nargs="*", default=[8, 16, 24, 32])
    a = ap.parse_args()
    phases, trunc, rounds = parse(a.trace)
    nw = 300
    print(f"{len(phases)} phases, {sum(len(t) for _, t in phases)} tasks, {trunc} truncated strings")
    print("\n== several samples per wave-level trip (lock-step rounds as in the product)")
    lookahead(rounds, nw)
    for ctx, name in ((2, "AO"), (3, "shadow")):
        ph = [t for c, t in phases if c == ctx]
        lane = collections.Counter()
        for t in ph:
            for s in t:
                lane.update(s)
        ntask = sum(len(t) for t in ph)
        print(f"\n== {name}: {len(ph)} phases, {ntask} tasks; lane-level bodies per task: " +
              " ".join(f"{k}={v / ntask:.2f}" for k, v in sorted(lane.items())))
        ideal = {k: v / 64 for k, v in lane.items()}
        print(f"   100 % packing: VALU {valu(ideal) / nw:8.0f} per wavefront, trips {ideal.get('T', 0) / nw:6.2f}")
        tot = collections.Counter()
        for t in ph:
            tot.update(lockstep(t))
        print(f"   product (rounds, lock step): VALU {valu(tot) / nw:8.0f}  " + " ".join(f"{k}={v / nw:.2f}" for k, v in sorted(tot.items())))
        for pname, pol in (("majority", pol_majority), ("walk-first", pol_walk_first), ("cheap-first", pol_cheap_first)):
            for thr in (a.thr if pname != "majority" else [0]):
                tot = collections.Counter()
                for t in ph:
                    tot.update(flat(t, pol, True, thr))
                print(f"   flat {pname:11s} thr {thr:2d}: VALU {valu(tot) / nw:8.0f}  " + " ".join(f"{k}={v / nw:.2f}" for k, v in sorted(tot.items())))


if __name__ == "__main__":
    main()
