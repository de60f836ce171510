#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc result databases: per kernel name, the average
counter value per dispatch.  usage: pmc_summary.py <db> [<db> ...] [--kernel substr]"""
import sqlite3
import sys


def summarise(path, kernel_filter=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    rows = cur.execute("select * from counters_collection").fetchall()
    ci = {c: i for i, c in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ci else "name"
    agg = {}
    for r in rows:
        k = r[ci[name_col]]
        if kernel_filter and kernel_filter not in k:
            continue
        key = (k, r[ci["counter_name"]])
        a = agg.setdefault(key, [0, 0.0, set()])
        a[1] += r[ci["value"]]
        a[2].add(r[ci["dispatch_id"]])
    out = {}
    for (k, cname), (_c, tot, disp) in agg.items():
        out.setdefault(k, {})[cname] = (tot / max(len(disp), 1), len(disp))
    return out


def main():
    args = sys.argv[1:]
    filt = None
    if "--kernel" in args:
        i = args.index("--kernel")
        filt = args[i + 1]
        args = args[:i] + args[i + 2:]
    for p in args:
        print(f"# {p}")
        for k, cs in summarise(p, filt).items():
            print(f"  kernel: {k[:110]}")
            for cname, (avg, nd) in sorted(cs.items()):
                print(f"    {cname:32s} avg/dispatch {avg:20.1f}   (dispatches {nd})")


if __name__ == "__main__":
    main()
