#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (rocpd sqlite, what `rocprofv3 --kernel-trace
--stats` writes on ROCm 7.2) into the plain-text per-kernel summary that is
committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_x.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    # median per kernel (the average of a short trace carries its cold first launches)
    med = {}
    for name, dur in cur.execute("select name, duration from kernels").fetchall():
        med.setdefault(name, []).append(dur)
    med = {k: sorted(v)[len(v) // 2] for k, v in med.items()}
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print("# durations in microseconds")
    print(f"{'calls':>6} {'total_us':>14} {'avg_us':>12} {'med_us':>12} {'min_us':>12} {'max_us':>12} {'pct':>6} "
          f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>7} {'grid_x':>9} {'wg_x':>5}  kernel")
    for r in rows:
        name, calls, tot, avg, mn, mx, vg, ag, sg, lds, scr, gx, wx = r
        print(f"{calls:6d} {tot/1e3:14.3f} {avg/1e3:12.3f} {med[name]/1e3:12.3f} {mn/1e3:12.3f} {mx/1e3:12.3f} {100*tot/total:6.2f} "
              f"{vg or 0:5d} {ag or 0:5d} {sg or 0:5d} {lds or 0:7d} {scr or 0:7d} {gx or 0:9d} {wx or 0:5d}  {name}")


if __name__ == "__main__":
    main(sys.argv[1])
