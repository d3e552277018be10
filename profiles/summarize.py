#!/usr/bin/env python3
"""Turns the rocprofv3 (rocpd sqlite) outputs of profiles/run_profile.sh into a text summary.

    python profiles/summarize.py gpurun_out/prof_<tag> > profiles/<tag>_summary.txt
"""
import glob
import os
import sqlite3
import sys


def main(d):
    tr = glob.glob(os.path.join(d, "trace", "*.db"))
    if tr:
        c = sqlite3.connect(tr[0])
        print("== kernel trace (rocprofv3 --kernel-trace --stats): name | calls | total us | avg us | % ==")
        for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print(f"{name[:110]:110s} | {calls:5d} | {tot:12.1f} | {avg:10.2f} | {pct:6.2f}")
        print("\n== per-dispatch durations of the main kernels in dispatch order (us) ==")
        try:
            rows = c.execute("select name, start, end from kernels order by start").fetchall()
            per = {}
            for name, st, en in rows:
                if "eh_upper" in name or ("eh_score" in name) or "t_score" in name or "knn2_kernel" in name or "eh_solve" in name:
                    per.setdefault(name[:80], []).append((en - st) / 1e3)
            for name, durs in per.items():
                big = [x for x in durs if x > 0.25 * max(durs)]
                print(f"{name:80s} | n={len(durs)} | full-size launches: " + " ".join(f"{x:.0f}" for x in big[:24]))
                tail = big[-20:]
                print(f"{'':80s} |   last {len(tail)} full-size launches (bench times these one at a time with events; earlier ones may "
                      f"overlap with another stream's launch): avg {sum(tail) / len(tail):.1f} us")
        except Exception as e:  # schema differences between rocprofv3 versions
            print("  (per-dispatch listing unavailable:", e, ")")
    for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(sub):
            continue
        dbs = glob.glob(os.path.join(sub, "*.db"))
        if not dbs:
            continue
        c = sqlite3.connect(dbs[0])
        print(f"\n== {os.path.basename(sub)} (rocprofv3 --pmc): per-kernel average over dispatches of the per-dispatch counter sum ==")
        rows = c.execute(
            "select name, counter_name, avg(v), count(*), avg(dur) from (select name, counter_name, dispatch_id, sum(counter_value) v, "
            "max(duration) dur from pmc_events group by name, counter_name, dispatch_id) group by name, counter_name"
        ).fetchall()
        for name, cn, v, n, dur in rows:
            if "eh_upper" in name or "eh_score" in name or "eh_solve" in name or "t_score" in name or "argmax" in name or "lm_kernel" in name or "knn2" in name:
                print(f"{name[:70]:70s} | {cn:22s} | {v:18.1f} | dispatches {n} | avg dur {dur/1e3:10.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
