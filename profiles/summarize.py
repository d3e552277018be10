#!/usr/bin/env python3
"""Turns the rocprofv3 (rocpd sqlite) outputs of profiles/run_profile.sh into a text summary and a machine-readable
counter file (the latter is what bench.py reads its per-launch instruction counts and HBM traffic from).

    python profiles/summarize.py gpurun_out/prof_<tag> [counters.json] > profiles/<tag>_summary.txt

Counter values are per launch: the per-dispatch sum over all counter instances, averaged over the FULL-SIZE dispatches of a
kernel (duration within a factor 2 of the median of that kernel's non-tiny dispatches), so the small launches the cascade makes with the same
kernels (leader and candidate re-scores) do not dilute the figures.
"""
import glob
import json
import os
import sqlite3
import sys

MAIN = ("eh_upper", "eh_score", "t_score", "t_prep", "knn2", "lm_kernel", "argmax", "triangulate", "index_sets")


def full_size(vals):
    """(value, duration) pairs of the full-size dispatches: within a factor 2 of the median duration of the dispatches that
    are not tiny (the first, cold launch of a kernel can take three times as long and must not set the scale)."""
    dmax = max(d for _, d in vals)
    big = sorted(d for _, d in vals if d >= 0.1 * dmax)
    ref = big[len(big) // 2]
    return [(v, d) for v, d in vals if 0.5 * ref <= d <= 2.0 * ref]


def clusters(vals):
    """[average duration (us), average value, launches] of every group of dispatches with similar durations (sorted by
    duration, a new group where the duration grows by more than 1.5x): one kernel serves several workloads of one run,
    and a reader that timed one of them picks the group that matches its own time."""
    out, cur = [], []
    for v, dur in sorted(vals, key=lambda x: x[1]):
        if cur and dur > 1.5 * cur[0][1]:
            out.append(cur)
            cur = []
        cur.append((v, dur))
    out.append(cur)
    return [[sum(d_ for _, d_ in g) / len(g) / 1e3, sum(v for v, _ in g) / len(g), len(g)] for g in out]


def main(d, json_out=None):
    tr = glob.glob(os.path.join(d, "trace", "*.db"))
    counters = {}
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
                if any(k in name for k in MAIN[:6]):
                    per.setdefault(name[:80], []).append((en - st) / 1e3)
            for name, durs in per.items():
                big = [x for _, x in full_size([(0, x) for x in durs])]
                print(f"{name:80s} | n={len(durs)} | full-size launches: " + " ".join(f"{x:.0f}" for x in big[:24]))
                tail = big[-20:]
                print(f"{'':80s} |   last {len(tail)} full-size launches (bench times these one at a time with events; earlier ones may "
                      f"overlap with another stream's launch): avg {sum(tail) / len(tail):.1f} us")
                counters.setdefault(name, {})["trace_avg_us_last_full_size"] = sum(tail) / len(tail)
            # The exact E/H scorer as ONE CALL: bench.py times sfm_eh_score_mixed with events, one call after the other on one
            # stream; in the trace those are the launches of eh_score_mfma2_kernel that no other launch of a scoring kernel
            # overlaps (the timed steps keep two calls in flight on two streams, which stretches both).
            big = [(name, st, en) for name, st, en in rows if any(k in name for k in ("eh_score_mfma", "eh_upper_mfma", "eh_score_kernel", "eh_upper_kernel"))]
            alone = {}
            for i, (name, st, en) in enumerate(big):
                if "eh_score_mfma2" not in name:
                    continue
                j = i - 1
                hit = False
                while j >= 0 and big[j][1] > st - 50_000_000:  # dispatches are ordered by start: look back 50 ms
                    if big[j][2] > st:
                        hit = True
                        break
                    j -= 1
                if not hit and i + 1 < len(big) and big[i + 1][1] < en:
                    hit = True
                if not hit:
                    alone.setdefault(name[:76], []).append((en - st) / 1e3)  # up to the last template argument (early loops or not)
            if alone:
                print("\n== launches of the exact E/H scorer that run alone on the device (us): instantiation | launches by duration group: count x average ==")
                call = {False: 0.0, True: 0.0}  # per form: without / with the early loops (last template argument)
                for name, durs in alone.items():
                    groups = clusters([(0.0, x * 1e3) for x in durs])
                    txt = "  ".join(f"{g[2]} x {g[0]:.1f}" for g in groups)
                    print(f"{name:76s} | {txt}")
                    # the groups of the default workload: the most frequent long group (main launch) and, for the E launch, the most frequent short one (its last round)
                    main = max((g for g in groups if g[0] > 1000.0), key=lambda g: g[2], default=None)
                    tail = max((g for g in groups if g[0] < 400.0), key=lambda g: g[2], default=None)
                    early = ", true>" in name
                    if main:
                        call[early] += main[0]
                    if tail and "<true, false" in name:
                        call[early] += tail[0]
                if call[False]:
                    print(f"one call on the default workload WITHOUT early loops (what `value` times) = E main launch + E last round + H launch: {call[False]:.1f} us "
                          "(bench.py roofline.kernel_ms also holds the operand preparation, ~10 us)")
                    counters["__exact_call_alone_us"] = {"sum_of_launches_us": call[False]}
                if call[True]:
                    print(f"one call WITH early loops (exact_counts_early_out.kernel_ms) = E main launch + E last round + H launch: {call[True]:.1f} us")
                    counters["__exact_early_call_alone_us"] = {"sum_of_launches_us": call[True]}
        except Exception as e:  # schema differences between rocprofv3 versions
            print("  (per-dispatch listing unavailable:", e, ")")
    for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(sub):
            continue
        dbs = glob.glob(os.path.join(sub, "*.db"))
        if not dbs:
            continue
        c = sqlite3.connect(dbs[0])
        print(f"\n== {os.path.basename(sub)} (rocprofv3 --pmc): per-kernel average over FULL-SIZE dispatches of the per-dispatch counter sum ==")
        rows = c.execute(
            "select name, counter_name, dispatch_id, sum(counter_value) v, max(duration) dur from pmc_events group by name, counter_name, dispatch_id"
        ).fetchall()
        groups = {}
        for name, cn, _, v, dur in rows:
            groups.setdefault((name, cn), []).append((v, dur))
        for (name, cn), vals in sorted(groups.items()):
            if not any(k in name for k in MAIN):
                continue
            big = full_size(vals)
            v = sum(x for x, _ in big) / len(big)
            dur = sum(x for _, x in big) / len(big)
            print(f"{name[:70]:70s} | {cn:26s} | {v:18.1f} | full-size dispatches {len(big)} of {len(vals)} | avg dur {dur / 1e3:10.1f} us")
            counters.setdefault(name[:80], {})[cn] = v
            counters[name[:80]][cn + "__dur_us"] = dur / 1e3
            # the full-size dispatches plus the small ones, per full-size dispatch: a call that cuts its last round off as a
            # second, small launch of the same kernel is then priced whole (longer dispatches belong to other workloads)
            small = [x for x, d_ in vals if d_ < 0.5 * min(d for _, d in big)]
            counters[name[:80]][cn + "__per_call"] = (sum(x for x, _ in big) + sum(small)) / len(big)
            counters[name[:80]].setdefault("__clusters", {})[cn] = clusters(vals)
    pt = glob.glob(os.path.join(d, "probe_trace", "*.db"))
    if pt:
        c = sqlite3.connect(pt[0])
        print("\n== kernel trace of profiles/tools/traffic_probe.py (each scoring call of the default workloads alone, a device synchronisation after "
              "every call: durations without another stream's launch beside them): name | launches | avg us of the launches after the first ==")
        try:
            per = {}
            for name, st, en in c.execute("select name, start, end from kernels order by start"):
                if any(k in name for k in ("eh_score_mfma", "eh_upper_mfma", "t_score_mfma", "corr_to_mfma", "corr_to_upper2", "t_prep_mfma")):
                    per.setdefault(name[:100], []).append((en - st) / 1e3)
            total = {}
            for name, durs in per.items():
                # one kernel name may stand for the main launch and the cut-off last round of a call: group by duration
                groups = clusters([(0.0, x * 1e3) for x in durs])
                for avg_us, _, cnt in groups:
                    print(f"{name:100s} | {cnt:4d} | {avg_us:10.1f}")
                key = None
                if "eh_score_mfma2" in name:
                    key = ("exact E/H call with early loops" if ", true>(" in name else "exact E/H call without early loops") + " (eh_score_mfma2: E main + E last round + H)"
                if key:
                    total[key] = total.get(key, 0.0) + sum(g[0] for g in groups)
            for k, v in total.items():
                print(f"sum of the launches of one {k}: {v:.1f} us")
        except Exception as e:
            print("  (unavailable:", e, ")")
    if json_out:
        with open(json_out, "w") as f:
            json.dump(counters, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
