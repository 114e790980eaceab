#!/usr/bin/env python3
"""What runs on the GPU between the end of the cold fill of minibatch t and the start of the probe of t+1?  Reads a `rocprofv3 --kernel-trace`
of tools/fetch_gap_probe.py (kernel_trace.csv: name, queue, start / end timestamps of every launch) and, for the last `--last` fetches, lists per
hand-over K2(t) -> K1(t+1) and K1(t) -> K2(t): its length, and every kernel of ANOTHER queue that started, ended or ran during it.  Development tool.

  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/fetch_gap_probe.py ; python3 tools/gap_trace_analyze.py <dir>"""
import collections
import csv
import glob
import os
import statistics as st
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("<")[0].split("(")[0][-40:]


def main():
    d = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows]
    ev.sort()
    k1 = [e for e in ev if "probe_gather_kernel" in e[2]]
    k2 = [e for e in ev if "miss_fill_kernel" in e[2]]
    n = min(len(k1), len(k2))
    k1, k2 = k1[-min(n, last):], k2[-min(n, last):]
    fq = k1[-1][3]
    others = [e for e in ev if e[3] != fq]
    starts = [e[0] for e in others]
    import bisect

    def overlapping(a, b):
        i = bisect.bisect_left(starts, a - 3_000_000)
        out = []
        while i < len(others) and others[i][0] < b:
            if others[i][1] > a:
                out.append(others[i])
            i += 1
        return out
    for label, pairs in (("K2(t) -> K1(t+1)", [(k2[i][1], k1[i + 1][0]) for i in range(len(k1) - 1) if k1[i + 1][0] > k2[i][1]]),
                         ("K1(t) -> K2(t)", [(k1[i][1], k2[i][0]) for i in range(len(k1)) if k2[i][0] > k1[i][1]])):
        gaps = [(b - a) / 1e3 for a, b in pairs]
        busy = collections.Counter()
        with_other = 0
        cover = []
        for a, b in pairs:
            ov = overlapping(a, b)
            if ov:
                with_other += 1
            for o in ov:
                busy[o[2]] += 1
            cover.append(sum(min(o[1], b) - max(o[0], a) for o in ov) / max(b - a, 1))
        q = sorted(gaps)
        print(f"{label}: {len(gaps)} hand-overs on queue {fq}: median {st.median(gaps):.1f} us, mean {st.mean(gaps):.1f}, p10 {q[len(q) // 10]:.1f}, p90 {q[9 * len(q) // 10]:.1f}; "
              f"{100 * with_other / len(gaps):.0f} % of them overlap a kernel of another queue (mean coverage of the gap by such kernels {100 * st.mean(cover):.0f} %)")
        quiet = [g for g, c in zip(gaps, cover) if c == 0]
        noisy = [g for g, c in zip(gaps, cover) if c > 0]
        if quiet and noisy:
            print(f"    hand-overs with NO other kernel in them: {len(quiet)}, median {st.median(quiet):.1f} us;   with one: {len(noisy)}, median {st.median(noisy):.1f} us")
        print("    kernels of other queues seen in the gaps: " + ", ".join(f"{k} x{v}" for k, v in busy.most_common(8)))
    # the steady state's budget per fetch: the two kernels, the two hand-overs, the period -- and what ran beside the probe (it starts together with the consumer's step)
    d1 = [(e[1] - e[0]) / 1e3 for e in k1]
    d2 = [(e[1] - e[0]) / 1e3 for e in k2]
    period = [(k1[i + 1][0] - k1[i][0]) / 1e3 for i in range(len(k1) - 1)]
    beside = collections.Counter()
    cov1 = []
    for e in k1:
        ov = overlapping(e[0], e[1])
        for o in ov:
            beside[o[2]] += 1
        cov1.append(min(1.0, sum(min(o[1], e[1]) - max(o[0], e[0]) for o in ov) / max(e[1] - e[0], 1)))
    alone = [d for d, c in zip(d1, cov1) if c == 0]
    shared = [d for d, c in zip(d1, cov1) if c > 0]
    print(f"last {len(k1)} fetches: K1 median {st.median(d1):.1f} us (mean {st.mean(d1):.1f}); K2 median {st.median(d2):.1f} us (mean {st.mean(d2):.1f}); "
          f"K1 start -> next K1 start median {st.median(period):.1f} us (mean {st.mean(period):.1f})")
    print(f"    K1 launches with NO kernel of another queue beside them: {len(alone)}" + (f", median {st.median(alone):.1f} us" if alone else "") +
          f";   with one: {len(shared)}" + (f", median {st.median(shared):.1f} us" if shared else "") +
          ";   beside K1: " + ", ".join(f"{k} x{v}" for k, v in beside.most_common(6)))
    durs = collections.defaultdict(list)
    for e in ev:
        durs[(e[3], e[2])].append((e[1] - e[0]) / 1e3)
    print("kernels by queue (launches, mean us): " + "; ".join(f"q{q} {k} x{len(v)} {st.mean(v):.1f}" for (q, k), v in sorted(durs.items(), key=lambda kv: -len(kv[1]))[:14]))


if __name__ == "__main__":
    main()
