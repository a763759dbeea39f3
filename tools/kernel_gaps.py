#!/usr/bin/env python3
"""Where a solve's time goes BETWEEN kernels: reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv), orders the dispatches
by start time and reports, per kernel name, launches / busy time / idle time on the device before it starts (gap to the end of
the previous kernel), over the last `--last-ms` milliseconds of the trace (the timed solve).  usage: kernel_gaps.py trace.csv [--last-ms 130]"""
import csv, sys, collections, re
path = sys.argv[1]
last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1]) if "--last-ms" in sys.argv else None
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if last_ms is not None:
    t_end = rows[-1][1]
    rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
short = lambda s: re.sub(r"\(.*", "", re.sub(r"^void ", "", s)).replace("khip::", "").replace("(anonymous namespace)::", "")[:60]
busy, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
biggest = []
prev_end = rows[0][0]
for s, e, name in rows:
    k = short(name)
    cnt[k] += 1; busy[k] += e - s
    g = max(0, s - prev_end)
    gap[k] += g
    biggest.append((g, k))
    prev_end = max(prev_end, e)
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} dispatches over {span / 1e6:.3f} ms; busy {sum(busy.values()) / 1e6:.3f} ms, idle {sum(gap.values()) / 1e6:.3f} ms ({100 * sum(gap.values()) / span:.1f} %)")
print(f"{'kernel':60s} {'n':>5s} {'busy ms':>9s} {'avg us':>9s} {'idle-before ms':>15s} {'avg gap us':>11s}")
for k in sorted(cnt, key=lambda k: -(busy[k] + gap[k])):
    print(f"{k:60s} {cnt[k]:5d} {busy[k] / 1e6:9.3f} {busy[k] / cnt[k] / 1e3:9.1f} {gap[k] / 1e6:15.3f} {gap[k] / cnt[k] / 1e3:11.1f}")
print("largest gaps (us, before kernel):", [(round(g / 1e3, 1), k) for g, k in sorted(biggest, reverse=True)[:12]])
