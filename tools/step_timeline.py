#!/usr/bin/env python3
"""What fills a pipelined step?  From a `rocprofv3 --kernel-trace --output-format csv` trace of
`python bench.py --steps N --no-e2e --repeats 0`: over the middle of the timed region (pipelined verify launches: the ones with
the smaller grids), the share of the time any kernel / a verify kernel is running, each kernel's summed duration, and the
step time the window implies.  usage: python tools/step_timeline.py <..._kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
grid = lambda r: int(r.get("Grid_Size", 0) or 0) or int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
v = [r for r in rows if "linesearch_verify_kernel" in r["Kernel_Name"]]
g = max(grid(r) for r in v)
pv = [r for r in v if grid(r) < g]
t0 = int(pv[len(pv) // 5]["Start_Timestamp"])
t1 = int(pv[len(pv) * 4 // 5]["End_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1]
busy = collections.Counter()
cnt = collections.Counter()
cur = vcur = t0
cover = vcover = 0
depth_time = collections.Counter()   # time with k kernels in flight
events = []
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("frdev::", "")
    busy[n] += e - s
    cnt[n] += 1
    cover += max(0, e - max(cur, s))
    cur = max(cur, e)
    if "verify" in n:
        vcover += max(0, e - max(vcur, s))
        vcur = max(vcur, e)
        events += [(s, 1), (e, -1)]
events.sort()
k, last = 0, t0
for t, d in events:
    depth_time[k] += t - last
    last, k = t, k + d
span = t1 - t0
nv = sum(1 for r in sel if "linesearch_verify_kernel" in r["Kernel_Name"])
print("window %.2f ms, %d verify launches (three per step): %.3f ms per step" % (span / 1e6, nv, span / 1e6 / (nv / 3.0)))
print("any kernel running %.1f %% of the window; a verify kernel running %.1f %%" % (100.0 * cover / span, 100.0 * vcover / span))
print("verify kernels in flight: " + ", ".join("%d: %.1f %%" % (d, 100.0 * t / span) for d, t in sorted(depth_time.items())))
for name, t in busy.most_common(8):
    print("  %-28s n=%-4d sum %7.2f ms  avg %.4f ms  (%.1f %% of the window if it ran alone)" % (name, cnt[name], t / 1e6, t / 1e6 / cnt[name], 100.0 * t / span))
