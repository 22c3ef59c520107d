"""Idle time between kernels of a rocprofv3 --kernel-trace CSV (development aid): busy time per kernel, the union of
the busy intervals (kernels of different streams overlap) and the largest idle gaps, over the last `frac` of the trace.
Usage: python tools/kernel_gaps.py <*_kernel_trace.csv> [frac=0.5] [dump_n=0]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
dump = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
short = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("frdev::", "")
busy = collections.Counter(); cnt = collections.Counter(); gaps = collections.Counter()
t0 = int(rows[0]["Start_Timestamp"]); cover = 0; cur_end = t0; prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = short(r); busy[n] += e - s; cnt[n] += 1
    if s > cur_end:
        gaps[(prev or "") + " -> " + n] += s - cur_end
        cover += e - s
    else:
        cover += max(0, e - cur_end)
    if e > cur_end: cur_end, prev = e, n
span = cur_end - t0
print("span_ms %.3f union_busy_ms %.3f (%.1f%%) sum_busy_ms %.3f" % (span / 1e6, cover / 1e6, 100.0 * cover / span, sum(busy.values()) / 1e6))
for k, v in busy.most_common(8): print("busy %-36s n=%-4d total %.3f ms  avg %.4f ms" % (k, cnt[k], v / 1e6, v / 1e6 / cnt[k]))
for k, v in gaps.most_common(8): print("gap  %-70s %.3f ms" % (k, v / 1e6))
for r in rows[:dump]:
    print("%10.3f %10.3f  q=%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), short(r)))
