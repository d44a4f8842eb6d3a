#!/usr/bin/env python3
"""The launches of one rocprofv3 --kernel-trace [--memory-copy-trace] run in time order, condensed: usage timeline.py <dir> [min_ms]
(prints launches longer than min_ms, per-kernel totals, the H2D copies' spacing and the GPU's idle time inside the span)."""
import csv, glob, os, re, sys
d = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
kt = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
mt = sorted(glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True))
rows = []
for path, kind in ((kt[0] if kt else "", "k"), (mt[0] if mt else "", "c")):
    if not path:
        continue
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Name") or r.get("Direction") or "?"
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        mm = re.search(r"(k[0-9zb]_?[a-z0-9_]*|scan_[a-z0-9_]*|finalize_kernel|__amd_rocclr_[a-zA-Z]*|MEMORY_COPY_[A-Z_]*)", name)
        rows.append((s, e, (mm.group(1) if mm else name)[:28], r.get("Queue_Id", r.get("Stream_Id", "")), kind))
rows.sort()
t0 = rows[0][0]
busy_end, idle = t0, 0
tot = {}
shown = 0
for s, e, n, q, kind in rows:
    if e - s > min_ms * 1e6 and not n.startswith("MEMORY") and shown < 400:
        print("%9.3f ms  +%8.3f ms  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n))
        shown += 1
    if s > busy_end:
        idle += s - busy_end
    busy_end = max(busy_end, e)
    a = tot.setdefault(n, [0, 0, 0])
    a[0] += 1; a[1] += e - s; a[2] = max(a[2], e - s)
print("per kernel: calls, total ms, average ms, longest ms")
for n, (c, t, m) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-28s %6d %10.3f %9.3f %9.3f" % (n, c, t / 1e6, t / 1e6 / c, m / 1e6))
# busy time of the inflate kernel and of everything but copies (union of intervals)
def union(sel):
    iv = sorted((s, e) for s, e, n, q, kind in rows if sel(n, kind))
    u, end = 0, 0
    for s, e in iv:
        if e > end:
            u += e - max(s, end)
            end = e
    return u / 1e6
print("kernels busy (union) %.3f ms, inflate busy (union) %.3f ms, H2D copies busy (union) %.3f ms" %
      (union(lambda n, k: k == "k"), union(lambda n, k: n.startswith("kz")), union(lambda n, k: n.startswith("MEMORY_COPY_HOST_TO_DEVICE"))))
cd = sorted((e - s) / 1e6 for s, e, n, q, kind in rows if n.startswith("MEMORY_COPY_HOST_TO_DEVICE") and e - s > 100000)
if cd:
    print("H2D piece copies: %d, duration min %.3f / median %.3f / max %.3f ms, %.1f ms in all" % (len(cd), cd[0], cd[len(cd) // 2], cd[-1], sum(cd)))
print("span %.3f ms, GPU idle inside it %.3f ms, launches %d" % ((busy_end - t0) / 1e6, idle / 1e6, len(rows)))
