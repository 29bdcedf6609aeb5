#!/usr/bin/env python
"""Where the wall time of the LAST step in a rocprofv3 --kernel-trace database goes: kernel time, idle gaps between consecutive dispatches
(by the pair of kernels either side), and what stands next to every blit copy:  python tools/step_timeline.py <db> [step-marker-substring]"""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
short = lambda n: n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:48]
rows = [(short(n), s, e) for n, s, e in rows]
mark = sys.argv[2] if len(sys.argv) > 2 else "patchify"
starts = [i for i, r in enumerate(rows) if mark in r[0]]
lo = starts[-1] if starts else 0
step = rows[lo:]
wall = (max(e for _, _, e in step) - step[0][1]) / 1e3
busy = sum(e - s for _, s, e in step) / 1e3
print(f"last step: {len(step)} dispatches, wall {wall:.1f} us, sum of kernel durations {busy:.1f} us, difference {wall - busy:.1f} us")
gaps = collections.defaultdict(lambda: [0, 0.0])
end = step[0][2]
for i in range(1, len(step)):
    n, s, e = step[i]
    g = (s - end) / 1e3
    k = (step[i - 1][0], n)
    gaps[k][0] += 1; gaps[k][1] += g
    end = max(end, e)
print("gap (us) between consecutive dispatches, by pair, largest total first (negative = overlap):")
for k, (cnt, tot) in sorted(gaps.items(), key=lambda kv: -abs(kv[1][1]))[:25]:
    print(f"  {tot:9.1f} us  {cnt:4d} x {tot / cnt:6.2f}   {k[0]}  ->  {k[1]}")
cp = [i for i, r in enumerate(step) if "copyBuffer" in r[0] or "fillBuffer" in r[0]]
print(f"{len(cp)} blit dispatches in the step; their neighbours:")
nb = collections.Counter((step[i - 1][0] if i else "-", step[i][0], step[i + 1][0] if i + 1 < len(step) else "-") for i in cp)
for k, v in nb.most_common(20):
    print(f"  {v:4d} x  {k[0]}  |  {k[1]}  |  {k[2]}")
# the head of the step (everything behind the tower): one line per dispatch, start relative to the first, duration, gap in front
hs = [i for i, r in enumerate(step) if "select_add_pos" in r[0]]
if hs:
    print("dispatches behind the tower (start us, duration us, idle in front us, kernel):")
    t0 = step[hs[-1]][1]; prev_end = step[hs[-1] - 1][2] if hs[-1] else t0
    for n, s_, e_ in step[hs[-1]:]:
        print(f"  {(s_ - t0) / 1e3:9.1f} {(e_ - s_) / 1e3:8.1f} {(s_ - prev_end) / 1e3:7.1f}  {n}")
        prev_end = max(prev_end, e_)
