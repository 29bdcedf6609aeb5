#!/usr/bin/env python
"""Kernel sequence of the LAST step in a rocprofv3 --kernel-trace database (run-length encoded): python tools/trace_seq.py <db> [substring-to-mark]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
names = [r[0].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60] for r in rows]
# the last occurrence of the patch-embedding marker starts the last step
mark = sys.argv[2] if len(sys.argv) > 2 else "patchify"
starts = [i for i, n in enumerate(names) if mark in n]
lo = starts[-1] if starts else 0
out, prev, cnt = [], None, 0
for i in range(lo, len(names)):
    n = names[i]
    if n == prev: cnt += 1
    else:
        if prev is not None: out.append(f"{cnt:3d} x {prev}")
        prev, cnt = n, 1
out.append(f"{cnt:3d} x {prev}")
print("\n".join(out))
