#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database: python tools/rocpd_pmc.py <db> [substr]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name, cn, v, dur in c.execute("select name, counter_name, counter_value, duration from pmc_events"):
    if sub in str(name):
        agg[str(name)[:70]][cn].append(v)
        agg[str(name)[:70]]["~duration_ns"].append(dur)
for kn, d in agg.items():
    print(kn)
    for cn, vs in sorted(d.items()):
        print(f"   {cn:34s} n={len(vs):4d} avg={sum(vs) / len(vs):.6g}")
