#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace: per-kernel calls / total / avg / share.
usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import re
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by {name_col}").fetchall()
    tot = sum(r[2] for r in rows) or 1
    rows.sort(key=lambda r: -r[2])
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,share"]
    for n, cnt, t, mn, mx in rows:
        short = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))[:90]
        lines.append(f"\"{short}\",{cnt},{t / 1e6:.3f},{t / cnt / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{t / tot:.4f}")
    lines.append(f"\"TOTAL\",{sum(r[1] for r in rows)},{tot / 1e6:.3f},,,,1.0")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:3])
