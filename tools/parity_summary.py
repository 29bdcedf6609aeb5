#!/usr/bin/env python
"""gpurun_out/<tag>/parity_raw.txt (written by tests/parity.py under SETOK_PARITY_LOG while `pytest -m gpu` runs) -> a table per test:
   python tools/parity_summary.py gpurun_out/r06/parity_raw.txt > profiles/r06_parity.txt
One line per (test, compared tensor): the worst max-norm, rms and floor-guarded element-wise relative error over the calls, and the tolerance."""
import collections, sys
rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    parts = line.rstrip("\n").split("\t")
    if len(parts) != 6:
        continue
    test, what, a, b, c, tol = parts
    vals = [float(x.split()[1]) for x in (a, b, c)] + [float(tol.split()[1])]
    key = (test.split("::", 1)[-1], what)
    old = rows.get(key)
    rows[key] = [max(x, y) for x, y in zip(old[:3], vals[:3])] + [vals[3], old[4] + 1] if old else vals + [1]
print("# fp32-parity assertions of `pytest -m gpu` (tests/parity.py): worst value over the calls of each assertion; bounds: max_rel < tol, rms_rel < tol, elem_rel < 4 tol")
print(f"# {'test :: compared tensors':118s} {'calls':>5s} {'max_rel':>10s} {'rms_rel':>10s} {'elem_rel':>10s} {'tol':>8s}")
cur = None
for (test, what), (mx, rms, el, tol, n) in rows.items():
    if test != cur:
        print(test)
        cur = test
    print(f"    {what:116s} {n:5d} {mx:10.3e} {rms:10.3e} {el:10.3e} {tol:8.1e}")
allv = list(rows.values())
if allv:
    print(f"# worst over everything: max_rel {max(v[0] / v[3] for v in allv):.3f} of tol, rms_rel {max(v[1] / v[3] for v in allv):.3f} of tol, elem_rel {max(v[2] / v[3] for v in allv):.3f} of tol")
