#!/usr/bin/env python
"""HBM traffic per launch of the GEMM kernels (per instantiation) and of the clustering call, from two rocprofv3 PMC passes
(FETCH_SIZE, WRITE_SIZE: separate passes, kernel-trace only) of the same command:

    python tools/gemm_traffic.py <fetch.db> <write.db> > profiles/rNN_traffic.json

Units and the gfx950 correction follow MI355X_MICROARCH.md's HBM / rocprofv3 section: both counters are in KiB; FETCH_SIZE
under-reports 16-byte-per-lane streams by 2x on gfx950.  The factor is CALIBRATED in the same run on a kernel whose traffic is known
exactly and whose access width is the same (16 B per lane): a row-wise pass that reads and writes the same number of bytes
(`layernorm_rows_kernel`, else `gather_rows_kernel`), so WRITE_SIZE / FETCH_SIZE of that kernel IS the factor.  The script fails loudly
when a kernel it needs is not in the databases — a renamed kernel must never turn into an empty artefact."""
import json
import re
import sqlite3
import sys

KIB = 1024.0
CLUSTER_KERNELS = ("dpc_",)                    # every kernel of setok_cluster_dpc_knn (csrc/cluster.hip) carries this prefix
CALIBRATION_KERNELS = ("layernorm_rows_kernel", "gather_rows_kernel")
ACT = {"0": "plain", "1": "quick_gelu", "2": "gelu_erf"}


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    agg = {}
    for name, cn, v in c.execute("select name, counter_name, counter_value from pmc_events"):
        if cn != counter:
            continue
        name = str(name).replace("(anonymous namespace)::", "")
        agg.setdefault(name, []).append(float(v))
    if not agg:
        sys.exit(f"{db}: no {counter} samples")
    return agg


def pick(d, sub):
    return {k: vs for k, vs in d.items() if sub in k}


def need(d, sub, what):
    got = pick(d, sub)
    if not got:
        sys.exit(f"gemm_traffic.py: no kernel matching '{sub}' in the {what} pass; kernels seen: {sorted(d)[:40]}")
    return got


GEMM_KERNELS = ("gemm_pp_kernel", "gemm_persist_kernel", "gemm_tail_kernel")   # csrc/gemm_persist.hip: ping-pong main kernel, its predecessor
                                                                                # (ragged / device-side row counts / edge tiles), small-tile tail


def _flag(x):
    return x in ("true", "1")


def gemm_class(name):
    """(kernel family, epilogue class) of one instantiation.  Template orders: gemm_pp_kernel<ACT, LNK, RESK>,
    gemm_persist_kernel<ACT, F32B, RESK, LNK>, gemm_tail_kernel<ACT, LNK, TTM, TTN, NS>."""
    m = re.search(r"(gemm_pp_kernel|gemm_persist_kernel|gemm_tail_kernel)<([^>]*)>", name)
    if not m:
        return "other", name
    fam, a = m.group(1), [x.strip() for x in m.group(2).split(",")]
    a += ["false"] * (4 - len(a))
    if fam == "gemm_pp_kernel":
        act, lnk, resk = a[0], _flag(a[1]), _flag(a[2])
    elif fam == "gemm_persist_kernel":
        if _flag(a[1]):
            return fam, "fp32_batched"
        act, resk, lnk = a[0], _flag(a[2]), _flag(a[3])
    else:
        act, lnk, resk = a[0], _flag(a[1]), False
    return fam, ACT.get(act, act) + ("+residual" if resk else "") + ("+layernorm" if lnk else "")


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    # ---- calibration ----------------------------------------------------------------------------------------------------------------
    cal = None
    for name in CALIBRATION_KERNELS:
        f, w = pick(fetch, name), pick(write, name)
        if f and w:
            lf = max(v for vs in f.values() for v in vs) * KIB           # the largest launches (the ViT-sized ones): rows * C * 2 bytes each way
            lw = max(v for vs in w.values() for v in vs) * KIB
            cal = (name, lf, lw)
            break
    if cal is None:
        sys.exit(f"gemm_traffic.py: none of the calibration kernels {CALIBRATION_KERNELS} is in both passes; kernels seen: {sorted(fetch)[:40]}")
    corr = float(round(cal[2] / cal[1])) if cal[1] > 0 else 1.0
    if corr not in (1.0, 2.0):
        sys.exit(f"gemm_traffic.py: implausible FETCH_SIZE factor {cal[2] / cal[1]:.3f} from {cal[0]}")

    # ---- GEMM, per instantiation --------------------------------------------------------------------------------------------------------
    gf, gw = {}, {}
    for fam in GEMM_KERNELS:
        gf.update(pick(fetch, fam + "<")); gw.update(pick(write, fam + "<"))
    if not pick(gf, "gemm_pp_kernel") and not pick(gf, "gemm_persist_kernel"):
        sys.exit(f"gemm_traffic.py: neither gemm_pp_kernel nor gemm_persist_kernel in the FETCH_SIZE pass; kernels seen: {sorted(fetch)[:40]}")
    per_class, per_kernel_out, tot_f, tot_w, tot_n = {}, {}, 0.0, 0.0, 0
    acc = {}
    for name in sorted(gf):
        if name not in gw:
            sys.exit(f"gemm_traffic.py: {name} is in the FETCH_SIZE pass but not in the WRITE_SIZE pass")
        f, w = gf[name], gw[name]
        if len(f) != len(w):
            sys.exit(f"gemm_traffic.py: {name}: {len(f)} launches in the FETCH_SIZE pass, {len(w)} in the WRITE_SIZE pass (not the same command?)")
        fam, cls = gemm_class(name)
        rf, rw = sum(f) / len(f) * KIB, sum(w) / len(w) * KIB
        per_kernel_out[f"{fam}:{cls}"] = {"launches": len(f), "fetch_size_raw_bytes": int(rf), "write_size_raw_bytes": int(rw),
                                          "traffic_bytes_per_launch": int(rf * corr + rw)}
        if fam != "gemm_tail_kernel":                    # the tail kernel covers the M / N remainders of a launch of the same class: its bytes are added
            a = acc.setdefault(cls, [0, 0.0, 0.0]); a[0] += len(f)      # to that class, its launches are not
        else:
            a = acc.setdefault(cls, [0, 0.0, 0.0])
        a[1] += sum(f) * KIB; a[2] += sum(w) * KIB
        tot_f += sum(f) * KIB; tot_w += sum(w) * KIB
        tot_n += len(f) if fam != "gemm_tail_kernel" else 0
    for cls, (n, sf, sw) in sorted(acc.items()):
        if n == 0:                                       # a class only the small-tile kernel runs (latency shapes): count its own launches
            n = sum(v["launches"] for k, v in per_kernel_out.items() if k == f"gemm_tail_kernel:{cls}")
            tot_n += n
        per_class[cls] = {"launches": n, "fetch_size_raw_bytes": int(sf / n), "write_size_raw_bytes": int(sw / n),
                          "traffic_bytes_per_launch": int((sf * corr + sw) / n)}

    # ---- clustering: all kernels of one setok_cluster_dpc_knn call ------------------------------------------------------------------------
    cl, calls = {}, None
    for sub in CLUSTER_KERNELS:
        cf, cw = need(fetch, sub, "FETCH_SIZE"), need(write, sub, "WRITE_SIZE")
        for name in sorted(cf):
            f, w = cf[name], cw.get(name)
            if w is None or len(w) != len(f):
                sys.exit(f"gemm_traffic.py: {name}: launch counts differ between the passes")
            short = re.sub(r"\(.*", "", name).replace("void ", "")
            cl[short] = {"launches": len(f), "traffic_bytes_per_launch": int(sum(f) / len(f) * KIB * corr + sum(w) / len(w) * KIB)}
            calls = len(f) if calls is None else min(calls, len(f))
    cl_total = sum(v["traffic_bytes_per_launch"] * v["launches"] for v in cl.values()) / max(calls or 1, 1)

    print(json.dumps({
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 --timed-only",
        "fetch_correction": corr,
        "calibration": f"{cal[0]}, same run: FETCH_SIZE {cal[1] / 1e6:.1f} MB vs WRITE_SIZE {cal[2] / 1e6:.1f} MB for a stream that reads and writes the "
                       f"same number of bytes (16 B per lane) -> x{corr:g}",
        "gemm": {"kernel": "gemm_pp_kernel<*> + gemm_persist_kernel<*> (+ gemm_tail_kernel<*> remainders, attributed to the launch they complete)",
                 "launches": tot_n, "traffic_bytes_per_launch": int((tot_f * corr + tot_w) / tot_n),
                 "per_class": per_class, "per_kernel": per_kernel_out},
        "clustering": {"kernels": cl, "calls": calls, "traffic_bytes_per_launch": int(cl_total),
                       "note": "bytes of ALL kernels of one setok_cluster_dpc_knn call (256 images)"},
    }, indent=1))


if __name__ == "__main__":
    main()
