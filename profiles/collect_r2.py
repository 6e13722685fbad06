#!/usr/bin/env python3
"""Turn the output of profiles/collect_r2.sh (gpurun_out/r2_final/) into the committed round-2 summaries:

  profiles/r2_bench_*.json              the bench lines (default 30 steps, the driver's 20/5, host loop, C3 tree / brute, C5 8M / 2M, identity 8M)
  profiles/r2_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary of `python3 bench.py --no-cpu-baseline`
  profiles/r2_identity_kernel_stats.csv the same for the 8M-point identity pass
  profiles/r2_pmc.json                  per-regime PMC figures of the default bench command, read by bench.py:
                                          whole_pass_hbm_bytes, and per regime (first pass / passes 2-3 / converged passes)
                                          HBM bytes and vector instructions per pass

PMC method (MI355X_MICROARCH.md, 'HBM' and 'rocprofv3 PMC slots'): FETCH_SIZE and WRITE_SIZE in two separate runs (they do not
fit one pass), both in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so fetch bytes =
FETCH_SIZE * 1024 * 2 -- calibrated here on k_pass_identity<4> (algorithmic read 8M x 48 B = 384.0 MB) and applied to the other
kernels, whose narrower (4-byte-per-lane planar) and gathered reads are uncalibrated: their figure may be overstated, by at most
2x.  Write bytes = WRITE_SIZE * 1024.  SQ_INSTS_VALU counts wave64 vector instructions.  A dispatch is attributed to a pass by its
place in the stream: a pass ends with k_final_reduce or k_reduce_solve<true>; a k_search_packet launch starts an alignment.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r2_final")
OUT = os.path.join(ROOT, "profiles")


def one(pattern):
    files = glob.glob(pattern)
    if len(files) != 1:
        sys.exit("expected one file for %s, got %s" % (pattern, files))
    return files[0]


def short(name):
    n = name.split("(")[0].replace("symmicp::", "").replace("void ", "")
    return n.split("<")[0]


def dispatches(dirname, counters):
    """[(dispatch id, kernel, {counter: value})] in stream order"""
    rows = defaultdict(dict)
    names = {}
    with open(one(os.path.join(SRC, dirname, "*", "*_counter_collection.csv"))) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] in counters:
                d = int(r["Dispatch_Id"])
                rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                names[d] = short(r["Kernel_Name"])
    return [(d, names[d], rows[d]) for d in sorted(rows)]


PASS_KERNELS = ("k_search_packet", "k_search_cells", "k_search_walk", "k_accumulate", "k_pass_fused", "k_pass_identity", "k_nn_brute", "k_pass_indexed")
PASS_END = ("k_final_reduce", "k_reduce_solve")


def per_pass(dirname, counter, scale):
    """{pass index within its alignment: [value per alignment]} summed over the pass's kernels, and per-kernel means"""
    out, kern = defaultdict(list), defaultdict(list)
    idx, acc, open_pass, started = 0, 0.0, False, False
    for _, k, v in dispatches(dirname, (counter,)):
        val = v.get(counter, 0.0) * scale
        if k == "k_search_packet":
            idx, acc, open_pass, started = 0, 0.0, False, True
        if not started:
            continue
        if k in PASS_KERNELS:
            acc += val
            open_pass = True
            kern[k].append(val)
        elif k in PASS_END and open_pass:
            kern[k].append(val)
            out[idx].append(acc)
            idx, acc, open_pass = idx + 1, 0.0, False
    return out, kern


def mean(v):
    return sum(v) / len(v) if v else None


def regimes(dirname, counter, scale):
    pp, kern = per_pass(dirname, counter, scale)
    conv = [x for i, vals in pp.items() if i >= 4 for x in vals]
    return dict(first_pass=mean(pp.get(0, [])), search_pass=mean(pp.get(1, []) + pp.get(2, [])), converged=mean(conv),
                all_passes=mean([x for vals in pp.values() for x in vals])), {k: mean(v) for k, v in kern.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    pmc_only = "--pmc-only" in sys.argv          # on the GPU box, between the PMC passes and the bench lines (collect_r2.sh)
    for name in () if pmc_only else ("bench_default", "bench_driver_20", "bench_host_loop", "bench_c3_100k_tree", "bench_c3_100k_brute", "bench_c5_8M_tree", "bench_c5_2M_tree", "bench_identity_8M"):
        line = [l for l in open(os.path.join(SRC, name + ".json")).read().splitlines() if l.startswith("{")][-1]
        json.loads(line)
        with open(os.path.join(OUT, "r2_" + name + ".json"), "w") as f:
            f.write(line + "\n")
    if not pmc_only:
        shutil.copy(one(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv")), os.path.join(OUT, "r2_kernel_stats.csv"))
        shutil.copy(one(os.path.join(SRC, "trace_identity", "*", "*_kernel_stats.csv")), os.path.join(OUT, "r2_identity_kernel_stats.csv"))
    for tag, log in () if pmc_only else (("", "trace.log"), ("identity_", "trace_identity.log")):
        line = [l for l in open(os.path.join(SRC, log)).read().splitlines() if l.startswith("{") and '"metric"' in l][-1]
        with open(os.path.join(OUT, "r2_%sbench_under_rocprof.json" % tag), "w") as f:
            f.write(line + "\n")

    fetch, kf = regimes("pmc_fetch", "FETCH_SIZE", 1024.0 * 2.0)
    write, kw = regimes("pmc_write", "WRITE_SIZE", 1024.0)
    valu, kv = regimes("pmc_sq", "SQ_INSTS_VALU", 1.0)
    kernels = {}
    for r in ("first_pass", "search_pass", "converged"):
        kernels[r] = dict(hbm_bytes=round(fetch[r] + write[r]) if fetch[r] is not None and write[r] is not None else None,
                          fetch_bytes=round(fetch[r]) if fetch[r] is not None else None, write_bytes=round(write[r]) if write[r] is not None else None,
                          valu_wave_insts=round(valu[r]) if valu[r] is not None else None)
    by_kernel = {k: dict(fetch_bytes=round(kf.get(k) or 0), write_bytes=round(kw.get(k) or 0), valu_wave_insts=round(kv.get(k) or 0)) for k in sorted(set(kf) | set(kw) | set(kv))}
    # calibration of the x2 on the identity stream
    idf, _ = per_pass("pmc_id_fetch", "FETCH_SIZE", 1024.0 * 2.0)
    idw, _ = per_pass("pmc_id_write", "WRITE_SIZE", 1024.0)
    # (identity runs have no k_search_packet: attribute by kernel name instead)
    idk_f = defaultdict(list)
    for _, k, v in dispatches("pmc_id_fetch", ("FETCH_SIZE",)):
        idk_f[k].append(v["FETCH_SIZE"] * 2048.0)
    idk_w = defaultdict(list)
    for _, k, v in dispatches("pmc_id_write", ("WRITE_SIZE",)):
        idk_w[k].append(v["WRITE_SIZE"] * 1024.0)
    ident = dict(k_pass_identity=dict(fetch_bytes=round(mean(idk_f["k_pass_identity"])), write_bytes=round(mean(idk_w["k_pass_identity"])),
                                      algorithmic_read_bytes=8_000_000 * 48, algorithmic_write_bytes=8_000_000 * 24))
    out = dict(
        method="rocprofv3 --kernel-trace --pmc, three separate runs of `python3 bench.py --no-cpu-baseline --warmup 0` (FETCH_SIZE; WRITE_SIZE; SQ_*): "
               "fetch = FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of wide reads, calibrated on k_pass_identity<4>: see calibration), write = WRITE_SIZE KiB x 1024, "
               "vector instructions = SQ_INSTS_VALU (wave64 instructions); per-pass sums over the pass's kernels, means over the alignments of the process "
               "(timed run through the device loop + the instrumented host-loop run); generated by profiles/collect_r2.py from profiles/collect_r2.sh",
        calibration=ident,
        workloads={"c4:1000000:paper:tree": dict(whole_pass_hbm_bytes=round(fetch["all_passes"] + write["all_passes"]), kernels=kernels, by_kernel=by_kernel)},
    )
    with open(os.path.join(OUT, "r2_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
