#!/usr/bin/env python3
"""Turn the output of profiles/collect_r3.sh (gpurun_out/r3_final/) into the committed round-3 summaries:

  profiles/r3_bench_*.json               the bench lines (default 30 steps, the driver's 20/5, host loop, C3 tree / brute, C5 8M / 2M, identity 8M)
  profiles/r3_kernel_stats.csv           rocprofv3 --kernel-trace --stats summary of `python3 bench.py --no-cpu-baseline`            (C4, 1M)
  profiles/r3_c5_kernel_stats.csv        the same for `bench.py --workload c5 --points 8000000 --steps 50`                             (C5, 8M)
  profiles/r3_identity_kernel_stats.csv  the same for the 8M-point identity pass
  profiles/r3_normals_kernel_stats.csv   the same for scratch/time_normals.py (k_normals_knn at 100k / 1M / 8M points) + r3_normals.log (wall times)
  profiles/r3_predict_ranks_{2,4,8}.log  per-rank kernel time of the N-way split 1M pair, measured on one GPU (scratch/predict_ranks.py)
  profiles/r3_pmc.json                   per-workload (C4 1M, C5 8M), per-regime and per-kernel PMC figures, read by bench.py

PMC method (MI355X_MICROARCH.md, 'HBM' and 'rocprofv3 PMC slots'): FETCH_SIZE and WRITE_SIZE in two separate runs (they do not
fit one pass), both in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so fetch bytes =
FETCH_SIZE * 1024 * 2 -- calibrated here on k_pass_identity<4> (algorithmic read 8M x 48 B = 384.0 MB) and applied to the other
kernels, whose narrower (4-byte-per-lane planar) and gathered reads are uncalibrated: their figure may be overstated, by at most
2x.  Write bytes = WRITE_SIZE * 1024.  SQ_INSTS_VALU counts wave64 vector instructions; SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count
quad-cycles summed over the waves (a vector instruction in flight / a wave resident).  A dispatch is attributed to a pass by its
place in the stream: a pass ends with k_final_reduce or k_reduce_solve<true>; a k_search_packet launch starts an alignment.
Kernel names keep their template arguments (k_pass_fused<true> is the accumulating pass of a converged alignment, <false> the
search-only compacting kernel: different byte counts).
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r3_final")
OUT = os.path.join(ROOT, "profiles")
SQ = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVES")


def one(pattern):
    files = glob.glob(pattern)
    if len(files) != 1:
        sys.exit("expected one file for %s, got %s" % (pattern, files))
    return files[0]


def short(name, keep_template=True):
    n = name.replace("symmicp::", "").replace("void ", "")
    m = re.match(r"\s*([A-Za-z_0-9]+)(<[^(]*>)?\(", n)
    if not m:
        return n.split("(")[0]
    return m.group(1) + ((m.group(2) or "").replace(" ", "") if keep_template else "")


def dispatches(dirname, counters):
    """[(dispatch id, kernel (with template arguments), {counter: value})] in stream order"""
    rows = defaultdict(dict)
    names = {}
    with open(one(os.path.join(SRC, dirname, "*", "*_counter_collection.csv"))) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] in counters:
                d = int(r["Dispatch_Id"])
                rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                names[d] = short(r["Kernel_Name"])
    return [(d, names[d], rows[d]) for d in sorted(rows)]


def base(k):
    return k.split("<")[0]


PASS_KERNELS = ("k_search_packet", "k_search_cells", "k_search_walk", "k_accumulate", "k_accumulate_list", "k_pass_fused", "k_pass_identity", "k_nn_brute", "k_pass_indexed")
PASS_END = ("k_final_reduce", "k_reduce_solve")


def per_pass(dirname, counter, scale, start_kernel="k_search_packet"):
    """{pass index within its alignment: [value per alignment]} summed over the pass's kernels, and per-kernel lists"""
    out, kern = defaultdict(list), defaultdict(list)
    idx, acc, open_pass, started = 0, 0.0, False, start_kernel is None
    for _, k, v in dispatches(dirname, (counter,)):
        val = v.get(counter, 0.0) * scale
        if start_kernel and base(k) == start_kernel:
            idx, acc, open_pass, started = 0, 0.0, False, True
        if not started:
            continue
        if base(k) in PASS_KERNELS:
            acc += val
            open_pass = True
            kern[k].append(val)
        elif base(k) in PASS_END and open_pass:
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


def workload(tag):
    fetch, kf = regimes("pmc_%s_fetch" % tag, "FETCH_SIZE", 1024.0 * 2.0)
    write, kw = regimes("pmc_%s_write" % tag, "WRITE_SIZE", 1024.0)
    sq = {c: regimes("pmc_%s_sq" % tag, c, 1.0) for c in SQ}
    valu = sq["SQ_INSTS_VALU"][0]
    kernels = {}
    for r in ("first_pass", "search_pass", "converged"):
        kernels[r] = dict(hbm_bytes=round(fetch[r] + write[r]) if fetch[r] is not None and write[r] is not None else None,
                          fetch_bytes=round(fetch[r]) if fetch[r] is not None else None, write_bytes=round(write[r]) if write[r] is not None else None,
                          valu_wave_insts=round(valu[r]) if valu[r] is not None else None,
                          valu_active_quad_cycles=round(sq["SQ_ACTIVE_INST_VALU"][0][r]) if sq["SQ_ACTIVE_INST_VALU"][0][r] is not None else None,
                          wave_quad_cycles=round(sq["SQ_WAVE_CYCLES"][0][r]) if sq["SQ_WAVE_CYCLES"][0][r] is not None else None)
    names = sorted(set(kf) | set(kw) | set(sq["SQ_INSTS_VALU"][1]))
    by_kernel = {}
    for k in names:
        e = dict(fetch_bytes=round(kf.get(k) or 0), write_bytes=round(kw.get(k) or 0))
        for c in SQ:
            v = sq[c][1].get(k)
            if v is not None:
                e[{"SQ_INSTS_VALU": "valu_wave_insts", "SQ_INSTS_SALU": "salu_wave_insts", "SQ_WAVE_CYCLES": "wave_quad_cycles", "SQ_ACTIVE_INST_VALU": "valu_active_quad_cycles",
                   "SQ_WAIT_ANY": "wait_any_quad_cycles", "SQ_WAIT_INST_ANY": "wait_inst_quad_cycles", "SQ_BUSY_CYCLES": "sq_busy_cycles", "SQ_WAVES": "waves"}[c]] = round(v)
        by_kernel[k] = e
    return dict(whole_pass_hbm_bytes=round(fetch["all_passes"] + write["all_passes"]), kernels=kernels, by_kernel=by_kernel)


def main():
    os.makedirs(OUT, exist_ok=True)
    pmc_only = "--pmc-only" in sys.argv          # on the GPU box, between the PMC passes and the bench lines (collect_r3.sh)
    for name in () if pmc_only else ("bench_default", "bench_driver_20", "bench_host_loop", "bench_c3_100k_tree", "bench_c3_100k_brute", "bench_c5_8M_tree", "bench_c5_2M_tree", "bench_identity_8M"):
        line = [l for l in open(os.path.join(SRC, name + ".json")).read().splitlines() if l.startswith("{")][-1]
        json.loads(line)
        with open(os.path.join(OUT, "r3_" + name + ".json"), "w") as f:
            f.write(line + "\n")
    if not pmc_only:
        for d, o in (("trace", "r3_kernel_stats.csv"), ("trace_c5", "r3_c5_kernel_stats.csv"), ("trace_identity", "r3_identity_kernel_stats.csv"), ("trace_normals", "r3_normals_kernel_stats.csv")):
            shutil.copy(one(os.path.join(SRC, d, "*", "*_kernel_stats.csv")), os.path.join(OUT, o))
        with open(os.path.join(OUT, "r3_normals.log"), "w") as f:
            f.write("".join(l for l in open(os.path.join(SRC, "trace_normals.log")) if "points:" in l))
        for n in (2, 4, 8):
            with open(os.path.join(OUT, "r3_predict_ranks_%d.log" % n), "w") as f:
                f.write("".join(l for l in open(os.path.join(SRC, "predict_ranks_%d.log" % n)) if "kernel ms per pass" in l))
        for tag, log in (("", "trace.log"), ("c5_", "trace_c5.log"), ("identity_", "trace_identity.log")):
            line = [l for l in open(os.path.join(SRC, log)).read().splitlines() if l.startswith("{") and '"metric"' in l][-1]
            with open(os.path.join(OUT, "r3_%sbench_under_rocprof.json" % tag), "w") as f:
                f.write(line + "\n")

    # calibration of the x2 on the identity stream (identity runs have no k_search_packet: attribute by kernel name)
    idk_f, idk_w = defaultdict(list), defaultdict(list)
    for _, k, v in dispatches("pmc_id_fetch", ("FETCH_SIZE",)):
        idk_f[base(k)].append(v["FETCH_SIZE"] * 2048.0)
    for _, k, v in dispatches("pmc_id_write", ("WRITE_SIZE",)):
        idk_w[base(k)].append(v["WRITE_SIZE"] * 1024.0)
    ident = dict(k_pass_identity=dict(fetch_bytes=round(mean(idk_f["k_pass_identity"])), write_bytes=round(mean(idk_w["k_pass_identity"])),
                                      algorithmic_read_bytes=8_000_000 * 48, algorithmic_write_bytes=8_000_000 * 24))
    out = dict(
        method="rocprofv3 --kernel-trace --pmc, three separate runs per workload of the bench command with --no-cpu-baseline --warmup 0 (FETCH_SIZE; WRITE_SIZE; SQ_*): "
               "fetch = FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of wide reads, calibrated on k_pass_identity<4>: see calibration), write = WRITE_SIZE KiB x 1024, "
               "vector instructions = SQ_INSTS_VALU (wave64 instructions), *_quad_cycles = SQ counters in quad-cycles summed over waves; per-pass sums over the pass's "
               "kernels, means over the alignments of the process (timed run through the device loop + the instrumented host-loop run); generated by "
               "profiles/collect_r3.py from profiles/collect_r3.sh",
        calibration=ident,
        workloads={"c4:1000000:paper:tree": workload("c4"), "c5:8000000:paper:tree": workload("c5")},
    )
    with open(os.path.join(OUT, "r3_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1)[:4000])


if __name__ == "__main__":
    main()
