#!/usr/bin/env python3
"""bench.py -- headline benchmark of the symmetric-ICP hot path on MI355X.

Workload (BASELINE.json configs[3], "C4"): synthetic 1M-point surface cloud pair with analytic
normals, 30 ICP iterations, paper-correct symmetric objective, exact nearest-neighbour
correspondences, source sharded over N GPUs with one RCCL all-reduce of 40 doubles per pass.

A "step" is one ICP iteration = one trip of the reference loop body (ICP/myicp.cpp:123-142):
6x6 solve + compose, one GPU pass (transform + NN search + M/N/c rows + 37 fp64 sums), reduce,
[all-reduce].  The timed region is begin() (the initial correspondence pass, myicp.cpp:122) plus
exactly K steps, with inputs already resident in HBM; upload and index build are reported separately.

    python bench.py                     # N=1, K=30, W=5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Beyond the driver's contract it carries
  passes               first / second+third / converged pass durations (HIP events inside the timed region) and the cold
                       exact-NN rate of the first pass (mcorr_per_sec_first_pass)
  roofline             the whole pass against the HBM roof (SURVEY 8(d) algorithmic bytes / mean pass duration)
  roofline_by_regime   each regime's dominant kernel against the resource that bounds it: HBM for the streaming kernels,
                       vector-instruction issue for the searches and for brute force
  cpu_baseline         the oracle's like-for-like iteration on 1 thread, plus the reference-faithful iteration (identity
                       pairing, materialised rows, two N x 3 SVD least squares: what the reference costs) and an all-core figure
  N > 1                exchange actually used, exchange_fallback, per-rank pass times
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "icp-symm_amd", "py")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
VALU_PEAK_GINST = 1228.8       # wave64 vector instructions per ns-second: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 op
                               # (= the 157.3 TFLOP/s fp32 vector peak / 128 flops per wave64 FMA)
PMC_FILE = os.path.join(ROOT, "profiles", "r3_pmc.json")
N_SIMD = 256 * 4                # CUs x SIMDs
CLOCK_GHZ = 2.4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--workload", default="c4", choices=["c3", "c4", "c5"])
    ap.add_argument("--corr", default="tree", choices=["tree", "brute", "identity"])
    ap.add_argument("--mode", default="paper", choices=["paper", "quirks"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=2, help="oracle iterations timed for cpu_baseline")
    ap.add_argument("--repeats", type=int, default=1, help="timed repetitions; the best is reported")
    ap.add_argument("--host-loop", action="store_true", help="every solve on the host (no device-driven runs of iterations)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "shm", "torch"],
                    help="N>1: rccl = the library's own RCCL all-reduce (the default); shm = the library's shared-memory exchange "
                         "(ranks of one node, no collective kernel); torch = external exchange, the 40-double "
                         "record summed with torch.distributed.all_reduce between C calls (slow; lets the multi-rank scaffolding of "
                         "this file run where RCCL cannot, e.g. several ranks on ONE GPU with --dist-backend gloo)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    args = ap.parse_args()

    import numpy as np
    import torch
    import symmicp
    from symmicp import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("SYMMICP_BENCH_DEVICE"):                 # rehearsal: every rank on the same GPU
        local_rank = int(os.environ["SYMMICP_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    coll_dev = "cuda" if args.dist_backend == "nccl" else "cpu"

    def barrier_sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- data (every rank regenerates the same bits) ----------------------------------------------
    gen = dict(c3=synth.c3_uniform, c4=synth.c4_surface, c5=synth.c5_scan)[args.workload]
    if args.workload == "c5":      # the ray casting of a large scan is spread over this rank's share of the host's cores
        d = gen(args.points, workers=max(1, min(16, (os.cpu_count() or 1) // max(1, world))) if args.points >= 1_000_000 else 1)
    else:
        d = gen(args.points)
    # Sharded runs: a rank's share is a contiguous range of the caller's rows (symmicp_shard_range), so the rows should come in a
    # spatially coherent order, as a scanner delivers them (c5 does: ring by ring).  c3 / c4 are drawn in random order: put the source in
    # sweep order first (every rank, same permutation; before the timed region).  Compact shares keep the first pass's 64-query packets
    # compact: measured on one GPU, per-rank first pass at N = 8: 0.79 ms against 1.22 ms for random rows.
    sweep = world > 1 and args.workload in ("c3", "c4")
    if sweep:
        o = synth.sweep_order(d["src"])
        d["src"], d["src_n"] = np.ascontiguousarray(d["src"][o]), np.ascontiguousarray(d["src_n"][o])
    n_s, n_t = d["src"].shape[0], d["tgt"].shape[0]
    K, W = args.steps, args.warmup

    def new_engine():
        return symmicp.Engine(device=local_rank, mode=getattr(symmicp, "MODE_" + args.mode.upper()),
                              corr=getattr(symmicp, "CORR_" + args.corr.upper()), max_iters=K, fixed_iters=1,
                              host_loop=1 if args.host_loop else 0)

    eng = new_engine()
    exchange_requested, exchange_fallback, fallback_reason = args.exchange, False, None

    def attach_shm():
        job = ["%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getpid()) if rank == 0 else None]
        dist.broadcast_object_list(job, src=0)
        eng.comm_init_shm(world, rank, job[0])

    if world > 1 and args.exchange == "torch":
        eng.comm_init_rank(world, rank, None)                   # sharded, no communicator inside the library
    elif world > 1 and args.exchange == "shm":
        attach_shm()
    elif world > 1:
        # The library's own RCCL communicator; the unique id travels over torch.distributed.  Every rank takes part in the same
        # collectives whatever fails where: rank 0 broadcasts the id OR None (RCCL could not even be loaded), every rank then
        # reports whether it holds a communicator, and if any rank does not, ALL ranks fall back to the shared-memory exchange
        # (one node).  The JSON line says so (exchange, exchange_fallback).
        uid, err = [None], None
        if rank == 0:
            try:
                if os.environ.get("SYMMICP_BENCH_FAIL_UID"):            # test hook: RCCL cannot be loaded on rank 0
                    raise symmicp.SymmIcpError(symmicp.ERR_COMM, "forced by SYMMICP_BENCH_FAIL_UID")
                uid = [symmicp.comm_get_unique_id()]
            except symmicp.SymmIcpError as ex:
                err = "rank 0: no unique id (%s)" % ex
        dist.broadcast_object_list(uid, src=0)
        ok = 0
        if uid[0] is not None:
            try:
                eng.comm_init_rank(world, rank, uid[0])
                ok = 1
            except symmicp.SymmIcpError as ex:
                err = "rank %d: RCCL communicator failed (%s)" % (rank, ex)
        if err:
            print("[bench] " + err, file=sys.stderr)
        t = torch.tensor([ok], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 0:
            reasons = [None] * world
            dist.all_gather_object(reasons, err)
            fallback_reason = next((r for r in reasons if r), "unknown")
            if ok:      # this rank holds a communicator the others could not join: start over without it
                eng.close()
                eng = new_engine()
            args.exchange = "shm"
            exchange_fallback = True
            attach_shm()
    elif os.environ.get("SYMMICP_FORCE_COMM"):
        # rehearsal of the multi-GPU data path on one GPU: a real 1-rank RCCL communicator (all-reduce per pass)
        eng.comm_init_rank(1, 0, symmicp.comm_get_unique_id())
    # the step just before the path (MyICP::estimateNormals, myicp.cpp:152-172): k = 10 PCA normals of this rank's source rows on the GPU,
    # wall time of the whole call (upload + index build + k_normals_knn + read-back); reported, not part of the timed region (C4 says
    # "with normals": they are an input).  The first call also loads the code objects: the second is reported.
    normals_ms = None
    if args.corr == "tree" and not os.environ.get("SYMMICP_BENCH_NO_NORMALS"):
        b0, b1 = n_s * rank // world, n_s * (rank + 1) // world
        symmicp.estimate_normals(d["src"][b0:b0 + 4096], 10)
        tn0 = time.perf_counter()
        symmicp.estimate_normals(d["src"][b0:b1], 10, device=local_rank)
        normals_ms = (time.perf_counter() - tn0) * 1e3
    t0 = time.perf_counter()
    eng.set_target(d["tgt"], d["tgt_n"])
    t1 = time.perf_counter()
    eng.set_source(d["src"], d["src_n"])
    t2 = time.perf_counter()
    setup_s, set_target_s, set_source_s = t2 - t0, t1 - t0, t2 - t1
    st0 = eng.stats()

    def run(k):
        if world > 1 and args.exchange == "torch":
            # external exchange: begin / k x (all-reduce of the record, set_sums, step), one C call per arrow
            it = eng.begin()
            for _ in range(k):
                t = torch.tensor(it["sums"], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                eng.set_sums(t.cpu().numpy())
                it = eng.step()
            return
        # begin + exactly k steps, inside one C call (symmicp_align with fixed_iters: myicp.cpp:122-142 without the stop rule)
        eng.set_config(max_iters=k)
        r = eng.align()
        if r["status"] != 0 or r["iters"] != k:
            raise SystemExit("symmicp_align failed: %s" % (r,))

    # ---- warmup: W untimed steps ------------------------------------------------------------------
    run(W)
    # ---- timed: begin + exactly K steps, barrier + sync on both sides, max over ranks --------------
    # two HIP events bracket every pass inside the timed region (engine stream): the pass durations the roofline uses
    no_events = bool(os.environ.get("SYMMICP_BENCH_NO_EVENTS"))
    # (mode 3: every pass of the host loop, every 4th pass of a device-driven run -- an event record costs ~2.5 us of GPU timeline, and two
    # per pass made the timed region 7 % slower than the same alignment without events; SYMMICP_BENCH_ALL_EVENTS=1: every pass)
    all_events = bool(os.environ.get("SYMMICP_BENCH_ALL_EVENTS"))
    eng.enable_timing(0 if no_events else (1 if all_events else 3))
    best = None
    for _ in range(max(1, args.repeats)):
        eng.reset_stats()
        barrier_sync()
        t0 = time.perf_counter()
        run(K)
        barrier_sync()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        st_timed = eng.stats()
        T_timed = eng.transform()               # (of the timed run itself: the device-driven loop when it ran)
        if best is None or el < best[0]:
            best = (el, st_timed, T_timed)
    elapsed, st_timed, T_timed = best
    err_truth_timed = float(np.abs(T_timed - d["truth"]).max())
    # ---- the same K steps once more with HIP events around EVERY kernel: the per-kernel table.  Kept out of the timed
    # region because each event record costs a few microseconds of GPU timeline, comparable to a converged pass's kernels;
    # this run goes through the host loop (separate search / accumulate / reduce kernels per pass).
    eng.enable_timing(0 if no_events else 2)
    eng.reset_stats()
    barrier_sync()
    t0 = time.perf_counter()
    run(K)
    barrier_sync()
    elapsed_instrumented = time.perf_counter() - t0
    st = eng.stats()
    eng.enable_timing(0)
    T = eng.transform()
    err_truth = float(np.abs(T - d["truth"]).max())

    # ---- per-pass durations of the timed region -----------------------------------------------------
    np_timed = int(st_timed["passes_timed"])
    head = [float(x) for x in st_timed["pass_ms_head"][:min(8, np_timed)]]
    first_ms = head[0] if head else None
    second_third = head[1:3]
    n_rest = np_timed - min(np_timed, 4)
    converged_ms = (st_timed["sum_pass_ms"] - sum(head[:4])) / n_rest if n_rest > 0 else None
    # mean pass over ALL K + 1 passes of the timed region: the first four as timed, the others at the mean of those that carried events
    n_all = K + 1
    if np_timed >= n_all or converged_ms is None:
        pass_ms = st_timed["sum_pass_ms"] / max(1, np_timed)
    else:
        pass_ms = (sum(head[:4]) + converged_ms * (n_all - min(4, len(head)))) / n_all
    n_loc = int(eng.local_count())
    # wall time per iteration of the converged tail: the timed region minus its first four passes, over the iterations behind them (this
    # is what an iteration COSTS there: pass kernel + reduce + solve + kernel boundaries; converged_ms is the pass kernel(s) alone)
    tail_iters = (K + 1) - min(4, len(head))
    converged_iteration_ms = (elapsed * 1e3 - sum(head[:4])) / tail_iters if tail_iters > 0 and np_timed > 4 else None
    passes = dict(first_ms=round(first_ms, 5) if first_ms is not None else None,
                  second_third_ms=[round(x, 5) for x in second_third],
                  converged_ms=round(converged_ms, 5) if converged_ms is not None else None,
                  converged_iteration_ms=round(converged_iteration_ms, 5) if converged_iteration_ms is not None else None,
                  loop_passes=int(st_timed["loop_passes"]), timed=np_timed, of=K + 1, events="every pass" if (all_events or no_events) else "every pass of the host loop, every 4th pass of a device-driven run (standing for the three behind it)", loop="host" if (args.host_loop or os.environ.get("SYMMICP_HOST_LOOP") == "1" or args.exchange in ("shm", "torch")) else "device")

    # (an alignment that is still settling behind its fourth pass -- the scan-like pair -- runs some of those passes through the separate
    # search kernels: the figure above is then a mean over two regimes, and says so)
    unsettled = (K + 1 - min(4, len(head))) - int(st_timed["loop_passes"]) if passes["loop"] == "device" else 0
    if unsettled > 0:
        passes["converged_ms_mixes"] = "%d of the %d passes behind the fourth still ran the separate search kernels (pairs not settled yet): converged_ms is the mean over both kinds" % (unsettled, K + 1 - min(4, len(head)))

    # ---- kernel table of the instrumented run ---------------------------------------------------------
    names = symmicp.KERNEL_SLOTS
    kern = {names[k]: dict(launches=int(st["kernel_launches"][k]), total_ms=round(st["kernel_ms"][k], 4),
                           avg_ms=round(st["kernel_ms"][k] / max(1, st["kernel_launches"][k]), 5))
            for k in range(len(names)) if st["kernel_launches"][k] > 0}
    dom = max((k for k in kern if k not in ("k_final_reduce", "(gap)", "whole_pass")), key=lambda k: kern[k]["total_ms"], default="n/a")
    alg_bytes = st["bytes_algorithmic_per_pass"]           # this rank's share: N_loc*(48+4+4) + N_t*12
    split = "k_search_cells" in kern
    unit_name = "+".join(k for k in names[:4] if k in kern and k != "(gap)") if split else dom
    achieved = alg_bytes / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0

    # ---- PMC figures (HBM traffic, vector instructions per launch) cannot be read from inside this process: they come from
    # the committed rocprofv3 --pmc runs of this same command (profiles/r3_pmc.json; method and dates stated there)
    pmc, wk = {}, "%s:%d:%s:%s" % (args.workload, n_s, args.mode, args.corr)
    try:
        pj = json.load(open(PMC_FILE))
        if world == 1:
            pmc = pj["workloads"].get(wk, {})
    except Exception:
        pj = None
    traffic = pmc.get("whole_pass_hbm_bytes")
    traffic_source = ("profiles/r3_pmc.json (%s; %s)" % (wk, pj.get("method", "")) if traffic is not None else None) if pj else None
    roofline = dict(bound="hbm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_source, kernel=unit_name + " (whole pass)",
                    kernel_ms=round(pass_ms, 5), algorithmic_bytes_per_launch=int(alg_bytes), launches=np_timed,
                    dominant_kernel=dom, kernels_host_loop=kern)

    # ---- each regime against what bounds it -----------------------------------------------------------
    regimes = []

    def valu_entry(regime, kernel, ms, key):
        k = pmc.get("kernels", {}).get(key, {})
        insts = k.get("valu_wave_insts")
        e = dict(regime=regime, kernel=kernel, ms=round(ms, 5), bound="valu", unit="G wave64-inst/s", peak=VALU_PEAK_GINST,
                 model="vector-instruction issue: instructions per launch (SQ_INSTS_VALU, %s) / live duration; the search kernels are latency- and "
                       "issue-bound tree / cell walks, not HBM streams" % ("profiles/r3_pmc.json" if insts else "no PMC figure for this workload"),
                 valu_wave_insts_per_launch=insts, hbm_bytes_per_launch=k.get("hbm_bytes"))
        if insts and ms > 0:
            e["achieved"] = round(insts / (ms * 1e-3) / 1e9, 2)
            e["frac"] = round(e["achieved"] / VALU_PEAK_GINST, 4)
        act = k.get("valu_active_quad_cycles")
        if act and ms > 0:
            # measured: per-wave cycles with a vector instruction in flight (SQ_ACTIVE_INST_VALU quad-cycles x 4), summed over waves, over
            # SIMDs x live duration x nominal clock.  The mix of these kernels (readlane, compares, selects, DPP, SGPR operands) issues at
            # ~4 cycles per instruction, not the 2 of the peak above (scratch/ubench/valu_rates*.hip): this says how much issue capacity
            # is left.  An UPPER estimate of SIMD utilisation, not a utilisation: counter and duration come from different runs, the
            # clock is nominal, and in-flight instructions of two waves of one SIMD overlap by their pipeline latency (C5 8M first
            # pass: 1.09) -- so values near or above 1 mean "issue-bound", nothing finer
            e["valu_inflight_over_simd_cycles"] = round(act * 4.0 / (N_SIMD * ms * 1e-3 * CLOCK_GHZ * 1e9), 4)
        return e

    if args.corr == "tree" and first_ms:
        regimes.append(valu_entry("first pass (no previous pairs: every query searched)", "k_search_packet | k_search_walk", first_ms, "first_pass"))
        if second_third:
            regimes.append(valu_entry("passes 2-3 (pairs invalidated by the first big move: cell scans + walk)", "k_search_cells+k_search_walk+k_accumulate",
                                      sum(second_third) / len(second_third), "search_pass"))
        if converged_ms:
            fused = passes["loop"] == "device"
            b = n_loc * (72 if fused else 104)      # fused: p 12 + n 12 + record copy 32 + certificate 16 (a certified pair stores nothing); split: cells 48 + accumulate 56
            a = b / (converged_ms * 1e-3) / 1e9
            e = dict(regime="converged passes (pairs certified: the pass is a stream)", kernel="k_pass_fused<true>" if fused else "k_search_cells+k_accumulate",
                     ms=round(converged_ms, 5), bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS, achieved=round(a, 1), frac=round(a / HBM_PEAK_GBS, 4),
                     bytes_per_launch=int(b), hbm_bytes_per_launch=pmc.get("kernels", {}).get("converged", {}).get("hbm_bytes"),
                     model="bytes the kernels of the pass read and write per point x points / live duration of the PASS KERNEL(S) (events around them: the reduce, "
                           "the solve and the kernel boundaries of an iteration are not inside; frac_of_iteration prices the same bytes against the wall time "
                           "of a converged iteration)")
            if passes.get("converged_ms_mixes"):
                e["note"] = passes["converged_ms_mixes"] + "; the byte model holds for the fused passes only"
            if converged_iteration_ms:
                e["iteration_ms"] = round(converged_iteration_ms, 5)
                e["frac_of_iteration"] = round(b / (converged_iteration_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            regimes.append(e)
    elif args.corr == "identity":
        regimes.append(dict(regime="identity pairing (what the reference does): one streaming kernel", kernel="k_pass_identity", ms=round(pass_ms, 5), bound="hbm",
                            unit="GB/s", peak=HBM_PEAK_GBS, achieved=round(achieved, 1), frac=round(achieved / HBM_PEAK_GBS, 4), bytes_per_launch=int(alg_bytes),
                            model="48 B/point read (+24 B/point written back in incremental mode) / live pass duration"))
    elif args.corr == "brute":
        insts = n_loc * n_t * 11.0 / 64.0          # 3 sub, 3 mul, 2 add, compare, 2 selects per pair, unfused (k_nn_brute)
        a = insts / (pass_ms * 1e-3) / 1e9
        regimes.append(dict(regime="brute-force exact NN", kernel="k_nn_brute", ms=round(pass_ms, 5), bound="valu", unit="G wave64-inst/s", peak=VALU_PEAK_GINST,
                            achieved=round(a, 2), frac=round(a / VALU_PEAK_GINST, 4), valu_wave_insts_per_launch=insts,
                            model="N_s x N_t pairs x 11 unfused fp32 vector instructions per pair / 64 lanes, over the live pass duration"))

    # ---- CPU baseline: the oracle (a port; the reference itself cannot be built here) ----------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        ncores = os.cpu_count() or 1
        omode = O.MODE_PAPER if args.mode == "paper" else O.MODE_QUIRKS
        ocorr = dict(tree=O.CORR_GRID, brute=O.CORR_GRID, identity=O.CORR_IDENTITY)[args.corr]
        ci = max(1, args.cpu_iters)

        def timed_align(**kw):
            t0 = time.perf_counter()
            O.align(d["src"], d["src_n"], d["tgt"], d["tgt_n"], fixed_iters=True, **kw)
            return time.perf_counter() - t0

        O.set_threads(1)
        ct = timed_align(mode=omode, corr=ocorr, max_iters=ci)
        cpu = dict(value=round(ci / ct, 4), unit="iter/s", cores=1, kind="port",
                   sample="%d of %d iterations of the same %d-point workload (oracle/symmicp_oracle.c, exact uniform-grid NN, 1 thread, incl. its grid "
                          "build), %.1f s" % (ci, K, n_s, ct), host_cores=ncores)
        # what the reference itself costs per iteration (func.cpp:43-102 as written): identity pairing, M / N / c materialised, two N x 3
        # thin-SVD least squares in fp32 -- no correspondence search at all (myicp.cpp:128-131 is a todo); needs N_s == N_t
        if n_s == n_t:
            fi = 5
            ft = timed_align(mode=O.MODE_QUIRKS, corr=O.CORR_IDENTITY, solve=O.SOLVE_LITERAL, max_iters=fi)
            cpu["reference_faithful"] = dict(value=round(fi / ft, 3), unit="iter/s", cores=1,
                                             sample="%d iterations, identity pairing + materialised rows + two N x 3 SVD least squares (oracle's literal route of "
                                                    "func.cpp:64-73,85-88), %.2f s" % (fi, ft))
        # the like-for-like iteration on every core of this host (the two O(N) loops over OpenMP threads; the grid build stays serial)
        O.set_threads(ncores)
        at = timed_align(mode=omode, corr=ocorr, max_iters=ci)
        O.set_threads(1)
        cpu["all_cores"] = dict(value=round(ci / at, 4), unit="iter/s", cores=ncores, sample="%d iterations, %d OpenMP threads, %.1f s" % (ci, ncores, at))

    # ---- per-rank figures for N > 1 (a bad scaling curve has to be diagnosable from the line) -------------
    per_rank = None
    ar_n = int(st_timed.get("allreduce_timed", 0))
    mine = dict(rank=rank, n_loc=n_loc, first_pass_ms=passes["first_ms"], second_third_ms=passes["second_third_ms"], converged_ms=passes["converged_ms"], pass_ms=round(pass_ms, 5),
                converged_iteration_ms=passes["converged_iteration_ms"], loop=passes["loop"], loop_passes=passes["loop_passes"],
                allreduce_us=round(1e3 * st_timed["allreduce_ms"] / ar_n, 2) if ar_n else None, allreduce_timed=ar_n,
                set_target_ms=round(set_target_s * 1e3, 2), set_source_ms=round(set_source_s * 1e3, 2), kernels_host_loop=kern)
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        exch_name = {"rccl": "RCCL", "shm": "shared-memory", "torch": "torch.distributed(" + args.dist_backend + ")"}[args.exchange]
        out = {
            "metric": "icp_iterations_per_sec",
            "value": round(K / elapsed, 3),
            "unit": "iter/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (" (source rows in sweep order for the row-range shares)" if sweep else ""),
            "config": {"workload": "%s: %d-pt synthetic cloud pair with normals, %d iters, %s mode, %s correspondences"
                                   % (args.workload.upper(), n_s, K, args.mode.upper(), args.corr),
                       "n_source": n_s, "n_target": n_t, "iters": K,
                       "parallelism": "source sharded x%d, target replicated, 40-double %s all-reduce per pass" % (world, exch_name)},
            "mcorr_per_sec": round(n_s * K / elapsed / 1e6, 2),
            "mcorr_per_sec_first_pass": round(n_loc * world / (first_ms * 1e-3) / 1e6, 2) if first_ms else None,
            "passes": passes,
            "ms_per_step_with_kernel_events": round(elapsed_instrumented / K * 1e3, 5),
            "final_transform_max_abs_err_vs_truth": err_truth_timed,
            "final_transform_max_abs_err_vs_truth_host_loop_run": err_truth,
            "setup_ms": {"normals": round(normals_ms, 2) if normals_ms is not None else None, "upload": round(st0["upload_ms"], 2), "index_build": round(st0["build_ms"], 2),
                         "set_target": round(set_target_s * 1e3, 2), "set_source": round(set_source_s * 1e3, 2),
                         "set_target+set_source_wall": round(setup_s * 1e3, 2), "grid_level": st0["grid_level"],
                         "tree_levels": st0["tree_levels"]},
            "roofline": roofline,
            "roofline_by_regime": regimes,
            "cpu_baseline": cpu,
        }
        if world > 1:
            out["exchange"] = args.exchange
            out["exchange_requested"] = exchange_requested
            out["exchange_fallback"] = exchange_fallback
            if exchange_fallback:
                out["exchange_fallback_reason"] = fallback_reason
            out["per_rank"] = per_rank
            # what ONE GPU said about this split beforehand (scratch/predict_ranks.py: N sharded contexts run one after the other, kernel time per
            # pass, max over ranks): lets a flat curve be attributed to the first pass, the collective or the solve
            try:
                out["predicted_per_rank_ms"] = open(os.path.join(ROOT, "profiles", "r3_predict_ranks_%d.log" % world)).read().strip().splitlines()[-1]
            except OSError:
                out["predicted_per_rank_ms"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
