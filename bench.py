#!/usr/bin/env python3
"""bench.py -- headline benchmark of the symmetric-ICP hot path on MI355X.

Workload (BASELINE.json configs[3], "C4"): synthetic 1M-point surface cloud pair with analytic
normals, 30 ICP iterations, paper-correct symmetric objective, exact nearest-neighbour
correspondences through the Morton grid + box tree, source sharded over N GPUs with one RCCL
all-reduce of 40 doubles per pass.

A "step" is one ICP iteration = one trip of the reference loop body (ICP/myicp.cpp:123-142):
host 6x6 solve + compose, one fused GPU pass (transform + NN search + M/N/c rows + 37 fp64 sums),
final reduce, [all-reduce], read-back.  The timed region is begin() (the initial correspondence
pass, myicp.cpp:122) plus exactly K steps, with inputs already resident in HBM; upload and index
build are reported separately.

    python bench.py                     # N=1, K=30, W=5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
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

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


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

    def barrier_sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- data (every rank regenerates the same bits) ----------------------------------------------
    gen = dict(c3=synth.c3_uniform, c4=synth.c4_surface, c5=synth.c5_scan)[args.workload]
    d = gen(args.points)
    n_s, n_t = d["src"].shape[0], d["tgt"].shape[0]
    K, W = args.steps, args.warmup

    eng = symmicp.Engine(device=local_rank, mode=getattr(symmicp, "MODE_" + args.mode.upper()),
                         corr=getattr(symmicp, "CORR_" + args.corr.upper()), max_iters=K, fixed_iters=1)
    if world > 1 and args.exchange == "torch":
        eng.comm_init_rank(world, rank, None)                   # sharded, no communicator inside the library
    elif world > 1 and args.exchange == "shm":
        job = ["%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getpid()) if rank == 0 else None]
        dist.broadcast_object_list(job, src=0)
        eng.comm_init_shm(world, rank, job[0])
    elif world > 1:
        # the library's own RCCL communicator (the unique id travels over torch.distributed).  Should RCCL refuse on any
        # rank, every rank falls back to the shared-memory exchange (one node) instead of failing the run; the JSON
        # line says which exchange was used.
        ok = 1
        try:
            uid = [symmicp.comm_get_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            eng.comm_init_rank(world, rank, uid[0])
        except symmicp.SymmIcpError as ex:
            ok = 0
            print("[bench] rank %d: RCCL communicator failed (%s)" % (rank, ex), file=sys.stderr)
        t = torch.tensor([ok], dtype=torch.int32, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 0:
            if ok:      # this rank holds a communicator the others could not join: start over without it
                eng.close()
                eng = symmicp.Engine(device=local_rank, mode=getattr(symmicp, "MODE_" + args.mode.upper()),
                                     corr=getattr(symmicp, "CORR_" + args.corr.upper()), max_iters=K, fixed_iters=1)
            args.exchange = "shm"
            job = ["%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getpid()) if rank == 0 else None]
            dist.broadcast_object_list(job, src=0)
            eng.comm_init_shm(world, rank, job[0])
    elif os.environ.get("SYMMICP_FORCE_COMM"):
        # rehearsal of the multi-GPU data path on one GPU: a real 1-rank RCCL communicator (all-reduce + publish per pass)
        eng.comm_init_rank(1, 0, symmicp.comm_get_unique_id())
    t0 = time.perf_counter()
    eng.set_target(d["tgt"], d["tgt_n"])
    eng.set_source(d["src"], d["src_n"])
    setup_s = time.perf_counter() - t0
    st0 = eng.stats()

    def run(k):
        if world > 1 and args.exchange == "torch":
            # external exchange: begin / k x (all-reduce of the record, set_sums, step), one C call per arrow
            dev = "cuda" if args.dist_backend == "nccl" else "cpu"
            it = eng.begin()
            for _ in range(k):
                t = torch.tensor(it["sums"], dtype=torch.float64, device=dev)
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
    # two HIP events bracket every pass inside the timed region (engine stream): the pass duration the roofline uses
    eng.enable_timing(0 if os.environ.get("SYMMICP_BENCH_NO_EVENTS") else 1)
    best = None
    for _ in range(max(1, args.repeats)):
        eng.reset_stats()
        barrier_sync()
        t0 = time.perf_counter()
        run(K)
        barrier_sync()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        st_timed = eng.stats()
        if best is None or el < best[0]:
            best = (el, st_timed)
    elapsed, st_timed = best
    # ---- the same K steps once more with HIP events around EVERY kernel: the per-kernel table.  Kept out of the timed
    # region because each event record costs a few microseconds of GPU timeline, comparable to a converged pass's kernels.
    eng.enable_timing(0 if os.environ.get("SYMMICP_BENCH_NO_EVENTS") else 2)
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

    # ---- roofline of the dominant kernel, measured live with HIP events on the ctx stream -------------
    # One pass = k_search_cells -> k_search_walk -> k_accumulate (-> k_final_reduce).
    # The dominant kernel is the one with the largest total time over the timed region; its average launch
    # duration prices the pass's algorithmic bytes (N_loc*(48+4+4) + N_t*12 per launch).
    names = symmicp.KERNEL_SLOTS
    kern = {names[k]: dict(launches=int(st["kernel_launches"][k]), total_ms=round(st["kernel_ms"][k], 4),
                           avg_ms=round(st["kernel_ms"][k] / max(1, st["kernel_launches"][k]), 5))
            for k in range(len(names)) if st["kernel_launches"][k] > 0}
    dom = max((k for k in kern if k not in ("k_final_reduce", "(gap)", "whole_pass")), key=lambda k: kern[k]["total_ms"], default="n/a")
    alg_bytes = st["bytes_algorithmic_per_pass"]           # this rank's share: N_loc*(48+4+4) + N_t*12
    # pass duration from the two events per pass recorded INSIDE the timed region (all kernels of one pass, without the final reduce)
    pass_ms = st_timed["sum_pass_ms"] / max(1, st_timed["passes"])
    split = "k_search_cells" in kern
    # The NN pass is three kernels; the contract's algorithmic bytes are per PASS, so they are priced against the
    # summed average duration of the pass's kernels (pricing them against one of the three would flatter it).
    unit_name = "+".join(k for k in names[:4] if k in kern and k != "(gap)") if split else dom
    achieved = alg_bytes / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
    # HBM traffic per pass: PMC counters cannot be read from inside this process, so they come from the committed
    # rocprofv3 --pmc runs of this same command (profiles/r1_pmc_traffic.json, method stated there)
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
        wk = "%s:%d:%s:%s" % (args.workload, n_s, args.mode, args.corr)
        if world == 1 and wk in tj["workloads"]:
            w = tj["workloads"][wk]
            traffic = int(w["_whole_pass_hbm_bytes"])       # all kernels of a pass, total bytes / passes
    except Exception:
        pass
    roofline = dict(bound="hbm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, kernel=unit_name, kernel_ms=round(pass_ms, 5),
                    algorithmic_bytes_per_launch=int(alg_bytes), launches=int(st_timed["passes"]),
                    dominant_kernel=dom, kernels=kern)

    # ---- CPU baseline: the oracle (a port; the reference itself cannot be built here), 1 thread -----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        ci = max(1, args.cpu_iters)
        t0 = time.perf_counter()
        ro = O.align(d["src"], d["src_n"], d["tgt"], d["tgt_n"], mode=O.MODE_PAPER if args.mode == "paper" else O.MODE_QUIRKS,
                     corr=dict(tree=O.CORR_GRID, brute=O.CORR_GRID, identity=O.CORR_IDENTITY)[args.corr], max_iters=ci, fixed_iters=True)
        ct = time.perf_counter() - t0
        cpu = dict(value=round(ci / ct, 4), unit="iter/s", cores=1, kind="port",
                   sample="%d of %d iterations of the same %d-point workload (oracle/symmicp_oracle.c, exact uniform-grid NN, "
                          "1 thread, incl. its grid build), %.1f s" % (ci, K, n_s, ct), host_cores=os.cpu_count())

    if rank == 0:
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
            "data": "synthetic",
            "config": {"workload": "%s: %d-pt synthetic cloud pair with normals, %d iters, %s mode, %s correspondences"
                                   % (args.workload.upper(), n_s, K, args.mode.upper(), args.corr),
                       "n_source": n_s, "n_target": n_t, "iters": K,
                       "parallelism": "source sharded x%d, target replicated, 40-double %s all-reduce per pass"
                                      % (world, {"rccl": "RCCL", "shm": "shared-memory", "torch": "torch.distributed(" + args.dist_backend + ")"}[args.exchange])},
            "mcorr_per_sec": round(n_s * K / elapsed / 1e6, 2),
            "ms_per_step_with_kernel_events": round(elapsed_instrumented / K * 1e3, 5),
            "final_transform_max_abs_err_vs_truth": err_truth,
            "setup_ms": {"upload": round(st0["upload_ms"], 2), "index_build": round(st0["build_ms"], 2),
                         "set_target+set_source_wall": round(setup_s * 1e3, 2), "grid_level": st0["grid_level"],
                         "tree_levels": st0["tree_levels"]},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
