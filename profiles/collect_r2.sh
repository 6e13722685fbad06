#!/bin/bash
# Round-2 measurement run (one gpurun call).  Order matters: the PMC passes of the default bench command come first and
# profiles/collect_r2.py --pmc-only turns them into profiles/r2_pmc.json ON THE BOX, so that the bench lines taken afterwards carry the
# PMC-derived fields (traffic, instructions per launch) of the very kernels they time.  PMC runs carry no trace domain but
# --kernel-trace (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass: separate runs).
# Output under gpurun_out/r2_final/; `python profiles/collect_r2.py` afterwards (here) copies the summaries into profiles/.
#   rm -rf gpurun_out/r2_final; gpurun --timeout 1150 -- 'bash profiles/collect_r2.sh'; python profiles/collect_r2.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2_final
rm -rf $O && mkdir -p $O
run() { echo "== $*" >&2; timeout -k 10 600 "$@"; }
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline --warmup 0 > $O/pmc_fetch.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline --warmup 0 > $O/pmc_write.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- python3 bench.py --no-cpu-baseline --warmup 0 > $O/pmc_sq.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_id_fetch -- python3 bench.py --corr identity --mode quirks --points 8000000 --no-cpu-baseline --warmup 0 > $O/pmc_id_fetch.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_id_write -- python3 bench.py --corr identity --mode quirks --points 8000000 --no-cpu-baseline --warmup 0 > $O/pmc_id_write.log 2>&1 || exit 1
run python3 profiles/collect_r2.py --pmc-only > $O/pmc_summary.log 2>&1 || exit 1
run python3 bench.py                                                        > $O/bench_default.json          2> $O/bench_default.err          || exit 1
run python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline                > $O/bench_driver_20.json        2>> $O/bench_default.err         || exit 1
run python3 bench.py --host-loop --no-cpu-baseline                          > $O/bench_host_loop.json        2>> $O/bench_default.err         || exit 1
run python3 bench.py --workload c3 --points 100000 --no-cpu-baseline        > $O/bench_c3_100k_tree.json     2>> $O/bench_default.err         || exit 1
run python3 bench.py --workload c3 --points 100000 --corr brute --no-cpu-baseline > $O/bench_c3_100k_brute.json 2>> $O/bench_default.err      || exit 1
run python3 bench.py --workload c5 --points 8000000 --steps 50 --no-cpu-baseline > $O/bench_c5_8M_tree.json  2>> $O/bench_default.err        || exit 1
run python3 bench.py --workload c5 --points 2000000 --steps 50 --no-cpu-baseline > $O/bench_c5_2M_tree.json  2>> $O/bench_default.err        || exit 1
run python3 bench.py --corr identity --mode quirks --points 8000000 --no-cpu-baseline > $O/bench_identity_8M.json 2>> $O/bench_default.err   || exit 1
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline            > $O/trace.log 2>&1   || exit 1
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_identity -- python3 bench.py --corr identity --mode quirks --points 8000000 --no-cpu-baseline > $O/trace_identity.log 2>&1 || exit 1
ls $O
