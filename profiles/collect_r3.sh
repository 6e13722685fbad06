#!/bin/bash
# Round-3 measurement run (one gpurun call; ~12 min of box time).  As in round 2 the PMC passes come first and profiles/collect_r3.py --pmc-only
# turns them into profiles/r3_pmc.json ON THE BOX, so that the bench lines taken afterwards carry the PMC-derived fields of the very
# kernels they time -- now for BOTH quoted workloads: C4 (1M surface pair, the contract workload) and C5 (8M scan-like pair, 50
# iterations: BASELINE config 5 asks for "rocprof HBM GB/s vs roofline" there).  PMC runs carry no trace domain but --kernel-trace
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass: separate runs); the program comes directly after `--`.
# Two gpurun calls (the 8M scan pair takes half a minute of host time to generate, per run):
#   rm -rf gpurun_out/r3_final; gpurun --timeout 1190 -- 'bash profiles/collect_r3.sh pmc' && python profiles/collect_r3.py --pmc-only
#   gpurun --timeout 1190 -- 'bash profiles/collect_r3.sh bench' && python profiles/collect_r3.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_final
mkdir -p $O
PART=${1:-all}
run() { echo "== $*" >&2; timeout -k 10 600 "$@"; }
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"
C4="bench.py --no-cpu-baseline --warmup 0"
C5="bench.py --workload c5 --points 8000000 --steps 50 --no-cpu-baseline --warmup 0"
ID="bench.py --corr identity --mode quirks --points 8000000 --no-cpu-baseline --warmup 0"
if [ "$PART" != bench ]; then
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_c4_fetch -- python3 $C4 > $O/pmc_c4_fetch.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c4_write -- python3 $C4 > $O/pmc_c4_write.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_c4_sq -- python3 $C4 > $O/pmc_c4_sq.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_c5_fetch -- python3 $C5 > $O/pmc_c5_fetch.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c5_write -- python3 $C5 > $O/pmc_c5_write.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_c5_sq -- python3 $C5 > $O/pmc_c5_sq.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_id_fetch -- python3 $ID > $O/pmc_id_fetch.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_id_write -- python3 $ID > $O/pmc_id_write.log 2>&1 || exit 1
run python3 profiles/collect_r3.py --pmc-only > $O/pmc_summary.log 2>&1 || exit 1
fi
[ "$PART" = pmc ] && { ls $O; exit 0; }
run python3 bench.py                                                        > $O/bench_default.json          2> $O/bench_default.err          || exit 1
run python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline                > $O/bench_driver_20.json        2>> $O/bench_default.err         || exit 1
run python3 bench.py --host-loop --no-cpu-baseline                          > $O/bench_host_loop.json        2>> $O/bench_default.err         || exit 1
run python3 bench.py --workload c3 --points 100000 --no-cpu-baseline        > $O/bench_c3_100k_tree.json     2>> $O/bench_default.err         || exit 1
run python3 bench.py --workload c3 --points 100000 --corr brute --no-cpu-baseline > $O/bench_c3_100k_brute.json 2>> $O/bench_default.err      || exit 1
run python3 bench.py --workload c5 --points 8000000 --steps 50 --no-cpu-baseline > $O/bench_c5_8M_tree.json  2>> $O/bench_default.err        || exit 1
run python3 bench.py --workload c5 --points 2000000 --steps 50 --no-cpu-baseline > $O/bench_c5_2M_tree.json  2>> $O/bench_default.err        || exit 1
run python3 bench.py --corr identity --mode quirks --points 8000000 --no-cpu-baseline > $O/bench_identity_8M.json 2>> $O/bench_default.err   || exit 1
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline            > $O/trace.log 2>&1   || exit 1
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -- python3 bench.py --workload c5 --points 8000000 --steps 50 --no-cpu-baseline > $O/trace_c5.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_identity -- python3 $ID > $O/trace_identity.log 2>&1 || exit 1
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_normals -- python3 scratch/time_normals.py > $O/trace_normals.log 2>&1 || exit 1
for n in 2 4 8; do run python3 scratch/predict_ranks.py c4 1000000 $n 20 sweep > $O/predict_ranks_$n.log 2>&1 || exit 1; done
ls $O
