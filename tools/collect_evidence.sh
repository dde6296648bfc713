#!/bin/bash
# Round evidence on the GPU box: bench lines of every workload, rocprofv3 kernel stats of the headline command, PMC passes per workload
# (one counter set per run, kernel-trace only; the program goes directly after `--`), EP sweep timeline, chain-kernel stamps, mesh
# throughput.  usage (from the repo root on the box): bash tools/collect_evidence.sh <tag> [quick|full] [first stage: 0 bench lines, 1 pmc c2/c3, 2 pmc c4/c5, 3 the rest]
# SKIP_PMC=1 leaves the PMC stages out (round 4: the PMC passes are collected first, `<tag> quick 1`, their summaries copied to
# profiles/<round>_pmc_*_summary.json, and the bench lines -- which read those files for roofline.traffic -- in a second call).
# Every step appends to $O/progress.log so that a long collection never looks hung.
set -o pipefail
TAG=${1:-r04_final}
QUICK=${2:-}
FROM=${3:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
say() { echo "$(date +%T) $*" | tee -a $O/progress.log; }
pmc() {   # pmc <name> <bench args...>: FETCH / WRITE / MFMA-busy passes of one workload, summarised per kernel
    local name=$1; shift
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        local tagset=$(echo $set | cut -d' ' -f1)
        timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${name}_$tagset -o p -- python3 $R/bench.py "$@" > $O/pmc_${name}_$tagset.log 2>&1 && say "pmc $name $tagset ok"
    done
    python3 $R/tools/pmc_summary.py $O/pmc_${name}_summary.json "rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 bench.py $*; one counter set per run (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)" $(find $O/pmc_${name}_* -name "*counter_collection.csv") > $O/pmc_${name}_summary.txt 2>&1 && say "pmc $name summary ok"
    rm -rf $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE $O/pmc_${name}_SQ_VALU_MFMA_BUSY_CYCLES
}
if [ $FROM -le 0 ]; then
python3 $R/bench.py > $O/bench_c2.json 2> $O/bench_c2.err && say "c2 ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c3 > $O/stats.log 2>&1 && say "stats ok"
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_c2_kernel_stats.csv 2>/dev/null; rm -rf $O/stats
python3 $R/bench.py --workload c3 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err && say "c3 ok"
python3 $R/bench.py --workload c4 --steps 2 > $O/bench_c4.json 2> $O/bench_c4.err && say "c4 ok"
python3 $R/bench.py --workload c5 --steps 1 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err && say "c5 ok"
fi
if [ $FROM -le 1 ] && [ -z "$SKIP_PMC" ]; then
pmc c2 --steps 1 --warmup 1 --no-cpu-baseline --no-c3
pmc c3 --workload c3 --steps 1 --warmup 1
fi
if [ $FROM -le 2 ] && [ -z "$SKIP_PMC" ]; then
# (c4: the fused chain kernel announces its solved rows to a kernel that waits on the side stream; under --pmc the profiler runs one
#  kernel at a time, so the waiter would only ever time out -- the counters are taken on the two-launch form, same GEMM kernels, and
#  without the ep_grid block, whose lockstep batch always runs the fused kernel; --no-roofline-events: with ~19 000 HIP event records
#  outstanding between two host synchronisations the WRITE_SIZE pass of round 3 stopped, profiles/r04_b_pmc_c4_WRITE_SIZE_*.log)
GPCORE_EP_FUSED=0 pmc c4 --workload c4 --steps 1 --no-c3 --no-roofline-events
pmc c5 --workload c5 --steps 1 --warmup 1 --test-points 262144
fi
[ "$QUICK" = "quick" ] && exit 0
rocprofv3 --kernel-trace -d $O/trace_c4 -o c4 -- python3 $R/tools/ep_sweeps.py 4096 30 > $O/trace_c4.log 2>&1 && say "c4 trace ok"
python3 $R/tools/trace_breakdown.py $(find $O/trace_c4 -name "*.db" | head -1) 15 > $O/c4_kernel_breakdown.txt 2>&1
python3 $R/tools/sweep_summary.py $(find $O/trace_c4 -name "*.db" | head -1) > $O/c4_sweep_summary.txt 2>&1
rm -rf $O/trace_c4
python3 $R/tools/chain_kernels.py 8192 3 > $O/chain_kernels.log 2>&1 && say "chain kernels ok"
# the factorisation alone, launch-per-step form against the single persistent launch, and where the single launch's time goes
for n in 4096 6144 8192 10240 12288 16384; do for m in 0 1; do echo "== GPCORE_CHOL_MEGA=$m"; GPCORE_CHOL_MEGA=$m python3 $R/tools/fit_only.py $n 10; done; done > $O/fit_mega.log 2>&1; say "fit ok"
python3 $R/tools/mega_trace.py 8192 > $O/mega_trace.log 2>&1 && say "mega trace ok"
for n in 1024 2048 8192; do python3 $R/tools/ep_sweeps.py $n 8 >> $O/ep_sizes.log 2>&1; GPCORE_EP_FUSED=0 python3 $R/tools/ep_sweeps.py $n 8 >> $O/ep_sizes.log 2>&1; done; say "ep sizes ok"
(GPCORE_EP_LOCKSTEP=0 python3 $R/tools/ep_mesh_perf.py 12 10 4096 1; python3 $R/tools/ep_mesh_perf.py 12 10 4096 1; GPCORE_EP_LOCKSTEP=0 python3 $R/tools/ep_mesh_perf.py 12 10 2048; python3 $R/tools/ep_mesh_perf.py 12 10 2048) > $O/ep_mesh_perf.log 2>&1; say "mesh ok"
python3 $R/tools/gram_perf.py > $O/gram_perf.log 2>&1; say "gram ok"
[ -f $R/tools/lab/libgpcore_stamps.so ] && for q in 0 1; do echo "== quiet=$q"; GPCORE_LIB_PATH=$R/tools/lab/libgpcore_stamps.so python3 $R/tools/ep_block2_stamps.py 4096 $q; done > $O/block2_stamps.txt 2>&1
say "done"
