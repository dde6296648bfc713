#!/bin/bash
# Round evidence on the GPU box: bench lines, rocprofv3 kernel stats of the headline command, PMC passes (one counter set per run,
# kernel-trace only), per-kernel breakdown of C4.  usage (from the repo root on the box): bash tools/collect_evidence.sh <tag>
set -o pipefail
TAG=${1:-r02_c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_c2.json 2> $O/bench_c2.err && echo "c2 ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1 && echo "stats ok"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1 && echo "fetch ok"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1 && echo "write ok"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_mfma.log 2>&1 && echo "mfma ok"
python3 $R/bench.py --workload c3 --steps 3 --warmup 1 > $O/bench_c3.json 2>&1 && echo "c3 ok"
python3 $R/bench.py --workload c4 --steps 2 > $O/bench_c4.json 2>&1 && echo "c4 ok"
python3 $R/bench.py --workload c5 --steps 1 --warmup 1 > $O/bench_c5.json 2>&1 && echo "c5 ok"
rocprofv3 --kernel-trace -d $O/trace_c4 -o c4 -- python3 $R/tools/ep_sweeps.py 4096 30 > $O/trace_c4.log 2>&1 && echo "c4 trace ok"
python3 $R/tools/trace_breakdown.py $O/trace_c4/c4_results.db 15 > $O/c4_kernel_breakdown.txt 2>&1
python3 $R/tools/sweep_summary.py $O/trace_c4/c4_results.db > $O/c4_sweep_summary.txt 2>&1
python3 $R/tools/chain_kernels.py 8192 3 > $O/chain_kernels.log 2>&1
[ -x $R/tools/lab/potrf_lab ] && $R/tools/lab/potrf_lab > $O/potrf_phases.log 2>&1
for n in 1024 2048 8192; do python3 $R/tools/ep_sweeps.py $n 8 >> $O/ep_sizes.log 2>&1; GPCORE_EP_PIPELINE=0 python3 $R/tools/ep_sweeps.py $n 8 >> $O/ep_sizes.log 2>&1; done
python3 $R/tools/gram_perf.py > $O/gram_perf.log 2>&1
python3 $R/tools/write_bw.py > $O/write_bw.log 2>&1
find $O -name "*counter_collection.csv" | head
find $O -name "*kernel_stats.csv" | head
