set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_a; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -3 $O/gpu_tests.log
timeout -k 10 300 tools/lab/gemm_lab 20 > $O/gemm_lab.log 2>&1 && echo "lab ok" | tee -a $O/progress.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_gram -o p -- python3 $R/tools/gram_perf.py 4096 8192 > $O/pmc_gram_WRITE_SIZE.log 2>&1 && echo "pmc gram ok" | tee -a $O/progress.log && \
GPCORE_BENCH_PROGRESS=1 GPCORE_EP_FUSED=0 timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c4 -o p -- python3 $R/bench.py --workload c4 --steps 1 --no-c3 > $O/pmc_c4_WRITE_SIZE.log 2>&1 && echo "pmc c4 ok" | tee -a $O/progress.log
echo "end" | tee -a $O/progress.log
ls $O
