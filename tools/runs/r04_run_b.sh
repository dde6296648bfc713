set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_b; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_configs.py -q -x -k "cholesky" > $O/chol_tests.log 2>&1; echo "chol tests rc=$?" | tee -a $O/progress.log; tail -3 $O/chol_tests.log
python -m pytest tests/test_gpu_ep_edge_cases.py -q > $O/ep_edge_tests.log 2>&1; echo "ep edge rc=$?" | tee -a $O/progress.log; tail -25 $O/ep_edge_tests.log
for n in 4096 6144 8192 12288 16384; do
  for cfg in "GPCORE_CHOL_SPLIT=0" "GPCORE_CHOL_SPLIT=1 GPCORE_CHOL_BELOW=3" "GPCORE_CHOL_SPLIT=1 GPCORE_CHOL_BELOW=2"; do
    echo "== $cfg" >> $O/fit.log; env $cfg python tools/fit_only.py $n 10 >> $O/fit.log 2>&1
  done
done
echo "fit done" | tee -a $O/progress.log; cat $O/fit.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c3 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?" | tee -a $O/progress.log
python -c "import json;d=json.load(open('$O/bench_c2.json'));print(d['value'],d['ms_per_step'],d['cholesky'],d.get('e2e_ms_per_step'))"
cd /tmp && export TMPDIR=/tmp
GPCORE_BENCH_PROGRESS=1 GPCORE_EP_FUSED=0 timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c4 -o p -- python3 $R/bench.py --workload c4 --steps 1 --no-c3 --no-roofline-events > $O/pmc_c4_WRITE_SIZE_noevents.log 2>&1 && echo "pmc c4 (no events) ok" | tee -a $O/progress.log && \
GPCORE_BENCH_PROGRESS=1 GPCORE_EP_FUSED=0 timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c4e -o p -- python3 $R/bench.py --workload c4 --steps 1 --no-c3 > $O/pmc_c4_WRITE_SIZE_events.log 2>&1 && echo "pmc c4 (events) ok" | tee -a $O/progress.log
rm -rf $O/pmc_c4 $O/pmc_c4e
echo end | tee -a $O/progress.log
