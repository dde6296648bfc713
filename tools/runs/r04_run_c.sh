set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_c; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log; tail -4 $O/gpu_tests.log
python tools/ep_grad_errors.py > $O/ep_grad_errors.log 2>&1; echo "ep grad rc=$?" | tee -a $O/progress.log; cat $O/ep_grad_errors.log
for n in 8192 12288; do
  for cfg in "GPCORE_X=0" "GPCORE_MAIN_PRIORITY=1" "GPCORE_FAR_SMALL=1" "GPCORE_MAIN_PRIORITY=1 GPCORE_FAR_SMALL=1"; do
    echo "== $cfg" >> $O/fit.log; env $cfg python tools/fit_only.py $n 10 >> $O/fit.log 2>&1
  done
done
echo "fit done" | tee -a $O/progress.log; cat $O/fit.log
python tools/gram_perf.py > $O/gram_perf.log 2>&1; cat $O/gram_perf.log
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?" | tee -a $O/progress.log
python -c "import json;d=json.load(open('$O/bench_c2.json'));print(d['value'],d['ms_per_step'],d['cholesky']['fit_ms'],d.get('e2e_ms_per_step'),d['cpu_baseline']['value'],d['c3_sharded']['settings_per_s'])"
python bench.py --workload c3 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc=$?" | tee -a $O/progress.log
python bench.py --workload c4 --steps 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?" | tee -a $O/progress.log
python -c "import json;d=json.load(open('$O/bench_c3.json'));print('c3',d['value']);d=json.load(open('$O/bench_c4.json'));print('c4',d['value'],d['ep_grid'])"
echo end | tee -a $O/progress.log
