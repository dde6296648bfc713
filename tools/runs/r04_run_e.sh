set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_e; mkdir -p $O; rm -f $O/*
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py -q -x -k "cholesky" > $O/chol_tests.log 2>&1; rc=$?; echo "chol tests rc=$rc" | tee -a $O/progress.log; tail -15 $O/chol_tests.log
[ $rc -eq 0 ] || exit 1
for n in 4096 6144 8192 12288 16384; do
  for cfg in "GPCORE_CHOL_MEGA=0" "GPCORE_CHOL_MEGA=1"; do
    echo "== $cfg" >> $O/fit.log; timeout -k 10 120 env $cfg python tools/fit_only.py $n 10 >> $O/fit.log 2>&1 || { echo "FAILED rc=$?" >> $O/fit.log; cat $O/fit.log; exit 1; }
  done
done
echo "fit done" | tee -a $O/progress.log; cat $O/fit.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c3 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?" | tee -a $O/progress.log
python -c "import json;d=json.load(open('$O/bench_c2.json'));print(d['value'],d['ms_per_step'],d['cholesky'])"
echo end | tee -a $O/progress.log
