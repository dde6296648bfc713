set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_t; mkdir -p $O; rm -f $O/*
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "cholesky" > $O/chol_tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $O/progress.log; tail -5 $O/chol_tests.log
[ $rc -eq 0 ] || exit 1
for n in 5120 6144 8192 10240 12288; do
  echo "== mega" >> $O/fit.log; timeout -k 10 120 env GPCORE_CHOL_MEGA=1 python tools/fit_only.py $n 10 >> $O/fit.log 2>&1 || { echo "FAILED rc=$?" >> $O/fit.log; cat $O/fit.log; exit 1; }
done
echo "fit done" | tee -a $O/progress.log; cat $O/fit.log
timeout -k 10 120 python tools/mega_trace.py 8192 >> $O/mega_trace.log 2>&1 || exit 1
head -14 $O/mega_trace.log
echo end | tee -a $O/progress.log
