set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_h; mkdir -p $O; rm -f $O/*
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $O/progress.log; tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit 1
for n in 4096 5120 6144 7168 8192 10240 12288 14336 16384; do
  for cfg in "GPCORE_CHOL_MEGA=0" "GPCORE_CHOL_MEGA=1"; do
    echo "== $cfg" >> $O/fit.log; timeout -k 10 120 env $cfg python tools/fit_only.py $n 10 >> $O/fit.log 2>&1 || { echo "FAILED rc=$?" >> $O/fit.log; cat $O/fit.log; exit 1; }
  done
done
echo "fit done" | tee -a $O/progress.log; cat $O/fit.log
for n in 4096 8192 16384; do timeout -k 10 120 python tools/mega_trace.py $n >> $O/mega_trace.log 2>&1 || exit 1; done
cat $O/mega_trace.log
echo end | tee -a $O/progress.log
