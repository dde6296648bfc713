set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_r; mkdir -p $O; rm -f $O/*
cd $R
for costs in "25.3,21.7,28.4,81.0,9.0,30.0,20.0" "27.5,20.8,25.1,77.7,9.0,27.8,30.0" "27.5,20.8,25.1,77.7,9.0,27.8,45.0" "27.5,24,29,85,9.0,32,30.0" "27.5,18,22,70,9.0,25,30.0" "40,20.8,25.1,77.7,9.0,27.8,45.0"; do
  for n in 6144 8192 10240; do
    echo "== GPCORE_MEGA_COSTS=$costs" >> $O/fit.log; timeout -k 10 120 env GPCORE_CHOL_MEGA=1 GPCORE_MEGA_COSTS=$costs python tools/fit_only.py $n 10 >> $O/fit.log 2>&1 || { echo "FAILED rc=$?" >> $O/fit.log; cat $O/fit.log; exit 1; }
  done
done
echo "fit done" | tee -a $O/progress.log; cat $O/fit.log
