set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_y; mkdir -p $O; rm -f $O/*
cd $R
for n in 8192 6400; do
  timeout -k 10 500 python tools/mega_soak.py $n 1500 1 >> $O/mega_soak_contended.log 2>&1 || { echo "FAILED n=$n rc=$?" >> $O/mega_soak_contended.log; tail -20 $O/mega_soak_contended.log; exit 1; }
  tail -1 $O/mega_soak_contended.log
done
echo end | tee -a $O/progress.log
