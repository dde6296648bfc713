set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_x; mkdir -p $O; rm -f $O/*
cd $R
for n in 8192 6400 5248 12288; do
  timeout -k 10 400 python tools/mega_soak.py $n 3000 >> $O/mega_soak.log 2>&1 || { echo "FAILED n=$n rc=$?" >> $O/mega_soak.log; tail -20 $O/mega_soak.log; exit 1; }
  tail -1 $O/mega_soak.log
done
echo end | tee -a $O/progress.log
