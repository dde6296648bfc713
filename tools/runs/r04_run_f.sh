set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_f; mkdir -p $O
cd $R
for mode in 0 1 2 4 3 7; do
  echo "== mode $mode" | tee -a $O/mega_debug.log
  GPCORE_MEGA_MODE=$mode timeout -k 5 40 python tools/mega_debug.py 2048 >> $O/mega_debug.log 2>&1 || { echo "mode $mode FAILED rc=$?" | tee -a $O/mega_debug.log; break; }
  tail -1 $O/mega_debug.log
done
echo end | tee -a $O/mega_debug.log
