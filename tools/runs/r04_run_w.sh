set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_w; mkdir -p $O; rm -f $O/*
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $O/progress.log; tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit 1
bash tools/collect_evidence.sh r04_final quick 1 || exit 1
tail -4 $R/gpurun_out/r04_final/progress.log
