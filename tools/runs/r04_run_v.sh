set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_v; mkdir -p $O; rm -f $O/*
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dist.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $O/progress.log; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
GPCORE_BENCH_DEVICE=0 GPCORE_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/c2_c3_2rank_selflaunch_gloo.json 2> $O/c2_c3_2rank_selflaunch_gloo.err || { tail -20 $O/c2_c3_2rank_selflaunch_gloo.err; exit 1; }
echo "2-rank rehearsal done" | tee -a $O/progress.log
tail -c 1500 $O/c2_c3_2rank_selflaunch_gloo.json; echo; tail -5 $O/c2_c3_2rank_selflaunch_gloo.err
