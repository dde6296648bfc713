set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_d; mkdir -p $O
cd $R
run() { echo "== $*" >> $O/c3_sweep.log; env "$@" python bench.py --workload c3 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print('   settings/s %.1f' % d['value'])" >> $O/c3_sweep.log 2>&1; tail -1 $O/c3_sweep.log; }
run GPCORE_X=0
run GPCORE_LML_GROUP=16
run GPCORE_LML_GROUP=16 GPCORE_LML_WORKERS=4
run GPCORE_LML_GROUP=22 GPCORE_LML_WORKERS=3
run GPCORE_LML_GROUP=32 GPCORE_LML_WORKERS=1
run GPCORE_OUTER=256
run GPCORE_OUTER=1024
run GPCORE_TINV_OUTER=256
run GPCORE_TINV_OUTER=1024
run GPCORE_OUTER=1024 GPCORE_TINV_OUTER=1024
run GPCORE_RESERVED_CUS=0
echo "c3 sweep done" | tee -a $O/progress.log
cat $O/c3_sweep.log
