set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_p; mkdir -p $O; rm -f $O/*
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $O/progress.log; tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit 1
for n in 4096 6144 8192 10240 12288 16384; do
  for cfg in "GPCORE_CHOL_MEGA=0" "GPCORE_CHOL_MEGA=1"; do
    echo "== $cfg" >> $O/fit.log; timeout -k 10 120 env $cfg python tools/fit_only.py $n 10 >> $O/fit.log 2>&1 || { echo "FAILED rc=$?" >> $O/fit.log; cat $O/fit.log; exit 1; }
  done
done
echo "fit done" | tee -a $O/progress.log; cat $O/fit.log
timeout -k 10 120 python tools/mega_trace.py 8192 >> $O/mega_trace.log 2>&1 || exit 1
head -12 $O/mega_trace.log
timeout -k 10 300 python bench.py --workload c3 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err || exit 1
echo "c3 done" | tee -a $O/progress.log
timeout -k 10 300 python bench.py --workload c4 --steps 2 > $O/bench_c4.json 2> $O/bench_c4.err || exit 1
echo "c4 done" | tee -a $O/progress.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err || exit 1
python - <<'P'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r04_p/"
for w in ("c2","c3","c4"):
    j=json.loads(open(O+"bench_%s.json"%w).read().strip().splitlines()[-1])
    print(w, j["value"], j["unit"], "ms/step", j.get("ms_per_step"), "roofline frac", j["roofline"]["frac"] if j.get("roofline") else None)
    if w=="c2": print("   cholesky", {k:(round(v,3) if isinstance(v,float) else v) for k,v in j["cholesky"].items() if not isinstance(v,str)}); print("   c3_sharded", j.get("c3_sharded",{}).get("value"))
    if w=="c4": print("   ep_grid", j.get("ep_grid"))
P
echo end | tee -a $O/progress.log
