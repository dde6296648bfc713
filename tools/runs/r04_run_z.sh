set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_z; mkdir -p $O; rm -f $O/*
cd $R
for cfg in "GPCORE_X=0" "GPCORE_RESERVED_CUS_EP=48" "GPCORE_RESERVED_CUS_EP=56" "GPCORE_RESERVED_CUS_EP=72" "GPCORE_RESERVED_CUS_EP=80" "GPCORE_RESERVED_CUS_EP=96" "GPCORE_RESERVED_CUS=16" "GPCORE_RESERVED_CUS=48" "GPCORE_RESERVED_CUS=64 GPCORE_RESERVED_CUS_EP=96" "GPCORE_EP_SIG_K=1" "GPCORE_EP_SIG_K=4" "GPCORE_X=1"; do
  echo "== $cfg" >> $O/ep_knobs.log
  timeout -k 10 120 env $cfg python tools/ep_sweeps.py 4096 50 >> $O/ep_knobs.log 2>&1 || { echo "FAILED rc=$?" >> $O/ep_knobs.log; cat $O/ep_knobs.log; exit 1; }
done
cat $O/ep_knobs.log
echo end | tee -a $O/progress.log
