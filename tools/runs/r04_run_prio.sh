set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_prio; mkdir -p $O; rm -f $O/*
cd $R
export GPCORE_LIB_PATH=$R/tools/lab/libgpcore_prio.so
for cfg in "GPCORE_X=0" "GPCORE_SIDE_PRIORITY=1" "GPCORE_FAR_LOW_PRIORITY=1" "GPCORE_SIDE_PRIORITY=1 GPCORE_FAR_LOW_PRIORITY=1" "GPCORE_SIDE_PRIORITY=0" "GPCORE_X=1"; do
  echo "== $cfg" >> $O/ep_priority.log
  timeout -k 10 120 env $cfg python tools/ep_sweeps.py 4096 50 >> $O/ep_priority.log 2>&1 || { echo "FAILED rc=$?" >> $O/ep_priority.log; cat $O/ep_priority.log; exit 1; }
done
# the same switches on the launch-per-step Cholesky's look-ahead (the side stream carries its far updates)
for cfg in "GPCORE_X=0" "GPCORE_SIDE_PRIORITY=1" "GPCORE_SIDE_PRIORITY=0"; do
  echo "== $cfg" >> $O/ep_priority.log
  timeout -k 10 120 env GPCORE_CHOL_MEGA=0 $cfg python tools/fit_only.py 8192 10 >> $O/ep_priority.log 2>&1 || { echo "FAILED rc=$?" >> $O/ep_priority.log; cat $O/ep_priority.log; exit 1; }
done
cat $O/ep_priority.log
