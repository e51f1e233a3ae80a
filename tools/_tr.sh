cd $GRAFT_REPO_ROOT; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/prof_train; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python $R/tools/profile_train_step.py --whole > $R/gpurun_out/train_run.log 2>&1 )
python tools/rocpd_summary.py detail $(find /tmp/prof_train -name "*_results.db" | head -1) 120 > gpurun_out/train_detail.txt 2>&1
grep "wall per step" gpurun_out/train_run.log; head -50 gpurun_out/train_run.log | tail -45 > gpurun_out/train_cprofile.txt
