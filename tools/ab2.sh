set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TMI_ADAM_UNDER_BACKWARD=0 bash tools/profile_one.sh adam0 7 --steps 4 --warmup 3 > /dev/null
TMI_ADAM_UNDER_BACKWARD=1 TMI_ADAM_OVERLAP_BLOCKS=128 bash tools/profile_one.sh adam1 7 --steps 4 --warmup 3 > /dev/null
head -12 gpurun_out/prof_adam0_gaps.txt; head -12 gpurun_out/prof_adam1_gaps.txt
python tools/host_step_time.py
