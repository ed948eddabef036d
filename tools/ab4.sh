set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/host_step_time_w2v.py
bash tools/profile_one.sh w2v 7 --workload wavvec2fix --steps 4 --warmup 3 > /dev/null 2>&1 || true
TMI_WGRAD_STREAM=0 bash tools/profile_one.sh w2vser 7 --workload wav2vec2 --steps 4 --warmup 3 > /dev/null
bash tools/profile_one.sh w2vov 7 --workload wav2vec2 --steps 4 --warmup 3 > /dev/null
head -45 gpurun_out/prof_w2vser_summary.txt | cut -c1-170
head -8 gpurun_out/prof_w2vov_gaps.txt
