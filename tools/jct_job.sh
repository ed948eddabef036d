# JCT of the reference's sample job (sample_tfjobs/whisper-dist.yaml:24: --batch_size 4 --num_batches 30) through the shim, twice
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  rm -rf /tmp/ws /tmp/res; mkdir -p /tmp/ws /tmp/res/job; echo job > /tmp/ws/model.txt
  TETHYS_WORKSPACE=/tmp/ws TETHYS_RESULT=/tmp/res python speech_jobs/whisper_dist.py --batch_size 4 --num_batches 30 2>/dev/null | grep -E "^jct|Step 29"
  cat /tmp/res/job/*_jct.txt; echo; ls -la /tmp/ws/checkpoints | tail -1
done
python tools/jct_breakdown.py 2>/dev/null
