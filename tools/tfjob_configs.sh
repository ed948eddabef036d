# The reference's own launch lines (sample_tfjobs/*.yaml: --batch_size 4 --num_batches 30) and a few other batch sizes, through the shims
cd $GRAFT_REPO_ROOT
export TETHYS_WORKSPACE=/tmp/ws TETHYS_RESULT=/tmp/res
mkdir -p /tmp/ws /tmp/res/job; echo job > /tmp/ws/model.txt
python speech_jobs/whisper_dist.py --batch_size 4 --num_batches 30 2>&1 | grep -E "Step (0|29),|jct|Error|Traceback" | cut -c1-120
python speech_jobs/wav2vec2_dist.py --batch_size 4 --num_batches 30 2>&1 | grep -E "Step (0|29),|jct|Error|Traceback" | cut -c1-120
python speech_jobs/wav2vec2_dist.py --batch_size 4 --num_batches 30 --model_size tiny 2>&1 | grep -E "Step (0|29),|jct|Error|Traceback" | cut -c1-120
python speech_jobs/whisper_single.py --batch_size 4 --num_batches 14 2>&1 | grep -E "Step (0|13),|jct|Error|Traceback" | cut -c1-120
python stable_jobs/wav2vec2_dist.py --batch_size 3 --num_batches 18 2>&1 | grep -E "Step (0|17),|jct|Error|Traceback" | cut -c1-120
python speech_jobs/wav2vec2_single.py --batch_size 8 --num_batches 8 --model_size base 2>&1 | grep -E "Step (0|7),|JCT|Error|Traceback" | cut -c1-120
for b in 1 3 16; do python bench.py --batch_size $b --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed|Error|Traceback" | cut -c1-120; done
TMI_WS_GUARD=4096 TMI_WS_POISON=1 python tools/workspace_probe.py whisper small 4 2>&1 | grep -E "overrun|finite" | cut -c1-160 | tail -4
TMI_WS_GUARD=4096 TMI_WS_POISON=1 python tools/workspace_probe.py wav2vec2 small 4 2>&1 | grep -E "overrun|finite" | cut -c1-160 | tail -4
