# A/B of environment settings on ONE box, alternating, three rounds.  Usage (on the GPU box, from the repo root):
#   bash tools/ab_env.sh [whisper|wav2vec2] "TMI_ADAM_LATE=0" "TMI_ADAM_LATE=1" ["TMI_ADAM_LATE=1 TMI_ADAM_LATE_BLOCKS=256" ...]
# Every round-3 scheduling change was accepted or dropped on such a run: TMI_GEMM_NO_KGROUPS, TMI_GEMM_WALK_M, TMI_WGRAD_CHUNKS,
# TMI_CONV_WGRAD_SIDE, TMI_ADAM_LATE / _BLOCKS / _LAYER (DESIGN (f), "What changed the step").  Two library builds: tools/ab_lib.sh.
cd $GRAFT_REPO_ROOT
W=whisper; STEPS=100
if [ "$1" = whisper ] || [ "$1" = wav2vec2 ]; then W=$1; shift; fi
[ $W = wav2vec2 ] && STEPS=200
for round in 1 2 3; do
  for setting in "$@"; do
    env $setting python bench.py --workload $W --steps $STEPS --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 \
      | python -c "import sys,json; print('$W [$setting]', round(json.loads(sys.stdin.read())['ms_per_step'],3))"
  done
done
