#!/bin/bash
# Counterpart of the reference's gpu.sh:7 — start the 6 Hz utilisation sampler in the background,
# writing <result>/<model>/<job>_gpu.txt.  Roots follow the job shims (TETHYS_WORKSPACE / TETHYS_RESULT).
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
WS="${TETHYS_WORKSPACE:-/workspace}"
RES="${TETHYS_RESULT:-/result}"
MODEL="$(cat "$WS/model.txt" 2>/dev/null || echo model)"
if [ -f "$WS/job_name.py" ]; then JOB="$(python3 "$WS/job_name.py")"; else JOB="${TETHYS_JOB:-job}"; fi
mkdir -p "$RES/$MODEL"
SAMPLER="$HERE/tools/gpu_sampler/amdsmi_sampler"
[ -x "$SAMPLER" ] || make -C "$HERE/tools/gpu_sampler" >/dev/null
"$SAMPLER" > "$RES/$MODEL/${JOB}_gpu.txt" &
