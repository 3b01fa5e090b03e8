#!/bin/bash
# round 4, GPU session 11: the peer transport's timeline on one GPU against the number of link streams and hardware queues
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
for cfg in "0 4" "1 4" "2 4" "0 8" "0 16" "1 16"; do
  set -- $cfg
  M4RI_AMD_LINK_STREAMS=$1 GPU_MAX_HW_QUEUES=$2 timeout 600 python bench.py --gpus 8 --transport peer --virtual-ranks --steps 4 --warmup 2 --no-cpu-baseline --no-verify > $O/s11_peer8_links$1_hwq$2.json 2> $O/s11_peer8_links$1_hwq$2.err
  python - <<PY
import json
d=json.loads([l for l in open("$O/s11_peer8_links$1_hwq$2.json") if l.startswith("{")][-1])
tl=d["config"]["timeline_ms_last_step"]
print("link streams cap $1, GPU_MAX_HW_QUEUES $2:", round(d["ms_per_step"],2), "ms/step, host issue", round(d["host_issue_ms_per_step"],2))
for r in ("0","1","2","3"): print("    rank", r, [round(x,2) for x in tl[r]])
PY
done
