#!/bin/bash
# which class of box is this?  (65536^3 resident: fast class 26.3 ... 26.6 ms, slow class 29.0 ... 29.5 ms.)  On a slow one: the depth sweep
# of the time model (review item 8), and the bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
TAG=probe python tools/time_product.py 65536 65536 65536 10 5 2>&1 | grep -v amdgpu.ids | tee $O/box_speed_probe.log
ms=$(awk '{for(i=1;i<=NF;i++) if ($i=="ms/product,") print $(i-1)}' $O/box_speed_probe.log | head -1)
if python -c "import sys; sys.exit(0 if float('$ms') > 28.0 else 1)"; then
  python tools/depth_model_sweep.py 65536,65536,65536 32768,32768,32768 16384,16384,16384 131072,8192,131072 16384,8192,131072 2>&1 | grep -v amdgpu.ids | tee $O/depth_model_slow_box.log
  python bench.py --no-cpu-baseline > $O/bench65536_slow_box.json 2>/dev/null; cut -c1-400 $O/bench65536_slow_box.json
else
  echo "fast class: nothing to do"
fi
