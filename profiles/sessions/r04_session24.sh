#!/bin/bash
# round 4, GPU session 24: the A-side pack pass without the transpose as the default -- its new parity cases, the passes alone, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_level or three_level or 65536" > $O/s24_pytest.log 2>&1
tail -15 $O/s24_pytest.log
timeout 300 python tools/passes_only.py 65536 10 > $O/s24_passes.log 2>&1
M4RI_AMD_DOWN4_PACK=transpose timeout 300 python tools/passes_only.py 65536 10 >> $O/s24_passes.log 2>&1
tail -12 $O/s24_passes.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/s24_bench.json 2> $O/s24_bench.err
head -c 300 $O/s24_bench.json; echo
