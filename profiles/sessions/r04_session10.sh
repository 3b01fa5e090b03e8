#!/bin/bash
# round 4, GPU session 10: the multi-device C path with one copy stream per peer (tests + bench on virtual ranks), small products, crossover
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dmat.py tests/test_gpu_threads.py tests/test_small_products.py -x -q -m gpu > $O/s10_pytest_dmat_threads_small.log 2>&1
tail -4 $O/s10_pytest_dmat_threads_small.log
timeout 2400 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > $O/s10_pytest_multi.log 2>&1
tail -4 $O/s10_pytest_multi.log
timeout 900 python bench.py --gpus 8 --transport peer --virtual-ranks --steps 5 --warmup 2 --no-cpu-baseline > $O/s10_bench_peer8_virtual.json 2> $O/s10_bench_peer8_virtual.err
timeout 900 python bench.py --gpus 8 --transport peer --virtual-ranks --workload rect131072 --steps 3 --warmup 1 --no-cpu-baseline > $O/s10_bench_peer8_virtual_rect.json 2> $O/s10_bench_peer8_virtual_rect.err
timeout 900 python bench.py --gpus 2 --transport peer --virtual-ranks --steps 3 --warmup 1 --no-cpu-baseline > $O/s10_bench_peer2_virtual.json 2> $O/s10_bench_peer2_virtual.err
for f in $O/s10_bench_peer8_virtual.json $O/s10_bench_peer8_virtual_rect.json $O/s10_bench_peer2_virtual.json; do python -c "
import json
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); c=d.get('config',{})
print('$f', d.get('ms_per_step'), d.get('host_issue_ms_per_step'), c.get('schedule_stats'), (d.get('verified') or {}).get('matches_reference'), d.get('error'))
print('   rank0 timeline', c.get('timeline_ms_last_step',{}).get('0'))"; done
timeout 900 python tests/crossover_cpu_gpu.py > $O/s10_crossover_cpu_gpu.log 2>&1
head -20 $O/s10_crossover_cpu_gpu.log
