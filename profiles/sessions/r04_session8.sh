#!/bin/bash
# round 4, GPU session 8: four-level passes with nontemporal stores (L2 kept for the siblings' re-reads?) + FETCH/WRITE counters of the
# depth-4 schedule; the world-size-1 distributed runs (RCCL and peer) taken apart by kernel; the round's bench profile set
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for nt in 2 3 7; do
  echo "M4RI_AMD_PASS_NT=$nt" >> $O/s8_depth4_nt.log
  M4RI_AMD_PASS_NT=$nt timeout 300 python tools/prof_product.py 65536 65536 65536 8 4096 4 >> $O/s8_depth4_nt.log 2>&1
done
grep -v amdgpu.ids $O/s8_depth4_nt.log
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c GRBM_GUI_ACTIVE -d $R/$O/pmc8 -o p -- python $R/tools/prof_product.py 65536 65536 65536 1 4096 4 > $R/$O/s8_pmc_$c.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/$O/pmc8 -name "*results.db" | head -1) > $R/$O/s8_depth4_fuse4_pmc_$c.summary.txt 2>&1; rm -rf $R/$O/pmc8
  grep -i "winograd\|rowwise" $R/$O/s8_depth4_fuse4_pmc_$c.summary.txt | head -6
done
# world size 1 through the distributed code paths, by kernel
rocprofv3 --kernel-trace --stats -d $R/$O/tr8a -o t -- python $R/bench.py --gpus 1 --force-dist --variant strassen --steps 5 --warmup 2 --no-cpu-baseline --no-verify > $R/$O/s8_ws1_rccl_strassen.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/$O/tr8a -name "*results.db" | head -1) > $R/$O/s8_ws1_rccl_strassen.summary.txt 2>&1; rm -rf $R/$O/tr8a
rocprofv3 --kernel-trace --stats -d $R/$O/tr8b -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-api --no-traffic > $R/$O/s8_n1.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/$O/tr8b -name "*results.db" | head -1) > $R/$O/s8_n1.summary.txt 2>&1; rm -rf $R/$O/tr8b
head -14 $R/$O/s8_ws1_rccl_strassen.summary.txt; head -8 $R/$O/s8_n1.summary.txt
grep -h '"metric"' $R/$O/s8_ws1_rccl_strassen.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ws1 rccl strassen', d['ms_per_step'], d['host_issue_ms_per_step'])"
grep -h '"metric"' $R/$O/s8_n1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n1', d['ms_per_step'], d['host_issue_ms_per_step'])"
cd $R
# host issue time at 8 ranks: peer transport on virtual ranks, rccl path under gloo
timeout 900 python bench.py --gpus 8 --transport peer --virtual-ranks --steps 5 --warmup 2 --no-cpu-baseline > $O/s8_bench_peer8_virtual.json 2> $O/s8_bench_peer8_virtual.err
timeout 1500 python bench.py --gpus 8 --transport rccl --backend gloo --steps 3 --warmup 1 --no-cpu-baseline > $O/s8_bench_gloo8.json 2> $O/s8_bench_gloo8.err
for f in $O/s8_bench_peer8_virtual.json $O/s8_bench_gloo8.json; do python -c "import sys,json; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', d.get('ms_per_step'), d.get('host_issue_ms_per_step'), d.get('error'))"; done
bash tools/prof_bench.sh r04 > $O/s8_prof_bench.log 2>&1
tail -3 $O/s8_prof_bench.log
