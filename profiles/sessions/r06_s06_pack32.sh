#!/bin/bash
# round 6, session 2: the packed-A scheme pass with a workgroup per DWORD column x 64 rows (256-byte runs) against the one per word
# column x 32 rows (two 128-byte runs); per-kernel times by rocprofv3 on one box, checksums must agree
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06/pack32; mkdir -p $O
run() {  # name, env...
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats -d $O/$name -o t -- python $R/tools/time_product.py 65536 65536 65536 6 3 > $O/$name.log 2>&1
  f=$(find $O/$name -name "*results.db" | head -1)
  python $R/tools/rocpd_summary.py $f > $O/$name.summary.txt 2>&1
  rm -rf $O/$name
  echo "== $name: $(grep ms/product $O/$name.log)"
  grep -E "scheme_|m4rm8q" $O/$name.summary.txt | awk '{printf "   %-60s calls %s avg %s ms\n", substr($1,1,60), $2, $4}'
}
run base TAG=base
run pack32x4 TAG=p4 M4RI_AMD_PACK32=4
run pack32x8 TAG=p8 M4RI_AMD_PACK32=8
run base2 TAG=base2
run pack32x4b TAG=p4b M4RI_AMD_PACK32=4
