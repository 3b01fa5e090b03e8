#!/bin/bash
# round 6: developer variants of the four-level scheme passes (positions per workgroup), per-kernel times by rocprofv3 on one box
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06/pass_variants; mkdir -p $O
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
run down32 TAG=down32 M4RI_AMD_SP_DOWN=32
run pack64 TAG=pack64 M4RI_AMD_SP_PACK=64
run up64 TAG=up64 M4RI_AMD_SP_UP=64
run base2 TAG=base2
