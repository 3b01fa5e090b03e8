#!/bin/bash
# usage (on the GPU box): bash tools/prof_one.sh <tag> <leaf_check args...>   -> gpurun_out/prof_<tag>.summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/prof_$tag
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $R/build/leaf_check "$@" > $O/trace.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/trace -name "*results.db" | head -1) > $R/gpurun_out/prof_$tag.summary.txt 2>&1
rm -rf $O
cat $R/gpurun_out/prof_$tag.summary.txt
