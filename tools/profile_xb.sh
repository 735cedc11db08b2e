#!/bin/bash
# Kernel trace + FETCH_SIZE / WRITE_SIZE passes of the batched exact scorer (tools/xb_quick.py) on the GPU box; small summaries
# land in gpurun_out/prof_<tag>/.   usage: tools/profile_xb.sh <tag>   (env N, D, B, SEL as for xb_quick.py)
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_xb_kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_xb_kt -- python3 $R/tools/xb_quick.py > $OUT/xb_quick_under_trace.log 2>&1
f=$(find /tmp/rp_xb_kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/rp_xb_$C; rm -rf $d
  REPS=3 rocprofv3 --pmc $C --kernel-include-regex "jvx_(qs|tile)_kernel" --output-format csv -d $d -- python3 $R/tools/xb_quick.py > $OUT/xb_quick_under_pmc_$C.log 2>&1
  c=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$c" ] && cp $c $OUT/pmc_$C.csv
done
ls -la $OUT
