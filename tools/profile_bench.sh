#!/bin/bash
# Runs bench.py under rocprofv3 on the GPU box and leaves only small summaries in gpurun_out/prof_<tag>/:
#   kernel_stats.csv        rocprofv3 --kernel-trace --stats summary (all kernels of the run)
#   jv_kernel_trace.csv     the per-dispatch rows of this repo's kernels only
#   pmc_fetch.csv / pmc_write.csv   FETCH_SIZE / WRITE_SIZE per dispatch of jv_search kernels (separate passes)
# usage: tools/profile_bench.sh <tag> <bench args...>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_kt /tmp/rp_f /tmp/rp_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_kt -- python3 $R/bench.py "$@" --profile-mode > $OUT/bench_under_trace.json 2> $OUT/bench_under_trace.err
f=$(find /tmp/rp_kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
t=$(find /tmp/rp_kt -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then head -1 $t > $OUT/jv_kernel_trace.csv; grep -E "jv_" $t | grep -v builder >> $OUT/jv_kernel_trace.csv; fi
for C in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/rp_$C; rm -rf $d
  rocprofv3 --pmc $C --kernel-include-regex "jv_(search_(lds|pqf|pqp|pqw)_kernel|visited)" --output-format csv -d $d -- python3 $R/bench.py "$@" --profile-mode > $OUT/bench_under_pmc_$C.json 2> $OUT/bench_under_pmc_$C.err
  c=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$c" ] && cp $c $OUT/pmc_$C.csv
done
# SQ instruction / wait counters of the main kernels (is the kernel vector-issue bound?), one pass per group (SKIP_SQ=1: not this time)
[ "${SKIP_SQ:-0}" = "1" ] && { ls -la $OUT; exit 0; }
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); d=/tmp/rp_sq$i; rm -rf $d
  rocprofv3 --pmc $SET --kernel-include-regex "jv_search_(lds|pqf|pqp|pqw)_kernel" --output-format csv -d $d -- python3 $R/bench.py "$@" --profile-mode > $OUT/bench_under_pmc_sq$i.json 2> $OUT/bench_under_pmc_sq$i.err
  c=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$c" ] && cp $c $OUT/pmc_sq$i.csv
done
ls -la $OUT
