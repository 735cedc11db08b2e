#!/bin/bash
# SQ instruction / wait counters of the persistent pool kernels on the pqw_quick workload (development aid).
# usage: tools/pmc_quick.sh <tag>   (env of tools/pqw_quick.py applies: RKS, N, B, JV_OPT_*)
set -u
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1)); d=/tmp/rp_q$i; rm -rf $d
  rocprofv3 --pmc $SET --kernel-include-regex "jv_search_pq[pw]_kernel" --output-format csv -d $d -- python3 $R/tools/pqw_quick.py > $OUT/run$i.log 2> $OUT/run$i.err
  c=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$c" ] && cp $c $OUT/pmc$i.csv
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in sorted(glob.glob("$OUT/pmc*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS"):
            cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        n = max(cnt.get((k, "SQ_WAVES"), 1), 1)
        print(f"   {c:24s} {v:.4g}")
PY
