#!/bin/bash
# FETCH_SIZE of the pqw kernel on the pqw_quick workload (2M docs, rk 1200)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/rp_f
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "jv_search_pqw_kernel" --output-format csv -d /tmp/rp_f -- python3 $R/tools/pqw_quick.py > /tmp/pq.log 2>&1
tail -1 /tmp/pq.log
c=$(find /tmp/rp_f -name "*counter_collection.csv" | head -1)
python3 - <<PY
import csv
v=[float(r["Counter_Value"]) for r in csv.DictReader(open("$c")) if "pqw" in r["Kernel_Name"]]
print("FETCH_SIZE KiB per launch:", [round(x) for x in v], "-> GB x2:", [round(x*1024*2/1e9,1) for x in v])
PY
