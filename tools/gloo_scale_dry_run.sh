#!/bin/bash
# The driver's 8-GPU scaling run (bench.py --gpus N for N = 1, 2, 4, 8), rehearsed on ONE GPU: every rank shares device 0, the
# all-gather goes through gloo (the host-staged DEBUG backend — a one-GPU box has no RCCL ring) and the corpus is tiny.  What it
# proves: the launcher starts N ranks, every N prints ONE line with per-N rerankK / merged recall / "scaling" / "rccl_ranks",
# and the shard / gather / merge path computes the same merged answers at every N.  What it cannot show: ncclAllGather over
# xGMI and peer copies between distinct devices (never executed anywhere this repository was built).
# usage: tools/gloo_scale_dry_run.sh <outdir> [workload] [docs per GPU (c4) or total (c3)]
OUT=${1:-gpurun_out/gloo_scale}; WL=${2:-c4}; N=${3:-100000}
mkdir -p $OUT
for G in 1 2 4 8; do
  JV_BENCH_BACKEND=gloo JV_BENCH_N=$N JV_BENCH_EXACT_BATCH=0 timeout 900 python bench.py --gpus $G --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-dist-comparison --batch 2048 \
      > $OUT/bench_${WL}_n${G}.json 2> $OUT/bench_${WL}_n${G}.err
  echo "N=$G rc=$? $(cut -c1-400 $OUT/bench_${WL}_n${G}.json)"
done
