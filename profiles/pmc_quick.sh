#!/bin/bash
# Run ON THE GPU BOX: bash profiles/pmc_quick.sh TAG  -- SQ instruction-mix counters for the bench kernel.
set -u
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 4 --warmup 1 --cpu-sample-s 0 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/sq -o $TAG -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU -d $OUT/sq2 -o $TAG -- $CMD > $OUT/sq2.log 2>&1
cd $ROOT
for d in sq sq2; do
  f=$(ls $OUT/$d/*_results.db 2>/dev/null | head -1)
  [ -n "$f" ] && python profiles/summarize_rocprof.py $f 2>&1 | grep -E "hs_station_run|hs_net_window|hs_" | grep -v reset > $OUT/${TAG}_$d.txt
done
cat $OUT/${TAG}_sq.txt $OUT/${TAG}_sq2.txt
