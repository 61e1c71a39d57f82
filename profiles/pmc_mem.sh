#!/bin/bash
# Run ON THE GPU BOX:  BENCH_ARGS="--workload lb" bash profiles/pmc_mem.sh TAG  -- FETCH_SIZE / WRITE_SIZE per kernel
# (separate rocprofv3 --pmc passes, kernel trace only).
set -u
TAG=${1:-m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps ${STEPS:-3} --warmup 1 --cpu-sample-s 0 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o $TAG -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o $TAG -- $CMD > $OUT/write.log 2>&1
cd $ROOT
for d in fetch write; do
  f=$(ls $OUT/$d/*_results.db 2>/dev/null | head -1)
  [ -n "$f" ] && python profiles/summarize_rocprof.py $f > $OUT/${TAG}_$d.txt 2>&1
done
grep -h "_SIZE" $OUT/${TAG}_fetch.txt $OUT/${TAG}_write.txt | cut -c1-60,200-400
