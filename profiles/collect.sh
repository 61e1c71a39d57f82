#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash profiles/collect.sh r01
# Three separate rocprofv3 runs of the same bench command (counters never share a run with tracing domains
# other than --kernel-trace): kernel trace + stats, PMC FETCH_SIZE, PMC WRITE_SIZE.  Raw output goes to
# gpurun_out/ (scratch); profiles/summarize_rocprof.py turns it into the committed text summaries.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps ${STEPS:-10} --warmup 2 --cpu-sample-s 0 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o $TAG -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o $TAG -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/sq -o $TAG -- $CMD > $OUT/sq.log 2>&1
cd $ROOT
for d in trace fetch write sq; do
  f=$(ls $OUT/$d/*_results.db 2>/dev/null | head -1)
  [ -n "$f" ] && python profiles/summarize_rocprof.py $f > $OUT/${TAG}_$d.txt 2>&1
done
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/${TAG}_bench_line.json
ls -la $OUT
