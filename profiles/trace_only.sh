#!/bin/bash
# Run ON THE GPU BOX:  BENCH_ARGS="--workload lb" bash profiles/trace_only.sh TAG  -- kernel trace + stats of one bench run.
set -u
TAG=${1:-t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps ${STEPS:-5} --warmup 1 --cpu-sample-s 0 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- $CMD > $OUT/trace.log 2>&1
cd $ROOT
f=$(ls $OUT/trace/*_results.db 2>/dev/null | head -1)
[ -n "$f" ] && python profiles/summarize_rocprof.py $f > $OUT/${TAG}_trace.txt 2>&1
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/${TAG}_bench_line.json
cat $OUT/${TAG}_trace.txt | head -40
