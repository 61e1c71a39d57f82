#!/bin/bash
# ON THE GPU BOX: SQ counters per kernel of the LB bench.  usage: WL=ring|lb|grid bash tools/sq_counters.sh TAG pattern
TAG=${1:-x}; PAT=${2:-sources}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/sq -o t -- python $ROOT/bench.py --workload ${WL:-ring} --steps 3 --warmup 1 --cpu-sample-s 0 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_BRANCH -d $OUT/sq2 -o t -- python $ROOT/bench.py --workload ${WL:-ring} --steps 3 --warmup 1 --cpu-sample-s 0 > $OUT/sq2.log 2>&1
cd $ROOT
for d in sq sq2; do
f=$(ls $OUT/$d/*_results.db 2>/dev/null | head -1)
[ -n "$f" ] && python profiles/summarize_rocprof.py $f > $OUT/$d.txt 2>&1
grep -E "$PAT" $OUT/$d.txt | grep "SQ_" | awk -F'|' '{printf "%-40.40s %-28s %14.0f\n", $1, $2, $5}'
rm -rf $OUT/$d
done
tail -3 $OUT/sq2.log | head -c 600
