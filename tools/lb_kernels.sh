#!/bin/bash
# ON THE GPU BOX: per-kernel times of the LB bench for a (variant) library.  usage: HS_HIP_LIB=... bash tools/lb_kernels.sh TAG [pattern]
TAG=${1:-x}; PAT=${2:-.}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $ROOT/bench.py --workload lb --steps 7 --warmup 2 --cpu-sample-s 0 > $OUT/trace.log 2>&1
cd $ROOT
f=$(ls $OUT/trace/*_results.db 2>/dev/null | head -1)
[ -n "$f" ] && python profiles/summarize_rocprof.py $f > $OUT/trace.txt 2>&1
echo "== $TAG"
grep -v "^#" $OUT/trace.txt | awk -F'|' 'NF>=6 && $2+0>0 && $2+0<100 {n=$1; gsub(/\(anonymous namespace\)::/,"",n); printf "%-60.60s calls %3d avg %8.1f\n", n, $2, $4}' | grep -E "$PAT" | head -24
grep -h '"metric"' $OUT/trace.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'device', d['config']['device_ms_per_step'])"
rm -rf $OUT/trace
