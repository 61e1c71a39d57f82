#!/bin/bash
# ON THE GPU BOX: kernel-trace summary of a bench command.  usage: bash tools/trace_cmd.sh TAG <bench.py args...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $ROOT/bench.py "$@" > $OUT/trace.log 2>&1
cd $ROOT
f=$(ls $OUT/trace/*_results.db 2>/dev/null | head -1)
[ -n "$f" ] && python profiles/summarize_rocprof.py $f > $OUT/trace.txt 2>&1
echo "== $TAG: $@"
grep -v "^#" $OUT/trace.txt | awk -F'|' 'NF>=6 && $2+0>0 && $2+0<1000 {n=$1; gsub(/\(anonymous namespace\)::/,"",n); gsub(/hs::/,"",n); printf "%-70.70s calls %4d avg %9.1f min %9.1f\n", n, $2, $4, $5}' | head -16
grep -h '"metric"' $OUT/trace.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms_avg'))"
rm -rf $OUT/trace
