#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/call4; mkdir -p $O
for v in main nolt main2 nolt2; do
  L=""; [[ $v == nolt* ]] && L=happy_simulator_amd/lib/instr/libhs_nolt.so
  HS_HIP_LIB=$L timeout 300 python tools/ring_fullsize.py --repeats 5 > $O/time_$v.log 2>&1
done
timeout 600 python -m pytest tests/test_gpu_api.py -q -x -k "partition or linked or parallel" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
# dynamic VALU class counts of the headline kernel (grid) and of the strong shard's wave kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $R/$O/cls_grid -o cls -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample-s 0 --extras 0 --api-run 0 > $R/$O/cls_grid.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F32 -d $R/$O/cls_grid2 -o cls -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample-s 0 --extras 0 --api-run 0 > $R/$O/cls_grid2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $R/$O/cls_wave -o cls -- python $R/bench.py --n-lp 8192 --steps 3 --warmup 1 --cpu-sample-s 0 --extras 0 --api-run 0 > $R/$O/cls_wave.log 2>&1
cd $R
for d in cls_grid cls_grid2 cls_wave; do
  f=$(ls $O/$d/*_results.db 2>/dev/null | head -1)
  [ -n "$f" ] && python profiles/summarize_rocprof.py $f > $O/$d.txt 2>&1
  rm -rf $O/$d
done
tail -n 2 $O/time_*.log; tail -n 3 $O/tests.log; grep -h "hs_station" $O/cls_grid.txt | grep SQ_ | cut -c1-60,200-400 | head -20
