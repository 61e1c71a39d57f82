#!/bin/bash
# ON THE GPU BOX: round-6 ring A/B -- parity first, then the timing of the quiet in-wavefront links / the binary64 scan
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/ring_ab; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o $O/valu_rates 2> $O/valu_build.err && $O/valu_rates > $O/r06_valu_rates.json 2> $O/valu_run.err
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_prologue.py tests/test_gpu_sharded.py -q -x > $O/tests_ring.log 2>&1; echo "ring tests rc=$?" >> $O/tests_ring.log
timeout 300 python -m pytest tests/test_gpu_fullsize_oracle.py -q -x -k ring > $O/tests_ring_full.log 2>&1; echo "ring full rc=$?" >> $O/tests_ring_full.log
for v in main i64; do
  L=""; [ $v = i64 ] && L=happy_simulator_amd/lib/instr/libhs_i64.so
  HS_HIP_LIB=$L timeout 300 python tools/ring_fullsize.py --repeats 5 > $O/time_$v.log 2>&1
done
timeout 300 python tools/ring_fullsize.py --repeats 5 --flags $((1<<21)) > $O/time_main_loud.log 2>&1
HS_HIP_LIB=happy_simulator_amd/lib/instr/libhs_cycles.so timeout 300 python tools/cycles.py --ring > $O/cycles.log 2>&1
tail -3 $O/tests_ring.log $O/tests_ring_full.log; tail -2 $O/time_*.log; tail -1 $O/cycles.log; head -c 600 $O/r06_valu_rates.json
