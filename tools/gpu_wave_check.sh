#!/bin/bash
# ON THE GPU BOX: the wide / wave kernel's tests + step times (+ optional instrumented libraries).  usage: bash tools/gpu_wave_check.sh TAG [sizes]
TAG=${1:-x}; SIZES=${2:-1024,8192}
O=gpurun_out/$TAG; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_wide.py -x -q > $O/wide_tests.log 2>&1; echo "wide rc $?"
timeout 200 python tools/wide_timing.py --sizes $SIZES > $O/wide_timing.log 2>&1
for lib in happy_simulator_amd/lib/instr/libhs_*.so; do
  [ -f "$lib" ] || continue
  b=$(basename $lib .so)
  if [[ $b == *cyc* ]]; then HS_HIP_LIB=$lib timeout 200 python tools/wide_timing.py --sizes 8192 --wave-cycles > $O/$b.log 2>&1
  else HS_HIP_LIB=$lib timeout 200 python tools/wide_timing.py --sizes 8192 > $O/$b.log 2>&1; fi
done
tail -4 $O/wide_tests.log; for f in $O/wide_timing.log $O/libhs_*.log; do echo "== $f"; grep -v amdgpu.ids $f; done
