#!/bin/bash
# ON THE GPU BOX: PartitionLink.packet_loss parity + link tests, the VALU rate table, the ring's PMC traffic passes
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/call3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_ring.py tests/test_gpu_sharded.py tests/test_gpu_dist.py tests/test_gpu_prologue.py -q -x --durations=8 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o $O/valu_rates 2> $O/valu_build.err && $O/valu_rates > $O/r06_valu_rates.json 2> $O/valu_run.err
export HS_PROFILE_COMMIT=wip
R=r06w
STEPS=3 BENCH_ARGS="--workload ring" bash profiles/collect.sh ${R}ring > $O/collect_ring.log 2>&1
for k in trace fetch write sq; do cp gpurun_out/prof_${R}ring/${R}ring_$k.txt $O/; done
for k in trace fetch write sq; do cp gpurun_out/prof_${R}ring/${R}ring_$k.txt profiles/; done
python profiles/derive_roofline.py ${R}ring ring "hs_net_async<1, false, true>" 1 > $O/derive.log 2>&1
cp profiles/${R}ring_roofline_ring.json $O/ 2>/dev/null
rm -rf gpurun_out/prof_${R}ring/*/   # (raw rocprof databases: scratch)
tail -n 5 $O/tests.log; cat $O/derive.log | tail -n 5; cat $O/${R}ring_roofline_ring.json
