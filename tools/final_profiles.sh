#!/bin/bash
# ON THE GPU BOX (gpurun): the round's committed evidence -- rocprofv3 kernel trace + the separate PMC passes (FETCH_SIZE, WRITE_SIZE,
# SQ_*) for the three workloads, the derived roofline JSONs bench.py quotes, and the bench lines themselves.
#   usage: bash tools/final_profiles.sh r06        (profiles/HEAD_COMMIT must name the commit of the tree)
R=${1:-r06}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
T0=$(date +%s)
mkdir -p gpurun_out/final
export HS_PROFILE_COMMIT=$(cat profiles/HEAD_COMMIT)
STEPS=6 BENCH_ARGS="--extras 0" bash profiles/collect.sh $R > gpurun_out/collect_$R.log 2>&1
for k in trace fetch write sq; do cp gpurun_out/prof_$R/${R}_$k.txt profiles/; done
cp gpurun_out/prof_$R/${R}_bench_line.json profiles/${R}_bench_line_under_rocprof.json 2>/dev/null
python profiles/derive_roofline.py $R grid "hs_station_run<1, false, true, true>" 2 > gpurun_out/final/derive.log 2>&1
# round 6: the issue cost per VALU instruction class on THIS box (tools/valu_rates.hip) and the kernels' dynamic instruction counts by
# class (one more PMC pass each) -> the class-weighted issue floor bench.py quotes as roofline.valu_floor_frac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o gpurun_out/final/valu_rates 2> gpurun_out/final/valu_build.err && gpurun_out/final/valu_rates > profiles/${R}_valu_rates.json 2> gpurun_out/final/valu_run.err
CLS="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
classes() {   # tag, bench args...
  local tag=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $CLS -d $GRAFT_REPO_ROOT/gpurun_out/prof_cls_$tag -o cls -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample-s 0 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_cls_$tag.log 2>&1)
  local f=$(ls gpurun_out/prof_cls_$tag/*_results.db 2>/dev/null | head -1)
  [ -n "$f" ] && python profiles/summarize_rocprof.py $f > profiles/${R}_valu_classes_$tag.txt 2>&1
  rm -rf gpurun_out/prof_cls_$tag
}
classes grid --extras 0 --api-run 0
classes ring --workload ring
classes wave_8192lp --n-lp 8192 --extras 0 --api-run 0
python profiles/derive_valu_floor.py $R grid "hs_station_run<1, false, true, true>" 2 profiles/${R}_valu_classes_grid.txt profiles/${R}_valu_rates.json >> gpurun_out/final/derive.log 2>&1
python profiles/derive_valu_floor.py $R ring "hs_net_async<1, false, true>" 1 profiles/${R}_valu_classes_ring.txt profiles/${R}_valu_rates.json >> gpurun_out/final/derive.log 2>&1
python profiles/derive_valu_floor.py $R wave8192 "hs_station_wave<16, true>" 8 profiles/${R}_valu_classes_wave_8192lp.txt profiles/${R}_valu_rates.json >> gpurun_out/final/derive.log 2>&1
cp profiles/${R}_valu_* gpurun_out/final/ 2>/dev/null
# the driver's own command (default flags: grid + ring + LB + strong shard in one line), kernel trace only
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${R}default -o ${R}default -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_${R}default.log 2>&1)
f=$(ls gpurun_out/prof_${R}default/*_results.db 2>/dev/null | head -1)
[ -n "$f" ] && python profiles/summarize_rocprof.py $f > profiles/${R}default_trace.txt 2>&1
grep -h '"metric"' gpurun_out/prof_${R}default.log | tail -1 > profiles/${R}default_bench_line_under_rocprof.json
cp profiles/${R}default_trace.txt profiles/${R}default_bench_line_under_rocprof.json gpurun_out/final/ 2>/dev/null
rm -rf gpurun_out/prof_${R}default
echo "grid profile done $(( $(date +%s) - T0 )) s"
STEPS=3 BENCH_ARGS="--workload ring" bash profiles/collect.sh ${R}ring > gpurun_out/collect_${R}ring.log 2>&1
for k in trace fetch write sq; do cp gpurun_out/prof_${R}ring/${R}ring_$k.txt profiles/; done
cp gpurun_out/prof_${R}ring/${R}ring_bench_line.json profiles/${R}ring_bench_line_under_rocprof.json 2>/dev/null
python profiles/derive_roofline.py ${R}ring ring "hs_net_async<1, false, true>" 1 >> gpurun_out/final/derive.log 2>&1
echo "ring profile done $(( $(date +%s) - T0 )) s"
STEPS=3 BENCH_ARGS="--workload lb" bash profiles/collect.sh ${R}lb > gpurun_out/collect_${R}lb.log 2>&1
for k in trace fetch write sq; do cp gpurun_out/prof_${R}lb/${R}lb_$k.txt profiles/; done
cp gpurun_out/prof_${R}lb/${R}lb_bench_line.json profiles/${R}lb_bench_line_under_rocprof.json 2>/dev/null
(cd profiles && python derive_lb_traffic.py ${R}lb) >> gpurun_out/final/derive.log 2>&1
echo "lb profile done $(( $(date +%s) - T0 )) s"
cp profiles/${R}*_roofline_*.json gpurun_out/final/ 2>/dev/null
for k in trace fetch write sq; do cp profiles/${R}_$k.txt profiles/${R}ring_$k.txt profiles/${R}lb_$k.txt gpurun_out/final/ 2>/dev/null; done
cp profiles/${R}*_bench_line_under_rocprof.json gpurun_out/final/ 2>/dev/null
python bench.py --steps 20 --warmup 5 2> gpurun_out/final/bench_default.err | tail -1 > gpurun_out/final/${R}_bench_default.json
python bench.py --workload ring --cpu-sample-s 6 2> gpurun_out/final/bench_ring.err | tail -1 > gpurun_out/final/${R}_bench_ring.json
python bench.py --workload lb --cpu-sample-s 6 2> gpurun_out/final/bench_lb.err | tail -1 > gpurun_out/final/${R}_bench_lb.json
python bench.py --n-lp 8192 --cpu-sample-s 0 --extras 0 --api-run 0 2> gpurun_out/final/bench_8192.err | tail -1 > gpurun_out/final/${R}_bench_8192.json
python bench.py --fake-ranks 2 --cpu-sample-s 0 --steps 5 --warmup 2 2> gpurun_out/final/bench_fake2.err | tail -1 > gpurun_out/final/${R}_bench_fake_ranks_2.json
python bench.py --workload ring --fake-ranks 2 --cpu-sample-s 0 --steps 5 --warmup 2 2> gpurun_out/final/bench_ring_fake2.err | tail -1 > gpurun_out/final/${R}_bench_ring_fake_ranks_2.json
python bench.py --workload ring --fake-ranks 2 --ring-exchange device --cpu-sample-s 0 --steps 3 --warmup 1 2> gpurun_out/final/bench_ring_fake2_rounds.err | tail -1 > gpurun_out/final/${R}_bench_ring_fake_ranks_2_rounds.json
# the strong shard (8 192 LPs: one wavefront per LP) under the kernel trace, the K sweep, and the ring past one cooperative launch
bash tools/trace_cmd.sh ${R}_wave8192 --n-lp 8192 --cpu-sample-s 0 --extras 0 --api-run 0 --steps 20 --warmup 5 > gpurun_out/final/${R}_wave8192.log 2>&1
cp gpurun_out/${R}_wave8192/trace.txt profiles/${R}_trace_wave_8192lp.txt 2>/dev/null
python tools/wide_timing.py --sizes 1024,4096,8192,16384,32768 > gpurun_out/final/${R}_wide_timing.log 2>&1
python tools/ring_fullsize.py > gpurun_out/final/${R}_ring_fullsize.log 2>&1
python tools/ring_fullsize.py --n 131072 >> gpurun_out/final/${R}_ring_fullsize.log 2>&1
python tools/api_profile.py 2>&1 | grep "^pass" > gpurun_out/final/${R}_api_profile.log
echo "bench done $(( $(date +%s) - T0 )) s"
cut -c1-400 gpurun_out/final/${R}_bench_default.json
tail -3 gpurun_out/final/derive.log
# round 6: windows over the ring -- continued from the last state against repeated from the start against one run
python - <<'PY' > profiles/${R}_ring_windows.json 2> gpurun_out/final/ring_windows.err
import json, subprocess, sys
out = {}
for tag, args in (("1000_windows_of_1ms", ["--end-s", "1", "--windows", "1000", "--repeat-windows", "100"]), ("60_windows_of_1s", ["--end-s", "60", "--windows", "60", "--repeat-windows", "60"])):
    r = subprocess.run([sys.executable, "tools/ring_windows.py"] + args, capture_output=True, text=True, timeout=900)
    out[tag] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else dict(error=r.stderr[-500:])
json.dump(dict(what="65 536-station ring (the bench's), run_until with growing ends: wall ms of one run, of W windows continued from the last state, of windows repeated from the start (debug flag 1 << 24)", **out), sys.stdout, indent=1)
PY
cp profiles/${R}_ring_windows.json gpurun_out/final/ 2>/dev/null
echo "windows done $(( $(date +%s) - T0 )) s"
