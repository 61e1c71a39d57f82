#!/bin/bash
O=$PWD/gpurun_out/call21; mkdir -p $O
timeout 1500 python tools/gpu_random_sweep.py --first 80000 --count 1000 --seconds 1400 --families ring_async,ring_windowed,jitter_ring_async,jitter_ring_windowed,multi_source_ring_async,multi_source_ring_windowed,ring_windows_async,ring_windows_windowed,jitter_ring_windows_async,multi_source_ring_windows_async > $O/sweep.log 2>&1; echo rc=$? >> $O/sweep.log
tail -n 16 $O/sweep.log
