#!/bin/bash
# Device ISA of one instantiation group (csrc/hs_kernels.hpp HS_INST_GROUP_k) for inspection: resources and the assembly.
#   usage: bash tools/isa.sh 15 [mangled-name-substring]     -> scratch/isa/inst<k>.s (+ kernel.s for the named kernel)
set -e
K=${1:-15}; PAT=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/scratch/isa && cd $ROOT/scratch/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-value -DHS_INST=$K ${EXTRA:-} \
    -c $ROOT/happy_simulator_amd/csrc/hs_inst.hip -o inst$K.o --save-temps 2>&1 | grep -E "error" -A5 || true
cp hs_inst-hip-amdgcn-amd-amdhsa-gfx950.s inst$K.s
grep -E "^\s+\.(name|vgpr_count|sgpr_count|sgpr_spill_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):" inst$K.s
if [ -n "$PAT" ]; then
  awk -v pat="$PAT" '$0 ~ "^_Z.*"pat".*:" {on=1} on {print} on && /s_endpgm/ {exit}' inst$K.s > kernel.s
  wc -l kernel.s
fi
