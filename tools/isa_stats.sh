#!/bin/bash
# Compiles csrc/grid.hip to gfx950 assembly (/tmp/grid.s) and prints registers / spills / LDS of the kernels matching $1.
cd /root/repo/adhoc-queries-pointclouds_amd/csrc || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I/root/repo/include -I. -S --cuda-device-only -o /tmp/grid.s ${2:-grid.hip} 2>&1 | grep -v hip-link
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|sgpr_spill_count):" /tmp/grid.s | paste - - - - - - | grep -E "${1:-.}" | sed -e 's/\s\+/ /g' -e 's/_ZN12_GLOBAL__N_1//' -e 's/.group_segment_fixed_size/lds/' -e 's/.private_segment_fixed_size/scratch/' | cut -c1-230
