#!/bin/bash
# Register / scratch / occupancy table of every kernel in one .hip source (cross-compiles without a GPU):
#   tools/kernel_resources.sh dalm_amd/csrc/lora2.hip
src="$1"; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o /tmp/_kr.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
  | grep -E "error|Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" \
  | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' -e 's/Function Name: _ZN4dalm12_GLOBAL__N_1[0-9]*//' \
  | awk '/^ *[a-z0-9_]+I/ || /kernel/ {if (line) print line; line=$0; next} {gsub(/^ +/,""); line=line " | " $0} END {print line}' | cut -c1-220
