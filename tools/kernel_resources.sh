#!/bin/bash
# Compiles one .hip file for gfx950 and prints VGPR / SGPR / scratch / occupancy per kernel (dev helper).
#   tools/kernel_resources.sh deep-video-mvs_amd/csrc/sweep_tiled.hip [filter-regex]
src="$1"; filt="${2:-.}"
cd "$(dirname "$src")" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c "$(basename "$src")" -o /tmp/kr_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | awk '
  /Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
  /VGPRs:/ && !/AGPR/ {v=$NF; sub(/.*VGPRs: /,"",$0); v=$1}
  /TotalSGPRs:/ {s=$0; sub(/.*TotalSGPRs: /,"",s); sub(/ .*/,"",s)}
  /ScratchSize/ {sc=$0; sub(/.*: /,"",sc); sub(/ .*/,"",sc)}
  /Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ .*/,"",o)}
  /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); print name, "VGPR", v, "SGPR", s, "scratch", sc, "occ", o, "lds", l}' | grep -E "$filt"
rm -f /tmp/kr_$$.o
