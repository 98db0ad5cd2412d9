#!/bin/bash
# per-kernel VGPR / spill / LDS / occupancy table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage)
# usage: tools/kernel_resources.sh ssd_tensorflow_amd/csrc/ops.hip [name filter regex]
f=$1; pat=${2:-.}
extra=""; case "$f" in *boxes.hip|*metrics.hip|*augment.hip) extra=-ffp-contract=off;; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 |
  sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /AGPRs:/ {a=$NF} /SGPRs Spill:/ {ss=$NF} /VGPRs Spill:/ {vs=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {o=$NF} /LDS Size/ {print name, "vgpr="v, "agpr="a, "sgpr_spill="ss, "vgpr_spill="vs, "scratch="sc, "occ="o, "lds="$NF}' |
  c++filt | grep -E "$pat"
exit 0
