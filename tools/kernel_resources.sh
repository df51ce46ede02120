#!/bin/bash
# Print registers / scratch / LDS / occupancy of every kernel in the given .hip files (default: all conv instances),
# from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
cd "$(dirname "$0")/../cvvae_amd/csrc" || exit 1
files=${@:-conv_inst_*.hip}
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c $f -o /dev/null \
     -Rpass-analysis=kernel-resource-usage 2>&1 | awk -v f=$f '
    /Function Name:/ {name=$NF}
    /VGPRs:/ && !/Spill/ && !/Occupancy/ {v=$NF} /AGPRs:/ {a=$NF} /ScratchSize/ {s=$NF} /VGPR Spill/ {sp=$NF}
    /Occupancy/ {occ=$NF} /LDS Size/ {printf "%-16s vgpr=%-4s agpr=%-4s scratch=%-5s spill=%-4s occ=%s lds=%-7s %s\n", f, v, a, s, sp, occ, $NF, name}'
done
