#!/bin/bash
# builds (here, no GPU needed) or runs (on the GPU box) the timing-only ablation binaries of k_rotate_pblock8
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for a in 0 1 2 3 4; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude -DPQHIP_TIMING_ONLY_BUILD -DROT8_ABLATE=$a tools/rot8_ablate.hip -o tools/rot8_ablate_$a 2>/dev/null &
  done
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude -DPQHIP_TIMING_ONLY_BUILD -DROT8_ABLATE=0 -DROT8_STORE=1 tools/rot8_ablate.hip -o tools/rot8_ablate_s 2>/dev/null &
  wait; ls tools/rot8_ablate_?
else
  for r in 1 2; do for a in 0 1 2 3 4; do timeout -k 5 60 tools/rot8_ablate_$a ${1:-2.0} 0; done; done
  echo "plain stores:"; timeout -k 5 60 tools/rot8_ablate_s ${1:-2.0} 0; timeout -k 5 60 tools/rot8_ablate_0 ${1:-2.0} 0; timeout -k 5 60 tools/rot8_ablate_s ${1:-2.0} 0
fi
