#!/bin/bash
# one binary per ablation mask of tools/exp/u8_patch_anatomy.hip (built HERE, run on the GPU box)
cd $(dirname $0)
for m in 0 1 2 4 8 16 3 7 15 31; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DTAMD_U8P_ABLATE=$m -I../../tengine_amd/csrc -o u8_patch_anatomy_$m.bin u8_patch_anatomy.hip ../../tengine_amd/csrc/direct.cc -lhsa-runtime64 &
done
wait
ls -la u8_patch_anatomy_*.bin
