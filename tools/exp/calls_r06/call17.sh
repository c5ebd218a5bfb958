#!/bin/bash
# round 6, call 17: dwpw -- which waves share a SIMD?  the opposite-order experiment with bit 0 / 1 / 2 of the wave index as the flip bit
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call17
mkdir -p $O
cd $R
for b in 1 2 4; do echo "== DWPW_FLIPBIT=$b"; timeout 200 tools/exp/dwpw_anatomy_flipbit$b.bin | head -8; done 2>&1 | tee $O/dwpw_anatomy_flipbits.txt
echo "== no flip"; timeout 200 tools/exp/dwpw_anatomy.bin | head -8 | tee -a $O/dwpw_anatomy_flipbits.txt
