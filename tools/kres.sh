#!/bin/bash
# usage: tools/kres.sh <file.hip> [grep pattern]  - one line per kernel: VGPRs, spills, occupancy (hipcc remarks, no GPU needed)
f=$1; pat=${2:-.}
cd "$(dirname "$0")/../deepsphere-weather_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${KRES_FLAGS} -c "$f" -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|Name:|VGPRs:|VGPRs Spill|Occupancy" \
 | sed -E 's/.*(Name|Function Name): ([^ ]*).*/\2/; s/.*VGPRs: ([0-9]+).*/v=\1/; s/.*VGPRs Spill: ([0-9]+).*/spill=\1/; s/.*Occupancy \[waves\/SIMD\]: ([0-9]+).*/occ=\1/' \
 | paste - - - - | while read n a b c; do echo "$(echo $n | c++filt | sed 's/(anonymous namespace):://g; s/(.*//') $a $b $c"; done | grep -E "$pat"
rm -f /tmp/kres_$$.o
