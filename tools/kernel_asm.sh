#!/bin/bash
# usage: tools/kernel_asm.sh <mangled-name-substring> [out.s]
# Device assembly of rtgpu.hip (build/asm/rtgpu.s, rebuilt when stale), the text of one kernel cut out of it, and that
# kernel's register / scratch / LDS / occupancy summary.
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/asm
if [ ! -f $R/build/asm/rtgpu.s ] || [ -n "$(find $R/raytracer_amd/csrc -newer $R/build/asm/rtgpu.s -name '*.h*')" ]; then
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -o $R/build/asm/rtgpu.s $R/raytracer_amd/csrc/rtgpu.hip 2>/dev/null || exit 1
fi
OUT=${2:-$R/build/asm/kernel.s}
awk -v pat="$1" '
    /^_Z[A-Za-z0-9_]*:/ { on = (index($1, pat) > 0) }
    on { print }
    on && /^; COMPUTE_PGM_RSRC2:TRAP_HANDLER/ { on = 0 }' $R/build/asm/rtgpu.s > $OUT
grep -E "^_Z.*:|codeLenInByte|TotalNumSgprs|NumVgprs:|ScratchSize|LDSByteSize|Occupancy" $OUT
