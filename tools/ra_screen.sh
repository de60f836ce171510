#!/bin/bash
# Static screen of register-allocation outcomes: compile rm_kernels.hip of a source directory
# with extra flags, report scratch size and where the spills sit (tools/isa_spills.py).
#   tools/ra_screen.sh <srcdir> <name> [extra hipcc flags...]
SRC=$1; NAME=$2; shift 2
OUT=/tmp/vb/out_$NAME; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -std=c++17 -ffp-contract=off "$@" --cuda-device-only -S $SRC/rm_kernels.hip -o $OUT/k.s -Rpass-analysis=kernel-resource-usage 2> $OUT/res.txt
SCR=$(grep -A11 "render_frame_kernelILb1ELi7ELb0ELb0\|render_frame_kernelILb1ELi7ELb0EEE" $OUT/res.txt | grep "ScratchSize" | head -1 | grep -o "[0-9]*$")
python3 /root/repo/tools/isa_spills.py $OUT/k.s render_frame_kernelILb1ELi7ELb0 > $OUT/sp.txt 2>&1
DEEP=$(awk '/loop depth/{d=$3+0; gsub(":","",d); if (d>=4) {l+=$4; s+=$7}} END{print l"+"s}' $OUT/sp.txt)
TIGHT=$(awk '/instr,/{ if ($3+0 < 300) t+=$5 } END{print t+0}' $OUT/sp.txt)
echo "$NAME: scratch ${SCR}B, scratch ops at depth>=4: $DEEP (loads+stores), in loops <300 instr: $TIGHT"
