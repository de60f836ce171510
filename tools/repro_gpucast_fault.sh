#!/bin/bash
# Reproducer of the allocation-dependent fault of the accelerated frame kernel.
#
# Same sources, one switch: -DRM_F2U_GPU_ASM=1 spells the GPU seed cast as the instruction itself
# (inline asm v_cvt_u32_f32) instead of the C expression the product uses.  Both forms compute the
# same value for every input.  With the asm form the GPU-cast instantiation of render_frame_kernel
# renders ~80 % of the pixels of a 64x48 frame wrong at 6 and 7 waves/SIMD (72 / 80 VGPRs) and
# every pixel right at 5 and 8; the x86-cast instantiation of the SAME library is right throughout,
# and so are the single-pass kernel and the plain (table-free) frame kernel in GPU-cast mode.
# tools/diag_cast.py then switches phases off through the option record: the wrong pixels need a
# light that passes the attenuation test (numLights > 0, minLightAtt small) and nothing else.
#
# (Works on the sources of commit 11be60c -- `git worktree add /tmp/w 11be60c`, built without the two -mllvm switches;
#  today's sources no longer carry the -DRM_F2U_GPU_ASM knob.  tools/isa_exec_lint.py on the assembly of that build finds
#  the allocator copy in an exec = 0 block that explains it; the live reproducer, with the faulting instruction
#  identified under rocgdb, is tools/repro_reload_exec0.sh.)
#
#   build container:  tools/repro_gpucast_fault.sh build
#   GPU box:          tools/repro_gpucast_fault.sh run     (gpurun -- tools/repro_gpucast_fault.sh run)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
case "$1" in
  build)
    python tools/ab_build.py asm7="-DRM_F2U_GPU_ASM=1" asm6="-DRM_F2U_GPU_ASM=1 -DRM_GPUCAST_MINW=6" \
                             asm8="-DRM_F2U_GPU_ASM=1 -DRM_GPUCAST_MINW=8" asm5="-DRM_F2U_GPU_ASM=1 -DRM_GPUCAST_MINW=5" ;;
  run)
    for v in asm7 asm6 asm8 asm5; do
      echo "== $v"; RAYMARCH_LIB=libraymarch_hip_ab_$v.so python tools/diag_cast.py 2>/dev/null | head -3
    done ;;
  *) echo "usage: $0 build|run"; exit 2 ;;
esac
