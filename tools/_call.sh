mkdir -p gpurun_out/r4l
export RAYMARCH_SKIP_LINT=1
RAYMARCH_LIB=libraymarch_hip_ab_dbg1.so timeout 600 python tools/_dbg_l3.py > gpurun_out/r4l/dbg3.txt 2>&1
tail -30 gpurun_out/r4l/dbg3.txt
