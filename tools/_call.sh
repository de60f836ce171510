mkdir -p gpurun_out/r4x
(for s in 111 112; do timeout 1200 python tools/fuzz_parity.py --contract gfx950 --cases 6000 --seed $s --seconds 300 --passes 1,2,3,4,8,16,20,25,32 2>&1 | tail -1; done
 for s in 121 122; do timeout 1200 python tools/fuzz_parity.py --contract cpu --cases 6000 --seed $s --seconds 300 --passes 1,2,3,4,8,16,20,25,32 2>&1 | tail -1; done
) > gpurun_out/r4x/fuzz.txt 2>&1
cat gpurun_out/r4x/fuzz.txt
