mkdir -p gpurun_out/r05f
./tools/micro/f2w > gpurun_out/r05f/fillers_two_waves.txt 2>&1
cat gpurun_out/r05f/fillers_two_waves.txt
