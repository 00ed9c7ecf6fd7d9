mkdir -p gpurun_out/r4c
./tools/mfmaprec/mfmaprec > gpurun_out/r4c/mfmaprec.txt 2>&1
for a in wide64B wide64x3 wide4x128x3 wide4x64; do
  pts=1000000; [ $a = wide4x128x3 ] && pts=100000
  for lib in build_ab/libpinn_hip_prev.so pinn_for_quantum_wavefunction_surfaces_amd/libpinn_hip.so; do
    echo "== $a $lib" >> gpurun_out/r4c/ab.txt
    PINN_HIP_LIB=$lib python tools/kbench.py --arch $a --variants 0 --rounds 4 --steps 20 --points $pts 2>&1 | grep arch >> gpurun_out/r4c/ab.txt
  done
done
cat gpurun_out/r4c/mfmaprec.txt; cat gpurun_out/r4c/ab.txt
