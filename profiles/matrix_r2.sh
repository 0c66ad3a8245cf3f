# profiles/matrix_r2.sh -- on the GPU box: build variants / run-time switches of the general instance
# against each other on one device (devices differ by 10 % and more)
for lib in g4 default pbinl pbinl3 pbnoi3; do
  if [ $lib = default ]; then unset RNAMOTIF_AMD_LIB; else export RNAMOTIF_AMD_LIB=$PWD/rnamotif_amd/csrc/build_var/$lib/librnamotif_amd.so; fi
  echo "== lib=$lib"
  python profiles/ablate.py --reps 3 pk1.descr qu+tr.descr pk_j1+2.descr 2>gpurun_out/matrix_$lib.err | cut -c1-200
done
