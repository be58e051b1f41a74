L=gaussian-opacity-fields_amd/lib
for lib in ${LIBS:-"" static clock}; do
  if [ "$lib" = "product" ] || [ -z "$lib" ]; then unset GOF_HIP_LIB; else export GOF_HIP_LIB=$PWD/$L/libgof_hip_$lib.so; fi
  timeout 120 python tests/devtools/dev_tile_schedule.py $SCENES 2>&1 | grep "^{" >> gpurun_out/r3_sched.jsonl
done
