L=gaussian-opacity-fields_amd/lib
if [ -z "$NOPYTEST" ]; then timeout 600 python -m pytest tests/test_parity_gpu.py -q --no-header -p no:cacheprovider -x -k "forward_bit_exact or backward_blend or fused_forward or bit_reproducible or integrate_bit_exact or clustered or integrate_view_cache" 2>&1 | tail -5 > gpurun_out/r3_sched_pytest.txt; fi
for lib in ${LIBS:-product static clock}; do
  if [ "$lib" = "product" ]; then unset GOF_HIP_LIB; else export GOF_HIP_LIB=$PWD/$L/libgof_hip_$lib.so; fi
  timeout 120 python tests/devtools/dev_tile_schedule.py $SCENES 2>&1 | grep "^{" >> gpurun_out/r3_sched.jsonl
done
tail -3 gpurun_out/r3_sched_pytest.txt
