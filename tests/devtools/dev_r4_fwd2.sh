cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_fwd2
timeout 400 python -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "forward_bit_exact or full_size_s1m or fuzz_bit_exact or lower_sh or precomputed_inputs or tight_tile" > gpurun_out/r4_fwd2/pytest.txt 2>&1; tail -3 gpurun_out/r4_fwd2/pytest.txt
bash tests/devtools/dev_r4_profiles.sh
