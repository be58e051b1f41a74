cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06h
timeout 600 python tests/golden/make_param_grad_errors.py gpurun_out/r06h/product_param_grad_errors.json 2>&1 | tail -12
cp gpurun_out/r06h/product_param_grad_errors.json tests/golden/product_param_grad_errors.json
timeout 900 python -m pytest tests/test_reference_gpu.py tests/test_parity_gpu.py -q -m gpu -x --tb=short -k "parameter_gradients or channel_slices or integrate_bit_exact or forward_bit_exact or backward_blend or per_call or two_streams" 2>&1 | tail -4
bash tests/devtools/dev_r6_final.sh 2>&1 | tail -45
