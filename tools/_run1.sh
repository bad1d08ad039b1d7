cd /root/repo
timeout 900 python -m pytest tests/test_mlp.py tests/test_gpu_kernels.py tests/test_paths.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/gemm_reference_point.py 2>&1 | tail -1 | tee gpurun_out/r2_g_gemm_reference_point.json.log
timeout 400 python bench.py --no-cpu --no-eager --no-f32 2>&1 | tail -1 | cut -c1-1500
