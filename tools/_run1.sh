cd /root/repo
timeout 900 python -m pytest tests/test_mlp.py tests/test_gpu_kernels.py tests/test_paths.py -x -q -m gpu 2>&1 | grep -v "^Extension" | tail -2
timeout 400 python bench.py --no-cpu --no-eager --no-f32 --no-frame 2>&1 | tail -1 | cut -c1-300
timeout 300 python bench.py --no-cpu --no-eager --no-f32 --no-frame --rays 512 2>&1 | tail -1 | cut -c1-300
