cd /root/repo
timeout 900 python -m pytest tests/test_gpu_grid.py -x -q -m gpu -k double 2>&1 | grep -E "^E|^tests.*Error|assert_allclose|passed|failed" | head -30
