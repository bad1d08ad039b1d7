cd /root/repo
timeout 600 python -m pytest tests/test_mlp.py -x -q -m gpu 2>&1 | grep -v "^Extension" | tail -3
timeout 200 python tools/fmlp_single.py 2>&1 | tail -4
timeout 300 python tools/bench_classic.py --steps 5 2>&1 | tail -1 | tee gpurun_out/r2_f_pathB_train_fused.json.log
