cd /root/repo
timeout 900 python -m pytest tests/test_zip_paths.py -x -q -m gpu 2>&1 | grep -v "^Extension" | tail -3
timeout 600 python tools/bench_zip.py --rays 65536 --train-only 2>&1 | tail -1
