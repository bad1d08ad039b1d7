cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^Extension" | tail -4
timeout 400 python bench.py --no-cpu --no-eager --no-f32 2>&1 | tail -1 | cut -c1-1500
timeout 300 python bench.py --no-cpu --no-eager --no-f32 --no-frame --rays 512 2>&1 | tail -1 | cut -c1-300
