export TMPDIR=/tmp
timeout 300 python3 tools/runs/tb_window_stats.py 2>&1 | tail -12
