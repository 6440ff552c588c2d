export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_ovlp.py tests/test_gpu_track.py -x -q 2>&1 | tail -3
timeout 900 python3 tools/bench_ovlp.py 2>&1 | tail -2
timeout 900 python3 tools/bench_track.py 2>&1 | tail -2
