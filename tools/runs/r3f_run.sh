export TMPDIR=/tmp
FZP_PIPE_TIMING=1 python3 bench.py --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2 2>&1 >/dev/null | grep -v amdgpu.ids | tail -8
