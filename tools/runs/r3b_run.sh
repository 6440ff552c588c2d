export TMPDIR=/tmp
FZP_PIPE_TIMING=1 SWEEP_TORCH_FIRST=1 timeout 900 python3 tools/e2e_sweep.py 2>&1 | grep "fzp_phase_contigs\|group_contigs\|fzp_job" | cut -c1-260
