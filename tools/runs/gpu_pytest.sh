#!/bin/bash
# usage (through gpurun, from the repo root): bash tools/runs/gpu_pytest.sh <tag> <pytest arguments ...>
# runs the given GPU tests, log to gpurun_out/<tag>.log (the tail comes back on stdout)
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python3 -m pytest -m gpu -x -q "$@" > gpurun_out/$tag.log 2>&1
rc=$?
tail -25 gpurun_out/$tag.log
exit $rc
