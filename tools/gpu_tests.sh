#!/bin/bash
# Runs on the GPU box (through gpurun): the whole GPU suite into gpurun_out/<tag>/pytest.log.
# Usage: tools/gpu_tests.sh <tag> [pytest args]
TAG=${1:-tests}; shift
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q --timeout 900 --durations=8 "$@" 2>&1 | tail -40 ) > $O/pytest.log 2>&1
cat $O/pytest.log
