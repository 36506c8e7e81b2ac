#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3t
export TMPDIR=/tmp
timeout 1200 python -m pytest "$@" -x -q > gpurun_out/r3t/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3t/pytest.log
