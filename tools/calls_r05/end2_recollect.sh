#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05end; mkdir -p $o
PMC_MIN=1 timeout 1000 bash tools/collect_profiles.sh r05z > $o/collect.log 2>&1; tail -3 $o/collect.log
ls gpurun_out/r05z
