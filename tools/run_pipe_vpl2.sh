#!/bin/bash
# On the GPU box: 256 disparities as two 128-disparity chunks (two per lane) with and without the program-managed ring.
CFG=${CFG:-cfg2}
run() { echo "== $*"; timeout 600 python tools/dev_prog_check.py --config $CFG --iters 20 --skip-small "$@" 2>&1 | grep -v "^shape\|amdgpu.ids" ; }
run --k 4 --w 20
run --k 4 --w 20 --vpl 2
run --k 4 --w 42 --pipe 21 --vpl 2
run --k 4 --w 42 --pipe 31 --vpl 2
run --k 4 --w 42 --pipe 21 --vpl 2 --minvgpr 168
run --k 4 --w 60 --pipe 31 --vpl 2
run --k 4 --w 30 --pipe 15 --vpl 2
