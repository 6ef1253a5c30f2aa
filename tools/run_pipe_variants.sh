#!/bin/bash
# On the GPU box: the program-managed ring window (cbca_prog_gen.py --pipe) against the plain kernel, programs from the
# plain-Python builder (tools/dev_prog_check.py), cfg2 unless CFG is set.   bash tools/run_pipe_variants.sh > out.txt
CFG=${CFG:-cfg2}
run() { echo "== $*"; timeout 600 python tools/dev_prog_check.py --config $CFG --iters 20 --skip-small "$@" 2>&1 | grep -v "^shape" ; }
run --k 4 --w 20
run --k 4 --w 42 --pipe 21
run --k 4 --w 42 --pipe 16
run --k 4 --w 42 --pipe 31
run --k 4 --w 42 --pipe 42
run --k 4 --w 20 --pipe 10
run --k 4 --w 20 --pipe 20
run --k 4 --w 20
