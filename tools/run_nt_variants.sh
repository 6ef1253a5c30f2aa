#!/bin/bash
# On the GPU box: non-temporal loads for region rows made of unit-region pixels (cbca_prog_gen.py --ntload) against the
# plain kernel, programs from the plain-Python builder.   bash tools/run_nt_variants.sh > out.txt
CFG=${CFG:-cfg2}
run() { echo "== $*"; timeout 600 python tools/dev_prog_check.py --config $CFG --iters 20 --skip-small "$@" 2>&1 | grep -v "^shape" ; }
run --k 4 --w 20
run --k 4 --w 20 --ntload
run --k 4 --w 20
run --k 4 --w 20 --ntload
