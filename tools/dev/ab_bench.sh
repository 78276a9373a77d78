#!/bin/bash
# A/B of two builds of the library on ONE box: tools/ab_bench.sh <libA or -> <libB> [bench args]; "-" = the in-tree build.
# Alternates the two (A B A B) so that box-to-box and warm-up differences cancel.
A=$1; B=$2; shift 2
ARGS="--steps 10 --warmup 3 --cpu-seconds 0 --side 0 $*"
one() { if [ "$1" = "-" ]; then python bench.py $ARGS; else REEVR_AMD_LIB=$1 python bench.py $ARGS; fi 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', d['value'], d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['kernels_ms'].items()})"; }
for i in 1 2; do one $A; one $B; done
