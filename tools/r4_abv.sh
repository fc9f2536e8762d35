#!/bin/bash
# A/B of library variants on the C3 bench: tools/r4_abv.sh v1 v2 ... ("-" = default build)
. tools/ab2.sh
for v in "$@"; do run $v $v; done
