#!/bin/bash
# A/B of env settings on the C3 bench: tools/r4_ab.sh "label ENV=.. ENV=.." "label2 ..." ...
. tools/ab2.sh
for spec in "$@"; do set -- $spec; label=$1; shift; run $label - "$@"; done
