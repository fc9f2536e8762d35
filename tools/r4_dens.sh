#!/bin/bash
. tools/ab2.sh
for d in 0 0.01 0.02 0.03 0.05 0; do run dens$d - NXS_GPU_SCANB_DENS=$d; done
