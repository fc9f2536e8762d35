#!/bin/bash
. tools/ab2.sh
for d in 16 32 64 256 16; do run gain$d - NXS_GPU_BM_GAIN=$d; done
