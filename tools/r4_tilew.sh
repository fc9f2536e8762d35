#!/bin/bash
# default limit (MODE_BIG tiles) on library variants (tile width, prefetch ring), one session; "base" = the default library
out=gpurun_out/r4_tilew; mkdir -p $out
run() { # tag, lib suffix
  tag=$1; sfx=$2
  ( if [ -n "$sfx" ] && [ "$sfx" != base ]; then export NXS_GPU_LIB=$PWD/nxsearch_amd/csrc/libnxsearch_gpu_$sfx.so; fi
    python3 bench.py --limit 1000 --steps 48 --warmup 4 --cpu-seconds 0 --no-extras > $out/$tag.json 2>> $out/err.log
    echo "$tag $(python3 tools/show_bench.py $out/$tag.json 2>/dev/null | head -1)" )
}
run base ""
for v in "$@"; do run $v $v; done
