#!/bin/bash
# MODE_BIG knobs at four batches in flight (default limit), one session
out=gpurun_out/r4_bigknobs; mkdir -p $out
run() { # tag, env...
  tag=$1; shift
  env "$@" python3 bench.py --limit 1000 --steps 48 --warmup 4 --cpu-seconds 0 --no-extras > $out/$tag.json 2>> $out/err.log
  echo "$tag $(python3 tools/show_bench.py $out/$tag.json 2>/dev/null | head -1)"
}
run base X=1
run mp8 NXS_GPU_BIG_MINPOST=8
run mp32 NXS_GPU_BIG_MINPOST=32
run mp64 NXS_GPU_BIG_MINPOST=64
run mp128 NXS_GPU_BIG_MINPOST=128
run noearly NXS_GPU_AND_NOEARLY=1
