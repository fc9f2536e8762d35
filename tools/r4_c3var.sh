#!/bin/bash
# the C3 headline on library variants, alternating with the default library, one session
out=gpurun_out/r4_c3var; mkdir -p $out
run() { # tag, lib suffix
  tag=$1; sfx=$2
  ( if [ -n "$sfx" ] && [ "$sfx" != base ]; then export NXS_GPU_LIB=$PWD/nxsearch_amd/csrc/libnxsearch_gpu_$sfx.so; fi
    python3 bench.py --cpu-seconds 0 --no-extras > $out/$tag.json 2>> $out/err.log
    echo "$tag $(python3 tools/show_bench.py $out/$tag.json 2>/dev/null | head -1) $(python3 -c "import json;d=json.loads(open('$out/$tag.json').read().strip().splitlines()[-1]);h=d['host_ms_per_step'];print('plan %.3f queue %.3f wait %.3f span %.3f' % (h['plan_ms'],h['queue_ms'],h['wait_ms'],d['roofline']['step']['span_ms']), ' '.join('%s:%.3f' % (k['kernel'][:22], k['ms']) for k in d['roofline']['per_kernel'][:2]))")" )
}
for rep in $(seq 1 ${REPS:-2}); do run base base; for v in "$@"; do run $v $v; done; done
