#!/bin/bash
# PMC comparison of two builds of the library on the bench workload (GPU box, via gpurun).
# Usage: tools/pmc_ab.sh <libsuffix-or-"default"> ...   -> gpurun_out/pmcab_<name>_{1,2}
export TMPDIR=/tmp
B="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras"
for name in "$@"; do
  if [ "$name" = default ]; then unset NXS_GPU_LIB; else export NXS_GPU_LIB=$PWD/nxsearch_amd/csrc/libnxsearch_gpu_$name.so; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d gpurun_out/pmcab_${name}_1 -o run -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d gpurun_out/pmcab_${name}_2 -o run -- $B > /dev/null 2>&1
done
