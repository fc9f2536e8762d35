# A/B inside one gpurun call (boxes differ by ~10 %): run <label> <lib suffix or -> [ENV=val ...]
run() { label=$1; sfx=$2; shift 2; ( for kv in "$@"; do export $kv; done
  if [ "$sfx" != "-" ]; then export NXS_GPU_LIB=$PWD/nxsearch_amd/csrc/libnxsearch_gpu_$sfx.so; fi
  NXS_BENCH_REPEATS=2 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extras --keep | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['repeat_ms_per_step'], d['roofline']['step']['span_ms'], d['roofline']['step']['of_which_after_last_scan_ms'], [(k['kernel'], k['ms']) for k in d['roofline']['per_kernel']], d['host_ms_per_step'])" ) }
