run() { # label, lib suffix, env...
  label=$1; libsfx=$2; shift 2
  ( for kv in "$@"; do export $kv; done
    if [ -n "$libsfx" ]; then export NXS_GPU_LIB=$PWD/nxsearch_amd/csrc/libnxsearch_gpu_$libsfx.so; fi
    python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['replay_ms'])" )
}
