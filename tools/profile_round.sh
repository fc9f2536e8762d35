#!/bin/bash
# Runs on the GPU box (via gpurun): the default bench line, a rocprofv3 kernel-trace/stats run of the
# same command, and separate --pmc passes (never combined with other trace domains).
# Usage: tools/profile_round.sh <tag>      outputs under gpurun_out/<tag>_*
set -u
tag=${1:-r1}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
B="python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extras"
rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats" -o run -- $B > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_stats.log"
rocprofv3 --pmc FETCH_SIZE -d "$out/${tag}_pmc_fetch" -o run -- $B > /dev/null 2> "$out/${tag}_pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE -d "$out/${tag}_pmc_write" -o run -- $B > /dev/null 2> "$out/${tag}_pmc_write.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d "$out/${tag}_pmc_sq1" -o run -- $B > /dev/null 2> "$out/${tag}_pmc_sq1.log"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$out/${tag}_pmc_sq2" -o run -- $B > /dev/null 2> "$out/${tag}_pmc_sq2.log"
# FETCH_SIZE against known byte counts, 16 B/lane and 8 B/lane (the scan kernels' width)
rocprofv3 --pmc FETCH_SIZE -d "$out/${tag}_pmc_calib" -o run -- python3 tools/pmc_calib.py > "$out/${tag}_pmc_calib.json" 2> "$out/${tag}_pmc_calib.log"
# the fuzzy path (C4) on its own: kernel stats of k_bk_level
rocprofv3 --kernel-trace --stats -d "$out/${tag}_fuzzy_stats" -o run -- python3 bench.py --workload C4 --steps 5 --warmup 1 --cpu-seconds 0 --no-extras > "$out/${tag}_fuzzy_bench.json" 2> "$out/${tag}_fuzzy_stats.log"
# condense on the box (the raw rocpd databases are ~10 MB each: gpurun merges at most 64 MiB back)
mkdir -p "$out/${tag}_summary"
python3 tools/pmc_summary.py --stats "$out/${tag}_stats" --pmc "$out/${tag}_pmc_fetch" "$out/${tag}_pmc_write" "$out/${tag}_pmc_sq1" "$out/${tag}_pmc_sq2" \
    --steps 6 --out-prefix "$out/${tag}_summary/${tag}" --command "$B" --calib "$out/${tag}_pmc_calib" --calib-json "$out/${tag}_pmc_calib.json"
python3 tools/pmc_summary.py --stats "$out/${tag}_fuzzy_stats" --steps 6 --out-prefix "$out/${tag}_summary/${tag}_fuzzy" \
    --command "python3 bench.py --workload C4 --steps 5 --warmup 1 --cpu-seconds 0 --no-extras"
rm -rf "$out/${tag}_stats" "$out/${tag}_pmc_fetch" "$out/${tag}_pmc_write" "$out/${tag}_pmc_sq1" "$out/${tag}_pmc_sq2" "$out/${tag}_pmc_calib" "$out/${tag}_fuzzy_stats"
tail -c 1500 "$out/${tag}_bench.json"
