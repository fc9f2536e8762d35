#!/bin/bash
# Runs on the GPU box (via gpurun): the default bench line, a rocprofv3 kernel-trace/stats run of the
# same command, and separate --pmc passes (never combined with other trace domains); then the other
# workloads (C2, C4, C5, the default limit, sparse doc ids, doc shards) with their own kernel stats and
# -- C2, C5 -- their own HBM traffic.   Usage: tools/profile_round.sh <tag>   outputs: gpurun_out/<tag>_summary/
set -u
tag=${1:-r4}
out=$PWD/gpurun_out
sum=$out/${tag}_summary
mkdir -p "$sum"
export TMPDIR=/tmp
note() { echo "[profile_round $(date +%T)] $*"; }

# (2) C3: kernel stats + counters.  8 steps = every one of the 4 rotated batches twice
B="python3 bench.py --steps 7 --warmup 1 --cpu-seconds 0 --no-extras --min-seconds 0"
prof() { # prefix, workload args..., then the counters come from the caller
  name=$1; shift
  note "$name: kernel trace"
  rocprofv3 --kernel-trace --stats -d "$out/${tag}_${name}_stats" -o run -- python3 bench.py --steps 7 --warmup 1 --cpu-seconds 0 --no-extras --min-seconds 0 "$@" \
      > "$sum/${tag}_${name}_bench_under_rocprof.json" 2> "$out/${tag}_${name}_stats.log"
}
pmc() { # name, dir suffix, counters..., -- workload args
  name=$1; sfx=$2; shift 2
  ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  note "$name: pmc ${ctrs[*]}"
  rocprofv3 --pmc "${ctrs[@]}" -d "$out/${tag}_${name}_pmc_${sfx}" -o run -- python3 bench.py --steps 7 --warmup 1 --cpu-seconds 0 --no-extras --min-seconds 0 "$@" \
      > /dev/null 2> "$out/${tag}_${name}_pmc_${sfx}.log"
}
# PARTS=a: C3 + C2 + C4 + default limit + sparse ids + doc shards; PARTS=b: C5 (two gpurun calls of <= 20 min each)
PARTS=${PARTS:-ab}
if [ "${PARTS#*a}" != "$PARTS" ]; then
prof c3
pmc c3 fetch FETCH_SIZE --
pmc c3 write WRITE_SIZE --
pmc c3 sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --
pmc c3 sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --
# FETCH_SIZE against known byte counts, 16 B/lane and 8 B/lane (the scan kernels' width)
note "FETCH_SIZE calibration"
rocprofv3 --pmc FETCH_SIZE -d "$out/${tag}_pmc_calib" -o run -- python3 tools/pmc_calib.py > "$sum/${tag}_pmc_calib.json" 2> "$out/${tag}_pmc_calib.log"
python3 tools/pmc_summary.py --stats "$out/${tag}_c3_stats" --pmc "$out/${tag}_c3_pmc_fetch" "$out/${tag}_c3_pmc_write" "$out/${tag}_c3_pmc_sq1" "$out/${tag}_c3_pmc_sq2" \
    --steps 8 --out-prefix "$sum/${tag}" --command "$B" --calib "$out/${tag}_pmc_calib" --calib-json "$sum/${tag}_pmc_calib.json"
# (1) the driver's command, as the driver runs it -- AFTER the counters: its roofline.traffic comes from the summary just
# made (stamped with the hash of the kernel sources: bench.py refuses a summary of other kernels)
cp "$sum/${tag}_pmc_summary.json" profiles/ 2>/dev/null
note "default bench"
python3 bench.py > "$sum/${tag}_bench.json" 2> "$out/${tag}_bench.err"

# (3) the other configurations
for w in C2 C4; do
  lw=$(echo $w | tr A-Z a-z)
  prof $lw --workload $w
  pmc $lw fetch FETCH_SIZE -- --workload $w
  pmc $lw write WRITE_SIZE -- --workload $w
  python3 tools/pmc_summary.py --stats "$out/${tag}_${lw}_stats" --pmc "$out/${tag}_${lw}_pmc_fetch" "$out/${tag}_${lw}_pmc_write" \
      --steps 8 --out-prefix "$sum/${tag}_${lw}" --command "$B --workload $w" --name $w \
      --docs $([ $w = C2 ] && echo 1000000 || echo 10000000) --terms $([ $w = C2 ] && echo 100000 || echo 1000000) \
      --calib "$out/${tag}_pmc_calib" --calib-json "$sum/${tag}_pmc_calib.json"
  cp "$sum/${tag}_${lw}_pmc_summary.json" profiles/ 2>/dev/null
  python3 bench.py --workload $w --cpu-seconds 10 > "$sum/${tag}_${lw}_bench.json" 2>> "$out/${tag}_bench.err"
done
# C2 with the plan cache on (the loop's query strings repeat: planning becomes a hash lookup)
python3 bench.py --workload C2 --plan-cache --cpu-seconds 0 --no-extras > "$sum/${tag}_c2_plancache_bench.json" 2>> "$out/${tag}_bench.err"
# the default limit (params == NULL => 1000): MODE_BIG kernels + k_replay_coop
prof l1000 --limit 1000
python3 tools/pmc_summary.py --stats "$out/${tag}_l1000_stats" --steps 8 --out-prefix "$sum/${tag}_l1000" --command "$B --limit 1000" --limit 1000
# sparse random u64 doc ids
note "sparse ids"
python3 bench.py --sparse-ids --cpu-seconds 10 --no-extras > "$sum/${tag}_sparse_ids_bench.json" 2>> "$out/${tag}_bench.err"
# doc shards (N4) on one GPU
note "doc shards"
python3 bench.py --docshard 4 --steps 10 --warmup 2 --cpu-seconds 10 > "$sum/${tag}_docshard4_c3_bench.json" 2>> "$out/${tag}_bench.err"
fi
if [ "${PARTS#*b}" != "$PARTS" ]; then
# C5 on one GPU: bench line, kernel stats, traffic (the calibration of part a is reused if its summary is there)
note "C5"
if [ ! -d "$out/${tag}_pmc_calib" ]; then
  rocprofv3 --pmc FETCH_SIZE -d "$out/${tag}_pmc_calib" -o run -- python3 tools/pmc_calib.py > "$sum/${tag}_pmc_calib.json" 2> "$out/${tag}_pmc_calib.log"
fi
prof c5 --workload C5
pmc c5 fetch FETCH_SIZE -- --workload C5
pmc c5 write WRITE_SIZE -- --workload C5
python3 tools/pmc_summary.py --stats "$out/${tag}_c5_stats" --pmc "$out/${tag}_c5_pmc_fetch" "$out/${tag}_c5_pmc_write" \
    --steps 8 --out-prefix "$sum/${tag}_c5" --command "$B --workload C5" --name C5 --docs 50000000 --terms 2000000 --batch 8192 \
    --calib "$out/${tag}_pmc_calib" --calib-json "$sum/${tag}_pmc_calib.json"
cp "$sum/${tag}_c5_pmc_summary.json" profiles/ 2>/dev/null
python3 bench.py --workload C5 --steps 8 --warmup 2 --cpu-seconds 10 --no-extras > "$sum/${tag}_c5_1gpu_bench.json" 2>> "$out/${tag}_bench.err"
python3 bench.py --workload C5 --docshard 4 --steps 5 --warmup 1 --cpu-seconds 0 > "$sum/${tag}_docshard4_c5_bench.json" 2>> "$out/${tag}_bench.err"
fi
# the raw rocpd databases are ~10 MB each: gpurun merges at most 64 MiB back
rm -rf "$out/${tag}"_*_stats "$out/${tag}"_*_pmc_* "$out/${tag}_pmc_calib"
note "done"
ls -la "$sum"
tail -c 1200 "$sum/${tag}_bench.json"
