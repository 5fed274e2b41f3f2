#!/bin/bash
# Kernel-trace stats of bench.py legs on the GPU box: tools/profile_quick.sh TAG [legs...]
# legs: B1048576 B65536 c4 (default: all three) -> gpurun_out/prof_TAG/<leg>/bench_kernel_stats.csv
TAG=${1:-x}
shift
LEGS=${@:-B1048576 B65536 c4}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --eval-users 0 --hybrid-steps 0 --c4-steps 0 --small-batch-steps 0 --zipf-steps 0"
for leg in $LEGS; do
  case $leg in
    B1048576) ARGS="--batch 1048576 --steps 30 --warmup 5" ;;
    B65536) ARGS="--batch 65536 --steps 30 --warmup 5" ;;
    c4) ARGS="--shape c4 --steps 10 --warmup 3" ;;
    *) echo "unknown leg $leg"; exit 1 ;;
  esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$leg -o bench -- \
    python3 $R/bench.py $COMMON $ARGS > $OUT/$leg.log 2>&1 || { echo "$leg failed"; tail -5 $OUT/$leg.log; exit 1; }
  grep '"metric"' $OUT/$leg.log | cut -c1-220
done
