#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box: kernel-trace stats of bench.py (two batch
# sizes) and the HBM-traffic counters of the same command in separate --pmc passes.
# usage: tools/profile_round.sh TAG      -> gpurun_out/prof_TAG/...
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --eval-users 0 --steps 30 --warmup 5"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_B65536 -o bench -- python3 $REPO/bench.py $ARGS --batch 65536 > $OUT/stats_B65536.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_B1048576 -o bench -- python3 $REPO/bench.py $ARGS --batch 1048576 > $OUT/stats_B1048576.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_B1048576 -o bench -- python3 $REPO/bench.py $ARGS --batch 1048576 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_B1048576 -o bench -- python3 $REPO/bench.py $ARGS --batch 1048576 > $OUT/pmc_write.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_eval -o bench -- python3 $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 --eval-users 65536 > $OUT/stats_eval.log 2>&1
ls -R $OUT | head -60
