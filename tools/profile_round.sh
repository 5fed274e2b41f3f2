#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box: kernel-trace stats of bench.py (C2 at two
# batch sizes, the C4 tables on one GPU, the eval leg) and the HBM-traffic counters of the same
# commands in separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/profile_round.sh TAG      -> gpurun_out/prof_TAG/...
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --eval-users 0 --hybrid-steps 0 --c4-steps 0 --small-batch-steps 0 --zipf-steps 0 --steps 30 --warmup 5"
C4="--no-cpu-baseline --eval-users 0 --hybrid-steps 0 --c4-steps 0 --small-batch-steps 0 --zipf-steps 0 --steps 10 --warmup 3 --shape c4"
stats() { timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -o bench -- python3 $REPO/bench.py ${@:2} > $OUT/$1.log 2>&1; }
pmc() { timeout -s KILL 300 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o bench -- python3 $REPO/bench.py ${@:3} > $OUT/$1.log 2>&1; }
# second argument "eval-hybrid": only the eval leg and the hybrid head (what changed late in round 2)
ONLY=${2:-all}
if [ "$ONLY" = all ]; then
stats stats_B65536 $ARGS --batch 65536
stats stats_B1048576 $ARGS --batch 1048576
stats stats_c4 $C4
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_hybrid -o bench -- python3 $REPO/tools/hybrid_bench.py --steps 10 --warmup 3 > $OUT/stats_hybrid.log 2>&1
stats stats_eval --no-cpu-baseline --hybrid-steps 0 --c4-steps 0 --small-batch-steps 0 --zipf-steps 0 --steps 2 --warmup 1 --eval-users 65536
EVAL="--no-cpu-baseline --hybrid-steps 0 --c4-steps 0 --small-batch-steps 0 --zipf-steps 0 --steps 1 --warmup 1 --eval-users 65536"
pmc pmc_sq_eval "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS" $EVAL
if [ "$ONLY" != all ]; then ls $OUT; exit 0; fi
pmc pmc_fetch_B1048576 FETCH_SIZE $ARGS --batch 1048576
pmc pmc_write_B1048576 WRITE_SIZE $ARGS --batch 1048576
pmc pmc_fetch_c4 FETCH_SIZE $C4
pmc pmc_write_c4 WRITE_SIZE $C4
ls $OUT
