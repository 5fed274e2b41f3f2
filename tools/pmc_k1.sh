#!/bin/bash
# kernel stats + SQ counters of bench.py at B=1M. usage: tools/pmc_k1.sh TAG
TAG=${1:-x}
OUT=$PWD/gpurun_out/k1_$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --eval-users 0 --steps 20 --warmup 5 --batch 1048576"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python3 $REPO/bench.py $ARGS > $OUT/stats.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -o b -- python3 $REPO/bench.py $ARGS > $OUT/pmc1.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc2 -o b -- python3 $REPO/bench.py $ARGS > $OUT/pmc2.log 2>&1 || exit 1
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('$OUT/stats/b_kernel_stats.csv')))
for r in rows[:8]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
for t in ('pmc1','pmc2'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f'$OUT/{t}/b_counter_collection.csv')):
        if 'mf_train_fwd' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, sum(v)/len(v))
PY
