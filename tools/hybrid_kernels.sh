#!/bin/bash
# Per-launch durations of the hybrid head's kernels (rocprofv3 kernel trace of tools/hybrid_bench.py,
# grouped by kernel instance): what DESIGN.md's K10 per-layer numbers are read from.
# usage (on the GPU box): bash tools/hybrid_kernels.sh
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/hybrid_kernels
rm -rf $O
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $O -o s -- python3 $R/tools/hybrid_bench.py --steps 8 --warmup 3 > $O.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $O.log
python3 - "$(find $O -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections, re
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'dense_' in n or 'narrow_' in n:
        short = re.search(r'(dense|narrow)_\w+(<[^>]*>)?', n).group(0)
        d[short].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(d.items()):
    v = v[len(v) // 3:]
    print('  %-46s %3d launches %8.1f us' % (k, len(v), sum(v) / len(v) / 1e3))
PY
rm -rf $O
