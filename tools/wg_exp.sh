set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in $EXPS; do
  export COLLIE_WG_EXP=$e
  timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/wg_exp$e -o s -- python3 $R/tools/hybrid_bench.py --steps 8 --warmup 3 > $R/gpurun_out/wg_exp$e.log 2>&1
  f=$(find $R/gpurun_out/wg_exp$e -name '*kernel_trace.csv' | head -1)
  echo "exp=$e $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/wg_exp$e.log)"
  python3 - "$f" <<'PY'
import csv, sys, collections, re
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'dense_' in n or 'narrow_' in n:
        short = re.search(r'(dense|narrow)_\w+(<[^>]*>)?', n).group(0)
        key = (short, r['Grid_Size_X'], r['Grid_Size_Y'])
        d[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(d.items()):
    v = v[len(v) // 3:]
    print('  ', k, len(v), round(sum(v) / len(v) / 1e3, 1), 'us')
PY
  rm -rf $R/gpurun_out/wg_exp$e
done
