"""Copy the rocprofv3 summaries of one tools/profile_round.sh run from gpurun_out/ into profiles/
(tracked) and derive the per-kernel HBM traffic JSON that bench.py's ``roofline.traffic`` reads.
usage: python tools/summarize_profiles.py TAG ROUND   (e.g. r01f r01)"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
dst = os.path.join(ROOT, 'profiles')
for name in ('B65536', 'B1048576', 'c4', 'eval'):
    f = os.path.join(src, f'stats_{name}', 'bench_kernel_stats.csv')
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, f'{rnd}_bench_{name}_kernel_stats.csv'))


def short(name):
    name = re.sub(r'collie::\(anonymous namespace\)::', '', name).replace('void ', '')
    return name.split('(')[0]


def counters(folder):
    d = collections.defaultdict(list)
    path = os.path.join(src, folder, 'bench_counter_collection.csv')
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if 'collie' in r['Kernel_Name']:
                d[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in d.items()}


out = {
    '_command': 'rocprofv3 --pmc FETCH_SIZE (and, in a separate run, --pmc WRITE_SIZE) -- python3 '
                'bench.py --no-cpu-baseline --eval-users 0 ... (tools/profile_round.sh)',
    '_correction': 'counter values are KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests at '
                   '64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is',
}
for key, suffix in (('c2_B1048576', 'B1048576'), ('c4_B1048576', 'c4')):
    fetch, write = counters(f'pmc_fetch_{suffix}'), counters(f'pmc_write_{suffix}')
    per = {}
    for k in sorted(set(fetch) | set(write)):
        fb = int(fetch.get(k, (0, 0))[0] * 1024 * 2)
        wb = int(write.get(k, (0, 0))[0] * 1024)
        per[k] = {'launches': fetch.get(k, write.get(k))[1], 'fetch_bytes_corrected': fb,
                  'write_bytes': wb, 'hbm_bytes_per_launch': fb + wb}
    out[key] = per
json.dump(out, open(os.path.join(dst, f'{rnd}_pmc_traffic.json'), 'w'), indent=1)
for key in ('c2_B1048576', 'c4_B1048576'):
    print(key)
    for k, v in out[key].items():
        print(f"  {k:45s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")
