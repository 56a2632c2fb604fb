#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by scripts/collect_profiles.sh) into tracked files under profiles/."""
import csv, glob, json, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = f'gpurun_out/prof_{tag}'
os.makedirs('profiles', exist_ok=True)
bench = json.loads(open(f'{src}/bench.json').read().strip().splitlines()[-1])
json.dump(bench, open(f'profiles/{tag}_bench.json', 'w'), indent=1)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)      # gpurun merges runs: take the latest
stats = newest(f'{src}/trace/*/*_kernel_stats.csv')
rows = list(csv.DictReader(open(stats)))
with open(f'profiles/{tag}_bench_kernel_stats.csv', 'w') as f:
    w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage'])
    for r in rows:
        w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage']])
def pmc(sub, counter):
    f = newest(f'{src}/{sub}/*/*counter_collection.csv')
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            d[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in d.items()}
fetch, write = pmc('pmc_fetch', 'FETCH_SIZE'), pmc('pmc_write', 'WRITE_SIZE')
out = {}
for k in fetch:
    if 'qhea' in k:
        # MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of a
        # wide (16 B/lane) coalesced read stream -> doubled; WRITE_SIZE is exact for 16-B stores (ours are 8/16 B: uncalibrated)
        fb, wb = fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
        out[k.split('(')[0]] = {'FETCH_SIZE_KiB': fetch[k], 'WRITE_SIZE_KiB': write.get(k, 0.0),
                                'hbm_bytes_corrected': 2.0 * fb + wb, 'hbm_bytes_raw': fb + wb}
json.dump(out, open(f'profiles/{tag}_pmc_traffic.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
for r in rows[:8]:
    print(r['Name'][:80].ljust(80), r['Calls'], r['AverageNs'], r['Percentage'])
