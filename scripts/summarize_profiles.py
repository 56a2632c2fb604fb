#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by scripts/collect_profiles.sh) into tracked files under profiles/:
<tag>_bench.json, <tag>_bench_k20.json, <tag>_bench_kernel_stats.csv, <tag>_pmc.json (per kernel, per launch)."""
import csv, glob, json, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
src = f'gpurun_out/prof_{tag}'
os.makedirs('profiles', exist_ok=True)
for nm in ('bench', 'bench_k20'):
    if os.path.exists(f'{src}/{nm}.json'):
        line = [l for l in open(f'{src}/{nm}.json').read().strip().splitlines() if l.startswith('{')][-1]
        json.dump(json.loads(line), open(f'profiles/{tag}_{nm}.json', 'w'), indent=1)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)      # gpurun merges runs: take the latest
rows = []
if glob.glob(f'{src}/trace/*/*_kernel_stats.csv'):                  # (a counters-only directory has no trace)
    stats = newest(f'{src}/trace/*/*_kernel_stats.csv')
    rows = list(csv.DictReader(open(stats)))
    with open(f'profiles/{tag}_bench_kernel_stats.csv', 'w') as f:
        w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage'])
        for r in rows:
            w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage']])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
pmc_files = []
for d in glob.glob(f'{src}/pmc_*'):                                 # gpurun merges runs into the same directory: newest pass only
    if os.path.isdir(d) and glob.glob(f'{d}/*/*counter_collection.csv'):
        pmc_files.append(newest(f'{d}/*/*counter_collection.csv'))
for f in pmc_files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'qhea' not in k:
            continue
        # keep template arguments, drop the parameter list: "void qhea::bwd_tri_kernel<5>(...)" -> "qhea::bwd_tri_kernel<5>"
        k = k.split('(')[0].replace('void ', '')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in acc.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e['launches_sampled'] = {c: len(v) for c, v in d.items()}
    if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
        # MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide
        # (16 B/lane) coalesced read stream -> doubled for the circuit kernels, whose reads are 16 B/lane streams (LDS-DMA
        # record fetches, gate / (cos,sin) tables).  "Other access widths are uncalibrated: calibrate on a known byte
        # count": the reduce kernels read 8 B/lane, and their known byte count -- the partial-sum matrix, rows x blk x
        # pad(3n) x 8 B, plus grad_x -- equals FETCH_SIZE as reported (cfg 2: 7.86 + 2.46 MB known, 7.6 MB + L2 hits
        # reported), so their factor is 1; the prep kernels read a few KB of parameters (factor immaterial).
        fb, wb = e['FETCH_SIZE'] * 1024.0, e['WRITE_SIZE'] * 1024.0
        factor = 1.0 if ('reduce' in k or 'prep' in k or 'adam' in k) else 2.0
        e['fetch_factor'] = factor
        e['hbm_bytes_corrected'] = factor * fb + wb
        e['hbm_bytes_raw'] = fb + wb
    out[k] = e
json.dump(out, open(f'profiles/{tag}_pmc.json', 'w'), indent=1)
for k, e in out.items():
    print(k, {c: (round(v) if isinstance(v, float) else v) for c, v in e.items() if c != 'launches_sampled'})
for r in rows[:8]:
    print(r['Name'][:80].ljust(80), r['Calls'], r['AverageNs'], r['Percentage'])
