#!/usr/bin/env python3
"""
Every number of bench.py's `roofline` object, recomputed from the files under profiles/ alone.

    python scripts/roofline_from_profiles.py [tag]          # tag defaults to the newest profiles/rNN_*

Inputs (written on the GPU box by scripts/collect_profiles.sh + scripts/summarize_profiles.py):
  profiles/<tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command: average duration
  profiles/<tag>_pmc.json                 per-kernel, per-launch PMC counters of the same command, one pass per
                                          counter group (FETCH_SIZE / WRITE_SIZE / SQ_*), as MI355X_MICROARCH.md
                                          (HBM and rocprofv3 sections) prescribes
bench.py imports `build_roofline` from this file and feeds it its live HIP-event durations instead of the rocprof
averages, so the two can only differ by the timing source (stated in `timing_source`).

Work model.  Workload cfg 2: n = 5 qubits, Net40-2-20-2, batch 1024 per launch.
  R = E + P = 300 + 1800 rotation gates, G = R + 600 CNOTs, S = 16 * 2^n bytes per state.
  ALGORITHMIC flops (6 flops per rotation gate per amplitude, CNOT 0):
      forward        6 * 2^n * R
      training       3 sweeps (forward, U^dagger on psi, U^dagger on lambda) + one <lambda|sigma|psi> pass per rotation
                     at 4 flops per amplitude = (18 + 4) * 2^n * R
  EXECUTED flops (what the kernels issue: fused SU(2) = 16 FMA per amplitude pair, RX = 8): reported beside it.
  Gate-streaming bytes (what an HBM-streaming simulator would move): forward 2*S*G + S,
  training S*(6G + 2R) + S -- kept as `effective_streaming_GBs`, NOT as a roofline: the state is chip-resident.
Peaks (MI355X_MICROARCH.md): HBM 8 TB/s; fp64 vector 78.6 TFLOP/s (half the 157.3 TFLOP/s fp32 vector rate; the fp64
matrix rate is the same on gfx950); LDS ~150 TB/s aggregate for ds_read_b64/b128 at 2.4 GHz.
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0
FP64_VECTOR_PEAK_TFLOPS = 78.6
LDS_PEAK_TBS = 150.0
SIMDS = 1024
CLOCK_GHZ = 2.4
N_QUBITS, NET, BATCH = 5, (40, 2, 20, 2), 1024


def circuit_counts(n=N_QUBITS, net=NET):
    bd, bl, td, tl = net
    E = (bd + td) * n
    blk = bd * bl + td * tl
    R = E + 3 * n * blk
    G = R + n * blk
    S = 16 * (1 << n)
    amps = 1 << n
    pairs = amps // 2
    fma_fwd = n * blk * pairs * 16 + E * pairs * 8
    fma_bwd = 2 * fma_fwd + n * blk * pairs * 12 + E * pairs * 4
    return dict(E=E, blk=blk, R=R, G=G, S=S,
                alg_flops_fwd=6 * amps * R, alg_flops_train=22 * amps * R,
                exec_flops_fwd=2 * fma_fwd, exec_flops_train=2 * (fma_fwd + fma_bwd),
                bytes_fwd=2 * S * G + S, bytes_train=S * (6 * G + 2 * R) + S,
                io_bytes_train=8 * (100 + 2 + 1) + 8 * E)          # inputs + target + materialised grad_x, per sample


def _pick(d, *subs):
    for k in d:
        if all(s in k for s in subs):
            return k, d[k]
    return None, None


def build_roofline(train_kernel_ms, fwd_kernel_ms, pmc, timing_source, batch=BATCH, train_kernel_hint='bwd_'):
    """pmc: {kernel name: {counter: per-launch value}} or None.  Durations in ms of ONE launch of the dominant
    (forward + adjoint backward) kernel and of the forward-only kernel."""
    cc = circuit_counts()
    t = train_kernel_ms * 1e-3
    tf = fwd_kernel_ms * 1e-3 if fwd_kernel_ms else None
    kname, kc = (None, None)
    if pmc:
        for hint in ('bwd_ztri_kernel<5', 'bwd_tri_kernel<5>', 'bwd_pair_kernel<5>', 'bwd_kernel<5', train_kernel_hint):
            kname, kc = _pick(pmc, hint)
            if kc:
                break
    alg = cc['alg_flops_train'] * batch
    ach = alg / t / 1e12
    roof = {
        "bound": "fp64_vector",
        "kernel": kname or "qhea::bwd_*_kernel<5> (fused forward sweep + MSE residual + adjoint reverse sweep)",
        "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VECTOR_PEAK_TFLOPS,
        "launch_ms": train_kernel_ms, "timing_source": timing_source,
        "algorithmic_flops_per_launch": alg,
        "algorithmic_flops_per_unit": cc['alg_flops_train'],
        "units_per_launch": batch,
        "executed_flops_per_launch": cc['exec_flops_train'] * batch,
        "executed_frac": cc['exec_flops_train'] * batch / t / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
        "traffic": None, "traffic_source": None,
        "effective_streaming_GBs": cc['bytes_train'] * batch / t / 1e9,
        "note": "bound = fp64 vector arithmetic (the state is wave-resident: HBM carries inputs and outputs only, see "
                "`hbm`); achieved = SURVEY.md 8(d) algorithmic flops (22 * 2^n * R per training sample) x samples per "
                "launch / duration of the kernel alone.  effective_streaming_GBs is the gate-streaming byte count of "
                "BASELINE.md section 2 over the same time -- the traffic an HBM-streaming simulator would need, kept "
                "for comparison only, not a fraction of anything.",
    }
    if tf:
        af = cc['alg_flops_fwd'] * batch / tf / 1e12
        roof["forward_kernel"] = {"launch_ms": fwd_kernel_ms, "achieved": af, "frac": af / FP64_VECTOR_PEAK_TFLOPS,
                                  "algorithmic_flops_per_launch": cc['alg_flops_fwd'] * batch,
                                  "effective_streaming_GBs": cc['bytes_fwd'] * batch / tf / 1e9}
    if kc:
        if 'hbm_bytes_corrected' in kc:
            hb = kc['hbm_bytes_corrected']
            roof["traffic"] = hb
            roof["hbm"] = {"achieved": hb / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hb / t / 1e9 / HBM_PEAK_GBS,
                           "irreducible_io_bytes_per_launch": cc['io_bytes_train'] * batch,
                           "traffic_over_irreducible": hb / (cc['io_bytes_train'] * batch),
                           "note": "PMC bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports half "
                                   "of a wide coalesced read, MI355X_MICROARCH.md HBM section), separate --pmc passes"}
        if 'SQ_INSTS_LDS' in kc:
            lds_instr = kc['SQ_INSTS_LDS']
            # upper bound on LDS bytes: every LDS wave-instruction moves at most 16 B per lane (ds_read/write_b128);
            # the cross-lane ds_swizzle / ds_bpermute move 4 B per lane and no memory
            ub = lds_instr * 64 * 16
            roof["lds"] = {"SQ_INSTS_LDS": lds_instr, "bytes_upper_bound": ub, "achieved_upper_bound": ub / t / 1e12,
                           "peak": LDS_PEAK_TBS, "unit": "TB/s", "frac_upper_bound": ub / t / 1e12 / LDS_PEAK_TBS}
        if 'SQ_WAVE_CYCLES' in kc and kc['SQ_WAVE_CYCLES']:
            wc = kc['SQ_WAVE_CYCLES']
            issue = {k: kc[k] for k in ('SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAVE_CYCLES',
                                        'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES') if k in kc}
            for k in ('SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY'):
                if k in kc:
                    issue[k + '_over_WAVE_CYCLES'] = kc[k] / wc
            if 'SQ_INSTS_VALU' in kc:
                # chip-level vector issue: every SIMD can start one wave-instruction per 4 clocks (fp64: 16 lanes/clock)
                slots = t * CLOCK_GHZ * 1e9 / 4 * SIMDS
                issue['valu_issue_slots_used'] = kc['SQ_INSTS_VALU'] / slots
            roof["issue"] = issue
        roof["traffic_source"] = "profiles/*_pmc.json (rocprofv3 --pmc passes of this command)"
    return roof


def newest_tag():
    """Newest set under profiles/: highest round; within a round the plain `rNN` set is the final one, suffixed sets
    (`rNNa` ...) are earlier snapshots of that round."""
    tags = [m.groups() for m in (re.match(r'r(\d+)([a-z]*)_pmc\.json$', os.path.basename(p))
                                 for p in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json'))) if m]
    if not tags:
        return None
    num, suffix = max(tags, key=lambda t: (int(t[0]), t[1] == '', t[1]))
    return f'r{num}{suffix}'


def load_pmc(tag=None):
    tag = tag or newest_tag()
    if not tag:
        return None, None
    path = os.path.join(ROOT, 'profiles', f'{tag}_pmc.json')
    if not os.path.exists(path):
        return None, None
    try:
        return json.load(open(path)), os.path.relpath(path, ROOT)
    except (OSError, ValueError):
        return None, None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else newest_tag()
    pmc, src = load_pmc(tag)
    stats = os.path.join(ROOT, 'profiles', f'{tag}_bench_kernel_stats.csv')
    rows = {r['Name']: r for r in csv.DictReader(open(stats))}
    k_train, r_train = _pick(rows, 'bwd_', '<5')
    k_fwd, r_fwd = None, None
    for hint in ('fwd_split_kernel<5', 'fwd_zyz_kernel<5', 'fwd_kernel<5'):
        k_fwd, r_fwd = _pick(rows, hint)
        if r_fwd:
            break
    roof = build_roofline(float(r_train['AverageNs']) * 1e-6, float(r_fwd['AverageNs']) * 1e-6 if r_fwd else None, pmc,
                          f"rocprofv3 --kernel-trace --stats average of {r_train['Calls']} launches ({os.path.basename(stats)})")
    if src:
        roof['traffic_source'] = src
    print(json.dumps(roof, indent=1))


if __name__ == '__main__':
    main()
