#!/usr/bin/env python
"""MFMA instruction counters per kernel from one rocprofv3 PMC pass of bench.py:

    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out/pmc_mfma -- python bench.py --steps 1 ...
    python tools/pmc_mfma.py --dir out/pmc_mfma --out profiles/r3_mfma_counters.json

SQ_INSTS_VALU_MFMA_MOPS_BF16 counts bf16 MFMA work in units of 512 FLOP (MI355X_MICROARCH.md, rocprofv3 PMC section: gfx950 has no
derived-metric table, so the raw counter is reported together with the unit check below).  Unit check: window_attn_fwd_mfma_k
issues exactly 16 v_mfma_f32_32x32x16_bf16 (32768 FLOP each) per (window, head) workgroup, so counter * 512 / (16 * 32768 * grid)
must be 1; the measured ratio is stored as `calibration`.  MFMA utilisation of a kernel = its MFMA FLOP / (duration * 2.5 PFLOP/s),
with the duration from GRBM_GUI_ACTIVE.  That counter is SUMMED over the 8 XCDs (each has its own GRBM): against rocprofv3's
kernel-trace durations of the same kernels it reads 17.4 - 17.9 cycles per ns for every kernel longer than 100 us = 8 x 2.2 GHz, so
the per-dispatch busy time is counter / 8 / clock.  Short kernels read higher (fixed per-dispatch overhead of the counter mode).
`--from-json` recomputes the derived columns of an existing output (the raw counters are kept in it).
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def short(name):
    name = re.sub(r'^void\s+', '', name.strip())
    return name.split('(')[0].strip()[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', default=None)
    ap.add_argument('--out', default='profiles/r3_mfma_counters.json')
    ap.add_argument('--from-json', default=None)
    a = ap.parse_args()
    if a.from_json:
        old = json.load(open(a.from_json))
        finish(old['kernels'], old.get('calibration'), a.out)
        return
    files = glob.glob(os.path.join(a.dir, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit(f'no *counter_collection.csv under {a.dir}')
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    grid = {}
    for f in files:
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                k = short(row['Kernel_Name'])
                acc[k][row['Counter_Name']] += float(row['Counter_Value'])
                calls[k].add(row.get('Dispatch_Id', row.get('Correlation_Id', '')))
                if 'Grid_Size' in row and row['Grid_Size']:
                    grid.setdefault(k, []).append((row.get('Dispatch_Id', ''), int(float(row['Grid_Size'])), int(float(row.get('Workgroup_Size', 64) or 64))))
    rows = []
    for k, c in acc.items():
        mops = c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0)
        if mops <= 0:
            continue
        n = max(1, len(calls[k]))
        rows.append(dict(kernel=k, dispatches=n, mfma_mops_bf16_per_dispatch=mops / n, mfma_gflop_per_dispatch=mops * 512 / n / 1e9,
                         sq_busy_cycles_per_dispatch=c.get('SQ_BUSY_CYCLES', 0.0) / n, grbm_gui_active_per_dispatch=c.get('GRBM_GUI_ACTIVE', 0.0) / n))
    rows.sort(key=lambda r: -r['mfma_gflop_per_dispatch'] * r['dispatches'])
    cal = None
    for r in rows:
        if r['kernel'].startswith('window_attn_fwd_mfma_k') and r['kernel'] in grid:
            seen = {}
            for did, g, wg in grid[r['kernel']]:
                seen[did] = g // max(1, wg)
            wgs = sum(seen.values())
            expect = 16 * 32768 * wgs / 1e9
            got = r['mfma_gflop_per_dispatch'] * r['dispatches']
            cal = dict(kernel=r['kernel'], workgroups=wgs, expected_gflop=expect, counted_gflop=got, ratio=got / expect if expect else None)
            break
    finish(rows, cal, a.out)


XCDS = 8
CLOCK_HZ = 2.4e9           # peak engine clock: the 2.5 PFLOP/s dense bf16 figure is quoted at it


def finish(rows, cal, out_path):
    class A:
        out = out_path
    a = A()
    for r in rows:
        cyc = r['grbm_gui_active_per_dispatch'] / XCDS
        r['busy_cycles_per_xcd'] = round(cyc, 1)
        # 2.5 PFLOP/s at 2.4 GHz = 1041.7 kFLOP per GPU cycle: utilisation in MFMA-issue cycles, independent of the clock the run held
        r['mfma_util_vs_2p5pf'] = round(r['mfma_gflop_per_dispatch'] * 1e9 / (cyc * 2.5e15 / CLOCK_HZ), 4) if cyc else None
    out = dict(source='rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE of bench.py --steps 1', unit='counter x 512 FLOP',
               calibration=cal, kernels=rows)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, 'w'), indent=1)
    print('calibration:', cal)
    for r in rows[:25]:
        print(f"{r['kernel'][:70]:70s} x{r['dispatches']:4d} {r['mfma_gflop_per_dispatch']:10.2f} GFLOP/dispatch  util {r['mfma_util_vs_2p5pf']}")


if __name__ == '__main__':
    main()
