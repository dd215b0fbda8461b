"""Summarise a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` pass into
profiles/<name>.json: per kernel, the fraction of SIMD cycles the matrix pipe was busy and the
effective shader clock.

    python tools/pmc_mfma_summary.py <counter_collection.csv> <out.json>

Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles summed over the 1024 SIMDs;
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (one GRBM each), so cycles per XCD = value / 8:
    mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024),  effective clock = GUI_ACTIVE / 8 / duration."""
import collections
import csv
import json
import sys

NSIMD, NXCD = 1024, 8


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n, dur = collections.Counter(), collections.defaultdict(float)
    for r in csv.DictReader(open(sys.argv[1])):
        k = r['Kernel_Name']
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            n[k] += 1
            dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    out = {}
    for k, v in agg.items():
        g, m = v.get('GRBM_GUI_ACTIVE', 0.0) / NXCD, v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        if g <= 0 or dur[k] < 1e5:
            continue
        out[k] = {'launches': n[k], 'mfma_util': round(m / (g * NSIMD), 4), 'effective_clock_ghz': round(g / dur[k], 3),
                  'ms_total': round(dur[k] / 1e6, 3)}
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]['ms_total'])[:8]:
        print('%-62s util %5.1f %%  clock %.2f GHz  %.2f ms' % (k[:62], 100 * v['mfma_util'], v['effective_clock_ghz'], v['ms_total']))


if __name__ == '__main__':
    main()
