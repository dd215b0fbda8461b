"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<name>.json.

usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

The two passes are separate runs of the same command (MI355X_MICROARCH.md: FETCH_SIZE and
WRITE_SIZE do not fit one pass).  Units and corrections as that guide prescribes for gfx950:
both counters are in KiB; FETCH_SIZE reports one half of the bytes actually read (128-byte
requests tallied at 64 bytes) and is doubled here; WRITE_SIZE is taken as is (checked on the
Adam kernel of the same run: 3 x 132 MB written -> 395.8 MB reported, 4 x 132 MB read -> 263.9 MB
reported)."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            agg.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']) * 1024.0)
    return agg


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {'units': 'bytes per launch (mean over the launches of the run)', 'fetch_correction': 2.0,
           'kernels': {}}
    for k in fetch:
        f = [2.0 * x for x in fetch[k]]
        w = write.get(k, [0.0])
        out['kernels'][k] = {'launches': len(f), 'fetch_bytes': sum(f) / len(f), 'write_bytes': sum(w) / len(w),
                             'traffic_bytes': sum(f) / len(f) + sum(w) / len(w)}
    rec = [v for k, v in out['kernels'].items() if 'lstm_persist' in k or 'lstm_mx' in k]     # the recurrent kernels
    n = sum(v['launches'] for v in rec)
    out['lstm_persist_traffic_bytes_per_launch'] = sum(v['traffic_bytes'] * v['launches'] for v in rec) / max(n, 1)
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps({k[:60]: v for k, v in out['kernels'].items() if v['traffic_bytes'] > 1e8}, indent=1))
    print('lstm_persist traffic per launch: %.3f GB' % (out['lstm_persist_traffic_bytes_per_launch'] / 1e9))


if __name__ == '__main__':
    main()
