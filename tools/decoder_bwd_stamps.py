"""Phase breakdown of the persistent decoder's BACKWARD kernel (speller_persist.hip, SPB_STAMP): NABU_PERSIST_DEBUG bit 2
makes block 0 stamp wall_clock64 at the phase boundaries of step L/2 into status[48..58] of the Speller workspace.
Usage (GPU box): NABU_SPELLER_PERSIST_BWD_LOC=2 python tools/decoder_bwd_stamps.py [cfg3|cfg5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('NABU_PERSIST_DEBUG', '4')

import numpy as np      # noqa: E402
import torch            # noqa: E402

import bench            # noqa: E402
from nabu_amd import _hip   # noqa: E402

NAMES = ['D1 gather carry (+ d features), alignments', 'D1 location features + carry share', 'D1 d alignment, d score',
         'D1 score backward (tanh, d features)', 'D1 publish d features / dq partial', 'D1b gather + dq block',
         'D2 gather dq, dh, cell backward', 'D3 gather dz', 'D3 product', 'D3 publish carry']


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg5'
    args = bench.parse_args(['--workload', wl, '--no-cpu-baseline'])
    w = bench.make_workload(args, bench.make_server())
    for i in range(3):
        w.step(i)
    torch.cuda.synchronize()
    buf = [v for (d, t), v in _hip.Workspace._bufs.items() if t == 'speller'][0]
    st = buf[:512].view(torch.int32).cpu().numpy().astype(np.int64)
    x = st[48:59]
    print('persistent decoder backward, %s, block 0, step L/2 (us); raw %s' % (wl, list(x)))
    for i in range(10):
        print('  %-2d %-48s %6.2f' % (i, NAMES[i] if i < len(NAMES) else '', ((x[i + 1] - x[i]) & 0xffffffff) / 100.0))
    print('  %-51s %6.2f' % ('step (stamps 0..10)', ((x[10] - x[0]) & 0xffffffff) / 100.0))


if __name__ == '__main__':
    main()
