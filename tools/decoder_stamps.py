"""Phase breakdown of the persistent decoder's forward kernel (speller_persist.hip): NABU_PERSIST_DEBUG bit 2 makes
block 0 stamp wall_clock64 at the phase boundaries of step L/2 into the status area of the Speller workspace.
Usage (GPU box): python tools/decoder_stamps.py [cfg3|cfg5] > profiles/<tag>_decoder_fwd_stamps.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('NABU_PERSIST_DEBUG', '4')

import numpy as np      # noqa: E402
import torch            # noqa: E402

import bench            # noqa: E402
from nabu_amd import _hip   # noqa: E402

NAMES = ['A gather [ctx|h]', 'A save + matrix product', 'A reduce + barrier + gates', 'barrier + B gather h',
         'B product + publish q', 'C gather q (+ alignments, conv features)', 'C scores', 'C statistics + partial context',
         'D gather partials', 'D combine + publish']


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
    args = bench.parse_args(['--workload', wl, '--no-cpu-baseline'])
    w = bench.make_workload(args, bench.make_server())
    for i in range(3):
        w.step(i)
    torch.cuda.synchronize()
    buf = [v for (d, t), v in _hip.Workspace._bufs.items() if t == 'speller'][0]
    st = buf[:256].view(torch.int32).cpu().numpy().astype(np.int64)
    x = st[16:27]
    print('persistent decoder forward, %s, block 0, step L/2 (us)' % wl)
    for i, n in enumerate(NAMES):
        print('  %-44s %6.2f' % (n, ((x[i + 1] - x[i]) & 0xffffffff) / 100.0))
    print('  %-44s %6.2f' % ('step', ((x[10] - x[0]) & 0xffffffff) / 100.0))
    y = st[16:32]
    if y[12] and y[13]:
        print('  inside "C scores": alignments gather %.2f, location features %.2f, scores %.2f' % (
            ((y[12] - y[6]) & 0xffffffff) / 100.0, ((y[13] - y[12]) & 0xffffffff) / 100.0, ((y[7] - y[13]) & 0xffffffff) / 100.0))


if __name__ == '__main__':
    main()
