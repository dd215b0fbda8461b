"""Bandwidth of the operand packs against plain device copies of the same bytes (is there headroom in the pack kernels?).
Run on the GPU box: python tools/experiments/pack_bw.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nabu_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    for name, R, C in (('L0 dz [32000 x 4096]', 32000, 4096), ('L1 x [16000 x 2048]', 16000, 2048), ('L1 dz [16000 x 4096]', 16000, 4096)):
        x = torch.randn(R, C, device='cuda')
        y = torch.empty_like(x)
        t_copy = timeit(lambda: y.copy_(x))
        z = torch.empty(R * C, device='cuda', dtype=torch.float32)
        t_fill = timeit(lambda: z.zero_())
        gb = 2 * R * C * 4 / 1e9
        line = '%-24s copy %.3f ms = %.2f TB/s; fill %.3f ms = %.2f TB/s' % (name, t_copy, gb / t_copy, t_fill, gb / 2 / t_fill)
        for planes in (2,):
            pr, pc = ops.PackedOperand(R, C, planes, 'cuda'), ops.PackedOperand(C, R, planes, 'cuda')
            t_rows = timeit(lambda: ops.pk_pack(pr, x))
            t_cols = timeit(lambda: ops.pk_pack(pc, x, True))
            line += '; planes %d: rows pack %.3f ms = %.2f TB/s, transposed pack %.3f ms = %.2f TB/s (with their maxima passes)' % (
                planes, t_rows, gb / t_rows, t_cols, gb / t_cols)
        print(line, flush=True)


if __name__ == '__main__':
    main()
