"""Times the packed bf16-plane GEMM (gemm_pk.hip) and its pack kernels at the product shapes of a cfg2 / cfg5
training step, next to the exact-fp32 MFMA kernel.  Run on the GPU box:
    python tools/experiments/gemm_pk_bench.py [--planes 3] [--reps 20]
Prints one line per shape: ms, effective TF/s (2MNK / time), MFMA TF/s (x6 for planes = 3, x3 for planes = 2 = f16x3;
the pack times of planes = 2 include the row-maximum pass)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nabu_amd import ops  # noqa: E402


def timeit(fn, reps):
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
    ap = argparse.ArgumentParser()
    ap.add_argument('--planes', type=int, default=3)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--fp32', type=int, default=1)
    ap.add_argument('--shapes', default='cfg2')
    ap.add_argument('--data', default='randn', choices=['randn', 'ones', 'pow2', 'sparse'],
                    help='operand values (the matrix pipe\'s power, hence its clock, depends on them): N(0,1); all 1; random signed powers of two (one mantissa bit pattern); N(0,1) with 90 %% zeros')
    a = ap.parse_args()
    shapes = {
        'cfg2': [  # (name, M, N, K)
            ('L1 fwd x.[Wfw|Wbw]', 16000, 4096, 2048), ('L1 dx dz.W^T', 16000, 2048, 8192),
            ('L1 dWx x^T.dz', 2048, 4096, 16000), ('L1 dWh', 512, 2048, 16000),
            ('L2 fwd', 8000, 4096, 2048), ('L2 dx', 8000, 2048, 8192), ('L2 dWx', 2048, 4096, 8000),
            ('L3 fwd', 4000, 4096, 2048), ('L3 dx', 4000, 2048, 8192), ('L3 dWx', 2048, 4096, 4000),
            ('L0 dWh', 512, 2048, 32000)],
        'cfg5': [('fwd', 51200, 4096, 2048), ('dx', 51200, 2048, 8192), ('dWx', 2048, 4096, 51200)],
        'sq': [('4096^3', 4096, 4096, 4096), ('8192^3', 8192, 8192, 8192)],
    }[a.shapes]
    torch.manual_seed(0)
    tot = tot32 = totpack = 0.0
    for name, M, N, K in shapes:
        def draw(r, c):
            if a.data == 'ones':
                return torch.ones(r, c, device='cuda')
            x = torch.randn(r, c, device='cuda')
            if a.data == 'pow2':
                return torch.sign(x) * torch.exp2(torch.floor(torch.log2(x.abs() + 1e-6)))
            if a.data == 'sparse':
                return x * (torch.rand(r, c, device='cuda') < 0.1)
            return x
        A, B = draw(M, K), draw(N, K)
        pa, pb = ops.PackedOperand(M, K, a.planes, 'cuda'), ops.PackedOperand(N, K, a.planes, 'cuda')
        tp = timeit(lambda: (ops.pk_pack(pa, A), ops.pk_pack(pb, B)), 5)
        At = A.t().contiguous()
        tpt = timeit(lambda: ops.pk_pack(pa, At, True), 5)
        C = torch.empty(M, N, device='cuda')
        t = timeit(lambda: ops.gemm_pk(pa, pb, C, a.planes), a.reps)
        t32 = 0.0
        if a.fp32:
            t32 = timeit(lambda: ops.gemm(A, B, C, False, True, precision='f32'), max(2, a.reps // 4))
        fl = 2.0 * M * N * K
        print('%-22s M=%6d N=%5d K=%6d  pk %7.3f ms  %7.1f TF/s eff  %7.1f TF/s mfma | pack A+B %6.3f ms, A^T %6.3f ms '
              '(%5.2f TB/s) | fp32 %7.3f ms %6.1f TF/s' %
              (name, M, N, K, t, fl / t * 1e-9, fl * {3: 6, 2: 3, 1: 1}[a.planes] / t * 1e-9, tp, tpt,
               M * K * (4 + 2 * a.planes) / tpt * 1e-9, t32, fl / t32 * 1e-9 if t32 else 0.0), flush=True)
        tot += t
        tot32 += t32
        totpack += tp
    print('sum: pk %.3f ms, fp32 %.3f ms, pack(A+B) %.3f ms' % (tot, tot32, totpack))


if __name__ == '__main__':
    main()
