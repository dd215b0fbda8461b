import sys; sys.path.insert(0,'.')
import torch
from nabu_amd import ops
def bench(name, ta, tb, M, N, K, reps=6, prec='f32'):
    a = torch.randn((K, M) if ta else (M, K), device='cuda')
    b = torch.randn((N, K) if tb else (K, N), device='cuda')
    c = torch.empty(M, N, device='cuda')
    ops.gemm(a, b, c, ta, tb, precision=prec); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(a, b, c, ta, tb, precision=prec)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    print('%-26s %s%s M=%6d N=%5d K=%6d: %.3f ms  %6.1f TF/s' % (name, 'T' if ta else 'N', 'T' if tb else 'N', M, N, K, ms, 2*M*N*K/ms/1e9))
    return ms
tot = 0
for prec in sys.argv[1:] or ['f32']:
    print(prec); tot = 0
    for l, (T, D) in enumerate([(1000, 40), (500, 2048), (250, 2048), (125, 2048)]):
        BT = 32 * T
        tot += 2 * bench('L%d fwd x.Wx (per dir)' % l, 0, 0, BT, 2048, D, prec=prec)
        if l: tot += 2 * bench('L%d dx = dz.Wx^T (per dir)' % l, 0, 1, BT, D, 2048, prec=prec)
        tot += 2 * bench('L%d dWx = x^T.dz (per dir)' % l, 1, 0, D, 2048, BT, prec=prec)
        tot += 2 * bench('L%d dWh = h^T.dz (per dir)' % l, 1, 0, 512, 2048, BT, prec=prec)
    print('sum %.2f ms' % tot)
