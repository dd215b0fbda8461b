"""200-step clip + Adam loss trajectory of a shrunken cfg2 (Listener 3 pyramidal + 1 BLSTM x 64 units, DNN decoder, CTC;
16 utterances x 1024 frames x 40: every layer has >= 2048 frames, so the packed f16x3 / bf16x6 products are the ones
that run) from the float64 oracle -> tests/golden/cfg2_traj200.npz.

    python tests/golden/make_traj200.py [steps=200]        # ~10 minutes of CPU

PROVENANCE: oracle/nabu_oracle.py (PARITY UNPINNED against TF, see make_golden.py); weights and batches are pure
functions of the seeds below, so the fixture stores the losses only."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import make_golden as G                                              # noqa: E402
from nabu_amd.processing.synthetic import SyntheticData              # noqa: E402

B, T, D, H, C, NL, SEED = 16, 1024, 40, 64, 40, 3, 7234


def setup():
    names, E = G.encoder_names('Listener', NL, D, H)
    names += G.ctc_decoder_names(E, C)
    w = G.draw_weights(names)
    data = SyntheticData(B, T, D, min_frames=int(0.6 * T), min_labels=10, max_labels=40, time_reduction=8, seed=SEED)
    return w, data


if __name__ == '__main__':
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    w, data = setup()
    t0 = time.time()
    losses, _, _ = G.trajectory(w, data, lambda ww, b: G.step_ctc(ww, b, 'Listener', NL), steps)
    print('%d steps in %.1f s; loss %.6f -> %.6f' % (steps, time.time() - t0, losses[0], losses[-1]))
    np.savez_compressed(os.path.join(HERE, 'cfg2_traj200.npz'), meta=np.array([B, T, D, H, NL, C, steps, SEED]), losses=losses)
