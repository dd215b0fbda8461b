"""CPU: the generation arithmetic of the tagged backward exchange ring (nabu_amd/csrc/lstm_persist_mxh.h): a reader
must never mistake what a slot held before for the piece it waits for.  Iteration it (0-based count of backward steps)
publishes into slot it & 1 with tag (it >> 1) & 1; the reader of iteration it polls slot (it - 1) & 1 for tag
((it - 1) >> 1) & 1; the ring starts as 0xFF bytes (every last bit 1).  Simulated with every interleaving of 'stale' and
'fresh' the protocol allows: a slot holds either the piece of iteration it - 1 or — not yet overwritten — that of it - 3
(or the initial bytes)."""


def slot_of(it):
    return it & 1


def tag_of(it):
    return (it >> 1) & 1


def test_a_stale_piece_never_carries_the_awaited_tag():
    INITIAL = 1                                     # 0xFFFFFFFF: last bit set
    for T in range(1, 200):
        for it in range(1, T):                      # the reader of iteration it waits for the piece of it - 1
            want = tag_of(it - 1)
            assert slot_of(it - 1) == slot_of(it - 3)                  # what may still be there was written two uses ago ...
            stale = tag_of(it - 3) if it - 3 >= 0 else INITIAL
            assert stale != want, (T, it)                              # ... and carries the other tag
        # a writer of iteration it overwrites the piece of it - 2: by then its reader (iteration it - 1) has finished
        # polling it — the writer polled that reader's own publish of it - 1, issued behind the reader's barrier
        # (and every reuse of a slot flips its tag)
        for it in range(2, T):
            assert slot_of(it) == slot_of(it - 2) and tag_of(it) != tag_of(it - 2)


def test_tag_bit_costs_at_most_one_unit_in_the_last_place():
    import numpy as np
    rng = np.random.default_rng(0)
    x = (rng.normal(size=100000) * np.exp(rng.normal(size=100000) * 4)).astype(np.float32)
    bits = x.view(np.uint32)
    for tag in (0, 1):
        y = ((bits & np.uint32(0xFFFFFFFE)) | np.uint32(tag)).view(np.float32)
        ulp = np.spacing(np.abs(x))
        assert np.all(np.abs(y.astype(np.float64) - x.astype(np.float64)) <= ulp.astype(np.float64))
    lo = (bits & np.uint32(0xFFFFFFFE)).view(np.float32).astype(np.float64)
    hi = (bits | np.uint32(1)).view(np.float32).astype(np.float64)
    # low in even generations, high in odd ones: the two errors of a value bracket it (no bias over time)
    assert np.all(np.abs(lo) <= np.abs(x.astype(np.float64))) and np.all(np.abs(hi) >= np.abs(x.astype(np.float64)))
