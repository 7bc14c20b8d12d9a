"""GPU (-m gpu): two device-resident batches in flight inside one encoder (mjh_set_inflight, the library's default since round 6):
consecutive mjh_encode_device calls on the encoder's own stream alternate between two complete buffer sets and overlap on the
chip.  Different inputs in consecutive calls must never meet: every call's files against the CPU oracle."""
import numpy as np
import pytest

import mozjpeg_amd as M
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [dict(quality=75, baseline=True), dict(quality=85, sample=(2, 2)), dict(quality=90, baseline=True, sample=(1, 1), restart=1),
                                dict(quality=75, baseline=True, notrellis=True)])
def test_consecutive_calls_with_two_batches_in_flight_never_mix(kw):
    import torch
    w, h, B = 531, 297, 5
    sets = [np.stack([O.synthetic_frame(w, h, 900 + 10 * s + i) for i in range(B)]) for s in range(3)]
    refs = [[O.encode(O.make_params(w, h, **kw), f) for f in fs] for fs in sets]
    dev = [torch.from_numpy(fs).cuda() for fs in sets]
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=B)
    # back to back, no synchronisation in between: call k + 1 is queued while call k runs; the accessors refer to the latest call
    order = [0, 1, 2, 1, 0, 0, 2, 1, 2, 0, 1]
    for n, s in enumerate(order):
        enc.encode_tensor(dev[s], stream="own")
        if n % 3 == 2 or n == len(order) - 1:
            assert [enc.get_jpeg(i) for i in range(B)] == refs[s], "call %d (input set %d)" % (n, s)
    # one batch at a time gives the same files, and switching back does too
    enc.set_inflight(1)
    for s in (2, 0):
        enc.encode_tensor(dev[s], stream="own")
        assert [enc.get_jpeg(i) for i in range(B)] == refs[s]
    enc.set_inflight(2)
    for s in (1, 2, 0):
        enc.encode_tensor(dev[s], stream="own")
    assert [enc.get_jpeg(i) for i in range(B)] == refs[0]
    enc.close()
