"""Pin the oracle against golden vectors produced by the REFERENCE's own source files
(oracle/gen_golden.py executes util.py / mixture.py / gaussian.py from /root/reference)."""
import os

import numpy as np
import torch

from oracle import mulaw as M
from oracle import wavenet_oracle as O


def test_mulaw_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, 'mulaw_golden.npz'))
    x = g['x']
    assert np.array_equal(M.mulaw_quantize(x), g['quantized'])            # integer indices: bit-exact
    assert np.array_equal(M.mulaw_quantize(g['x64']), g['quantized64'])
    assert np.array_equal(M.mulaw(x), g['mulaw'])
    assert np.array_equal(M.inv_mulaw_quantize(np.arange(256)), g['inv_q_all'])
    assert np.array_equal(M.inv_mulaw(np.linspace(-1, 1, 4097).astype(np.float32)), g['inv_mulaw'])
    assert int(g['q0']) == 127 and int(M.mulaw_quantize(np.float64(0.0))) == 127   # wavenet.py:361,434
    assert float(g['m0']) == 0.0


def test_mulaw_roundtrip_properties():
    x = np.linspace(-1, 1, 100001)
    q = M.mulaw_quantize(x)
    assert q.min() == 0 and q.max() == 255
    assert np.all(np.diff(q) >= 0)                                         # monotone
    xr = M.inv_mulaw_quantize(q)
    # decode(encode(x)) stays inside the bin that contains x
    lo = M.inv_mulaw_quantize(np.maximum(q - 1, 0)); hi = M.inv_mulaw_quantize(np.minimum(q + 1, 255))
    assert np.all(xr >= lo - 1e-7) and np.all(xr <= hi + 1e-7)
    # decode lands on the bin's lower edge (truncating quantiser), so re-encoding is within one bin
    assert np.all(np.abs(M.mulaw_quantize(M.inv_mulaw_quantize(np.arange(256)).astype(np.float64)) - np.arange(256)) <= 1)


def test_mol_loss_and_sampler(golden_dir):
    g = np.load(os.path.join(golden_dir, 'mol_golden.npz'))
    y_hat, y = torch.from_numpy(g['y_hat']), torch.from_numpy(g['y'])
    for nc, lsm in ((65536, float(np.log(1e-14))), (256, -7.0)):
        l = O.discretized_mix_logistic_loss(y_hat, y, num_classes=nc, log_scale_min=lsm)
        np.testing.assert_allclose(l.numpy(), g['loss_%d' % nc], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(float(l.sum()), float(g['loss_sum_%d' % nc]), rtol=1e-6)
    s = O.sample_from_discretized_mix_logistic(y_hat, torch.from_numpy(g['u1']), torch.from_numpy(g['u2']),
                                               log_scale_min=float(np.log(1e-14)))
    np.testing.assert_allclose(s.numpy(), g['sample'], rtol=0, atol=1e-7)


def test_gaussian_loss_and_sampler(golden_dir):
    g = np.load(os.path.join(golden_dir, 'gaussian_golden.npz'))
    y_hat, y = torch.from_numpy(g['y_hat']), torch.from_numpy(g['y'])
    for tag, lsm in (('a', float(np.log(1e-7))), ('b', float(np.log(9.1188196e-4)))):
        for cdf in (False, True):
            l = O.gaussian_mle_loss(y_hat, y, lsm, 65536, cdf)
            np.testing.assert_allclose(l.numpy(), g['loss_cdf%d_%s' % (int(cdf), tag)], rtol=1e-6, atol=1e-6)
    s = O.sample_from_gaussian(y_hat, torch.from_numpy(g['eps']), float(np.log(1e-7)))
    np.testing.assert_allclose(s.numpy(), g['sample'], rtol=0, atol=1e-7)
