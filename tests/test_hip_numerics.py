"""include/glabc_numerics.h evaluated on the gfx950 device equals its evaluation on the host,
bit for bit -- the premise of every HIP-vs-oracle parity claim.  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _device(hip, op, words):
    w = torch.from_numpy(words.view(np.int32)).cuda()
    n = words.size if op < 4 else words.size // 2
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    assert hip.glabc_selftest_numerics(op, w.data_ptr(), out.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)


def test_exp_log_sincos_device_equals_host(hip, oracle):
    rng = np.random.default_rng(0)
    n = 1 << 22
    # exp over its whole interesting range + specials
    x = np.concatenate([rng.uniform(-110, 90, n).astype(np.float32),
                        np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 88.72283, 88.72284, -103.9, -104.1], np.float32)])
    h = np.empty_like(x)
    oracle.oracle_expf_v(x.ctypes.data, x.size, h.ctypes.data)
    d = _device(hip, 0, x.view(np.uint32))
    ok = (d == h.view(np.uint32)) | (np.isnan(h) & np.isnan(d.view(np.float32)))
    assert ok.all()
    # log: every accept-uniform (k 2^-24), a wide sample, subnormals, specials
    x = np.concatenate([(np.arange(0, 1 << 24, dtype=np.float64) * 2.0 ** -24).astype(np.float32),
                        np.exp(rng.uniform(-104, 88, n)).astype(np.float32),
                        np.array([np.nan, np.inf, -np.inf, -1.0, -0.0, 1e-45, 1e-40, 3e38], np.float32)])
    h = np.empty_like(x)
    oracle.oracle_logf_v(x.ctypes.data, x.size, h.ctypes.data)
    d = _device(hip, 1, x.view(np.uint32))
    ok = (d == h.view(np.uint32)) | (np.isnan(h) & np.isnan(d.view(np.float32)))
    assert ok.all()
    # sin / cos of 2 pi u for every possible u
    u = (np.arange(1 << 24, dtype=np.float64) * 2.0 ** -24).astype(np.float32)
    s = np.empty_like(u)
    c = np.empty_like(u)
    oracle.oracle_sincos2pi_v(u.ctypes.data, u.size, s.ctypes.data, c.ctypes.data)
    assert np.array_equal(_device(hip, 2, u.view(np.uint32)), s.view(np.uint32))
    assert np.array_equal(_device(hip, 3, u.view(np.uint32)), c.view(np.uint32))


def test_normal_pair_device_equals_host(hip, oracle):
    rng = np.random.default_rng(1)
    n = 1 << 22
    a = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    a[:4] = [0, 0xffffffff, 0xffffff80, 1]                    # u1 extremes: largest radius, log(1) = 0
    z0 = np.empty(n, np.float32)
    z1 = np.empty(n, np.float32)
    oracle.oracle_normal_pair_v(a.ctypes.data, b.ctypes.data, n, z0.ctypes.data, z1.ctypes.data)
    words = np.stack([a, b], axis=1).reshape(-1).copy()
    assert np.array_equal(_device(hip, 4, words), z0.view(np.uint32))
    assert np.array_equal(_device(hip, 5, words), z1.view(np.uint32))


def test_sqrt_normal_is_correctly_rounded_for_every_input(hip):
    """glabc_sqrtf_normal == exactly rounded sqrt for +-0 and EVERY float in [2^-64, FLT_MAX]
    (its documented domain; Box-Muller's argument is 0 or in [1e-7, 46])."""
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    assert hip.glabc_selftest_sqrt(0x1f800000, 0x7f7fffff, bad.data_ptr(), None) == 0
    assert hip.glabc_selftest_sqrt(0x00000000, 0x00000000, bad.data_ptr(), None) == 0
    assert hip.glabc_selftest_sqrt(0x80000000, 0x80000000, bad.data_ptr(), None) == 0      # -0 -> -0
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
