"""include/glabc_numerics.h on the CPU: Philox known answers, accuracy of the f32
elementary functions against float64 libm, and the u32 -> uniform conversions."""
import numpy as np


def _philox(L, ctr, key):
    c = np.array(ctr, np.uint32)
    k = np.array(key, np.uint32)
    o = np.zeros(4, np.uint32)
    L.oracle_philox4x32_10(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return [int(v) for v in o]


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    assert _philox(oracle, [0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox(oracle, [0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox(oracle, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def _ulp_err(got, ref):
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref) / ulp


def test_expf_accuracy_and_edges(oracle):
    x = np.random.default_rng(0).uniform(-87, 88.7, 2_000_000).astype(np.float32)
    o = np.empty_like(x)
    oracle.oracle_expf_v(x.ctypes.data, x.size, o.ctypes.data)
    assert _ulp_err(o, np.exp(x.astype(np.float64))).max() < 1.1
    x = np.random.default_rng(1).uniform(-104, -87, 500_000).astype(np.float32)      # subnormal results
    o = np.empty_like(x)
    oracle.oracle_expf_v(x.ctypes.data, x.size, o.ctypes.data)
    assert (np.abs(o.astype(np.float64) - np.exp(x.astype(np.float64))) / 2.0 ** -149).max() < 1.0
    x = np.array([np.nan, np.inf, -np.inf, 0, -0.0, 88.72283, 88.72284, -103.9, -104.1, -1e30], np.float32)
    o = np.empty_like(x)
    oracle.oracle_expf_v(x.ctypes.data, x.size, o.ctypes.data)
    with np.errstate(over="ignore"):
        ref = np.exp(x)
    assert np.isnan(o[0]) and np.array_equal(o[1:], ref[1:])


def test_logf_accuracy_and_edges(oracle):
    x = (np.arange(1, 2 ** 24 + 1, dtype=np.float64) * 2.0 ** -24).astype(np.float32)   # every accept-uniform
    o = np.empty_like(x)
    oracle.oracle_logf_v(x.ctypes.data, x.size, o.ctypes.data)
    assert _ulp_err(o, np.log(x.astype(np.float64))).max() < 1.0
    x = np.exp(np.random.default_rng(1).uniform(-100, 88, 1_000_000)).astype(np.float32)
    o = np.empty_like(x)
    oracle.oracle_logf_v(x.ctypes.data, x.size, o.ctypes.data)
    assert _ulp_err(o, np.log(x.astype(np.float64))).max() < 1.0
    x = np.array([0, -0.0, np.inf, 1e-45, 1e-40, 1.0, 3e38], np.float32)
    o = np.empty_like(x)
    oracle.oracle_logf_v(x.ctypes.data, x.size, o.ctypes.data)
    with np.errstate(divide="ignore"):
        assert np.array_equal(o, np.log(x))
    x = np.array([np.nan, -1.0, -np.inf], np.float32)
    oracle.oracle_logf_v(x.ctypes.data, x.size, o.ctypes.data)
    assert np.isnan(o[:3]).all()


def test_sincos2pi_all_inputs(oracle):
    u = (np.arange(2 ** 24, dtype=np.float64) * 2.0 ** -24).astype(np.float32)          # every possible angle
    s = np.empty_like(u)
    c = np.empty_like(u)
    oracle.oracle_sincos2pi_v(u.ctypes.data, u.size, s.ctypes.data, c.ctypes.data)
    a = 2 * np.pi * u.astype(np.float64)
    assert np.abs(s - np.sin(a)).max() < 2e-7 and np.abs(c - np.cos(a)).max() < 2e-7


def test_uniform_conversions(oracle):
    a = np.array([0, 1, 255, 256, 0x7fffffff, 0x80000000, 0xffffff00, 0xffffffff], np.uint32)
    b = a[::-1].copy()
    u = np.empty(a.size, np.float32)
    up = np.empty(a.size, np.float32)
    u64 = np.empty(a.size, np.float64)
    oracle.oracle_uniforms_v(a.ctypes.data, b.ctypes.data, a.size, u.ctypes.data, up.ctypes.data, u64.ctypes.data)
    assert np.array_equal(u, (a >> 8).astype(np.float32) * np.float32(2.0 ** -24))
    assert u.min() == 0.0 and u.max() < 1.0
    assert up.min() > 0.0 and up.max() <= 1.0
    ref64 = ((a >> 5).astype(np.float64) * 67108864.0 + (b >> 6).astype(np.float64)) / 9007199254740992.0
    assert np.array_equal(u64, ref64) and u64.max() < 1.0


def test_normal_pair_statistics(oracle):
    rng = np.random.default_rng(3)
    n = 4_000_000
    a = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    z0 = np.empty(n, np.float32)
    z1 = np.empty(n, np.float32)
    oracle.oracle_normal_pair_v(a.ctypes.data, b.ctypes.data, n, z0.ctypes.data, z1.ctypes.data)
    z = np.concatenate([z0, z1]).astype(np.float64)
    assert np.isfinite(z).all()
    assert abs(z.mean()) < 2e-3 and abs(z.var() - 1) < 3e-3
    assert abs((z ** 3).mean()) < 1e-2 and abs((z ** 4).mean() - 3) < 3e-2
    assert abs(np.mean(z0.astype(np.float64) * z1)) < 2e-3
    assert np.abs(z).max() < 6.8


def test_expf_select_form_equals_early_return_form(oracle):
    """glabc_expf (clamp + selects, the chain step's form) and glabc_expf_b (early returns, the MFMA kernel's form)
    return the same bits everywhere: a dense sweep of the finite range, both range ends, specials."""
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.uniform(-110, 92, 2_000_000).astype(np.float32),
        np.linspace(-104.5, -103.5, 200001, dtype=np.float32), np.linspace(88.0, 89.5, 200001, dtype=np.float32),
        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 88.72283935546875, -104.0, 1e30, -1e30, 1e-45, -1e-45],
                 dtype=np.float32),
        rng.integers(0, 2 ** 32, 2_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    a, b = np.empty_like(x), np.empty_like(x)
    oracle.oracle_expf_v(x.ctypes.data, x.size, a.ctypes.data)
    oracle.oracle_expf_b_v(x.ctypes.data, x.size, b.ctypes.data)
    nan = np.isnan(a)
    assert np.array_equal(nan, np.isnan(b)) and np.array_equal(nan, np.isnan(x))
    assert np.array_equal(a[~nan].view(np.uint32), b[~nan].view(np.uint32))


def test_split_fixed_point_accumulator_equals_the_128_bit_one(oracle):
    """glabc_fxsplit (three 64-bit multiply-accumulates per term, the gradient loop's form) and glabc_fxsum (128-bit square)
    give the same integers: random streams up to the documented limits (|q| < 2^47, 65 536 terms), all-extreme streams."""
    rng = np.random.default_rng(1)
    streams = [rng.integers(-(1 << 47) + 1, 1 << 47, 65536), np.full(65536, (1 << 47) - 1), np.full(65536, -(1 << 47) + 1),
               rng.integers(-(1 << 20), 1 << 20, 1000), np.array([0, 1, -1, (1 << 24) - 1, -(1 << 24), 1 << 24, -(1 << 24) - 1]),
               (rng.standard_normal(400) * 0.22 * 2.0 ** 40).astype(np.int64)]
    for q in streams:
        q = np.ascontiguousarray(q, np.int64)
        out = np.zeros(6, np.uint64)
        oracle.oracle_fx_both(q.ctypes.data, q.size, out.ctypes.data)
        assert np.array_equal(out[:3], out[3:]), (out, q[:4])
        # and both equal the exact integer sums
        s2 = sum(int(v) * int(v) for v in q)
        assert int(out[1]) + (int(out[2]) << 64) == s2 and np.int64(out[0]) == q.sum()


def test_fx_quantize_is_rint_times_2_40(oracle):
    """glabc_fx_quantize's add-the-magic-constant form equals (int64) rint(d * 2^40) -- ties to even included -- on its
    domain |d| < 2^11 (the sampler stays below 2^7)."""
    rng = np.random.default_rng(2)
    d = np.concatenate([rng.standard_normal(1_000_000) * 0.3, rng.uniform(-2047, 2047, 1_000_000), rng.uniform(0, 1, 200_000),
                        (rng.integers(-(1 << 50), 1 << 50, 200_000) + 0.5) * 2.0 ** -40,        # exact ties
                        np.array([0.0, -0.0, 2.0 ** -41, -2.0 ** -41, 3 * 2.0 ** -41, 1.0, -1.0, 2047.999, -2047.999])])
    out = np.empty(d.size, np.int64)
    oracle.oracle_fx_quantize_v(d.ctypes.data, d.size, out.ctypes.data)
    assert np.array_equal(out, np.rint(d * 2.0 ** 40).astype(np.int64))
