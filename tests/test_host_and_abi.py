"""CPU-side checks: the C-ABI library loads and exports every symbol include/glabc.h
declares (no compute calls without a GPU), the ctypes structs match the C layout, the
host mirror keeps the reference's API surface and side effects, and the product refuses
to run without a GPU instead of falling back."""
import ctypes as C
import inspect
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from glabcmcmc_amd import _capi
    assert os.path.exists(_capi.LIB_PATH), "run `python __graft_entry__.py build` first"
    header = open(os.path.join(ROOT, "include", "glabc.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(glabc_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_capi.ENTRY_POINTS), declared ^ set(_capi.ENTRY_POINTS)
    h = C.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(h, name), name
    h.glabc_version.restype = C.c_int
    assert h.glabc_version() == 301 == _capi.VERSION and h.glabc_stream_layout() == _capi.STREAM_LAYOUT
    h.glabc_status_string.restype = C.c_char_p
    assert h.glabc_status_string(0) == b"ok"


def test_struct_layout_matches_c(tmp_path):
    from glabcmcmc_amd import _capi as A
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "glabc.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu '
                   '%zu %zu\\n", sizeof(glabc_dist), sizeof(glabc_model), sizeof(glabc_chains), sizeof(glabc_moments), '
                   'sizeof(glabc_tape), sizeof(glabc_run), offsetof(glabc_model, y_obs), offsetof(glabc_run, history));}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = list(map(int, subprocess.check_output([str(exe)]).split()))
    want = [C.sizeof(A.Dist), C.sizeof(A.Model), C.sizeof(A.Chains), C.sizeof(A.Moments), C.sizeof(A.Tape),
            C.sizeof(A.Run), A.Model.y_obs.offset, A.Run.history.offset]
    assert got == want


def test_package_surface_matches_reference():
    import glabcmcmc_amd as g
    for name in ("GlobalMCMC", "MCMCRunner", "Uniform", "Gamma", "DiagGaussian", "GaussianMixture", "GLMALA", "GLMCMC",
                 "AGLMCMC", "GLMCMC_NF", "esjd"):
        assert hasattr(g, name), name
    # positional signatures of the reference (GLMCMC.py:24-25, GlobalMCMC.py:6-7, MCMCRunner.py:17,35,78,100)
    pos = lambda f: [p.name for p in inspect.signature(f).parameters.values()            # noqa: E731
                     if p.kind == p.POSITIONAL_OR_KEYWORD]
    assert pos(g.GLMCMC) == ["ABCset", "num_ite", "Initial_theta", "Initial_y", "Local_Proposal", "filelocation",
                             "global_frequency", "Importance_Proposal", "batch_size"]
    assert pos(g.GlobalMCMC) == ["ABCset", "num_ite", "Initial_theta", "Initial_y", "Global_Proposal", "filelocation",
                                 "global_frequency", "Local_Proposal"]
    assert pos(g.GLMALA) == ["ABCset", "num_ite", "Initial_theta", "Initial_y", "tau", "num_grad", "filelocation",
                             "global_frequency", "Importance_Proposal", "batch_size"]
    r = g.MCMCRunner
    assert pos(r.run_glmcmc) == ["self", "num_iterations", "initial_theta", "initial_y", "global_frequency",
                                 "local_proposal", "importance_proposal", "batch_size", "output_file"]
    assert pos(r.run_global_mcmc) == ["self", "num_iterations", "initial_theta", "initial_y", "global_frequency",
                                      "local_proposal", "global_proposal", "output_file"]
    assert pos(r.run_glmala) == ["self", "num_iterations", "initial_theta", "initial_y", "global_frequency",
                                 "importance_proposal", "batch_size", "tau", "num_grad", "output_file"]


def test_no_cpu_fallback():
    """Without a GPU the samplers and esjd raise; they never compute on the host."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import glabcmcmc_amd as g
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    m = Mixture_set(0.05)
    dg = g.DiagGaussian(2, torch.zeros(2), torch.zeros(2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.GLMCMC(m, 10, torch.zeros(2), torch.zeros(1, 2), dg, None, 0.5, dg, 5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.GlobalMCMC(m, 10, torch.zeros(2), torch.zeros(1, 2), dg, None, 0.5, dg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.esjd(torch.zeros(5, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.GLMALA(m, 10, torch.zeros(2), torch.zeros(1, 2), 0.3, 10, None, 0.5, dg, 5)
    with pytest.raises((RuntimeError, g._capi.HipLibraryMissing)):
        g.GLMCMC_NF(m, 10, torch.zeros(2), torch.zeros(1, 2), dg, None, 0.5, 10, 5, None, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.KernelDensity().fit(torch.zeros(4, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.AGLMCMC(m, 10, torch.zeros(2), torch.zeros(1, 2), dg, dg, None, 0.5, 10, 5, 0.8, 0.2)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "gl-abc-mcmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in text and "libglabc_oracle" not in text and "oracle/" not in text.replace(
                    "in oracle/", ""), os.path.join(dirpath, f)


def test_model_without_descriptor_is_refused():
    from glabcmcmc_amd import engine

    class Plain:
        theta_dim = 2

    with pytest.raises(TypeError, match="descriptor"):
        engine.model_descriptor(Plain())


def test_host_distributions_match_reference_goldens():
    """CPU-tensor path of the host mirror (torch formulas) against the reference's values."""
    from helpers import bits, load_golden
    from glabcmcmc_amd import distribution
    p = load_golden("primitives")
    for tag in ("std", "gen", "d4", "d8"):
        g = distribution.DiagGaussian(len(p["dg_%s_loc" % tag]), torch.from_numpy(p["dg_%s_loc" % tag]),
                                      torch.from_numpy(p["dg_%s_log_scale" % tag]))
        assert np.array_equal(bits(g.log_prob(torch.from_numpy(p["dg_%s_z" % tag])).numpy()),
                              bits(p["dg_%s_log_prob" % tag]))
        d = g.descriptor()
        assert np.array_equal(bits(np.array(d.p2[:d.dim])), bits(p["dg_%s_scale" % tag]))
    u = distribution.Uniform(2, torch.tensor([-3.0, -3.0]), torch.tensor([3.0, 3.0]))
    assert np.array_equal(bits(u.log_prob(torch.from_numpy(p["un_box_z"])).numpy()), bits(p["un_box_log_prob"]))
    assert np.float32(distribution.Uniform(2).descriptor().c0) == p["un_default_log_prob_val"]
    ga = distribution.Gamma(torch.tensor([2.0, 3.0]), torch.tensor([1.0, 2.0]))
    assert abs(ga.log_prob(torch.tensor([[1.0, 1.0]], dtype=torch.float64)).item() - (-1.6137056407847639)) < 1e-12
    assert ga.log_prob(torch.tensor([[-1.0, 1.0]], dtype=torch.float64)).item() == -np.inf


def test_csv_side_effect_matches_reference_schedule(tmp_path):
    """_host.write_csv reproduces both of the reference's dump schedules, including the
    duplicated tail block of GlobalMCMC.py:70-76."""
    from glabcmcmc_amd import _host

    def reference_rows(num_ite, variant):
        rows = [0]
        for i in range(1, num_ite):
            if i % 10000 == 0 or i == num_ite - 1:
                k = (i - 1) // 10000 if variant == "glmcmc" else i // 10000
                lo = max(1, k * 10000 + 1) if variant == "glmcmc" else max(1, (k - 1) * 10000 + 1)
                rows += list(range(lo, i + 1))
        return rows

    for num_ite in (2, 50, 10001, 12005, 20001):
        chain = torch.arange(num_ite, dtype=torch.float32).view(-1, 1).repeat(1, 2)
        for variant in ("glmcmc", "global"):
            f = tmp_path / ("c_%d_%s.csv" % (num_ite, variant))
            _host.write_csv(chain, str(f), variant)
            got = [int(float(line.split(",")[0])) for line in open(f)]
            assert got == reference_rows(num_ite, variant), (num_ite, variant)
    assert len(reference_rows(12005, "global")) > 12005          # the reference really duplicates rows


def test_csv_equals_the_file_the_reference_wrote(tmp_path):
    """tests/golden/csv.npz: chains of 12 005 iterations and the SHA-256 of the CSV files the REFERENCE's GlobalMCMC
    (GlobalMCMC.py:70-76, duplicated tail block) and GLMCMC (GLMCMC.py:105-111) wrote for them; _host.write_csv must
    produce the same bytes."""
    import hashlib
    from helpers import load_golden
    from glabcmcmc_amd import _host
    g = load_golden("csv")
    for algo, variant in (("globalmcmc", "global"), ("glmcmc", "glmcmc")):
        f = tmp_path / (algo + ".csv")
        _host.write_csv(torch.from_numpy(g[algo + "_chain"]), str(f), variant)
        data = open(f, "rb").read()
        lines = data.decode().splitlines()
        assert len(lines) == int(g[algo + "_n_lines"]), algo
        assert "\n".join(lines[:3]) == str(g[algo + "_head"]) and "\n".join(lines[-3:]) == str(g[algo + "_tail"])
        assert hashlib.sha256(data).hexdigest() == str(g[algo + "_sha256"]), algo
    assert int(g["globalmcmc_n_lines"]) == 22005 and int(g["glmcmc_n_lines"]) == 12005      # B9: 10 000 rows written twice


def resample_cases():
    from helpers import load_golden
    p = load_golden("primitives")
    for i, P, N, kind, n_out in eval(str(p["resample_cases"])):
        yield i, P, N, kind, n_out, p["resample_%d_w" % i], p["resample_%d_u0" % i], p["resample_%d_idx" % i]


def test_resample_matches_the_reference_function():
    """GLMCMC_NFs.resample against the reference's resample (GLMCMC_NFs.py:29-40) on the golden cases, including
    cumulative sums that end below 1 (the reference then returns fewer than N indices) and zero weights"""
    from glabcmcmc_amd.GLMCMC_NFs import resample
    seen_short = 0
    for i, P, N, kind, n_out, w, u0, want in resample_cases():
        got = resample(torch.from_numpy(w), N, u0=float(u0)).numpy()
        assert got.shape == want.shape and np.array_equal(got, want), (i, kind)
        seen_short += int(len(want) < N)
    assert seen_short >= 2


@pytest.mark.gpu
def test_resample_on_the_gpu_matches_the_reference_function():
    from glabcmcmc_amd.GLMCMC_NFs import resample
    for i, P, N, kind, n_out, w, u0, want in resample_cases():
        got = resample(torch.from_numpy(w).cuda(), N, u0=float(u0)).cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got, want), (i, kind)


def test_runner_creates_output_dir(tmp_path):
    import glabcmcmc_amd as g
    d = tmp_path / "a" / "b"
    r = g.MCMCRunner(object(), output_dir=str(d))
    assert d.is_dir() and r._path("x.csv") == os.path.join(str(d), "x.csv") and r._path(None) is None


def test_integration_stub_matches_the_abi():
    """The ctypes structures printed in INTEGRATION.md (the binding a maintainer of the reference would add) have the
    sizes and field names of the tested declarations in _capi.py."""
    from glabcmcmc_amd import _capi
    src = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = src[src.index("import ctypes as C, numpy as np, torch"):src.index("lib = C.CDLL")]
    ns = {}
    exec(code, ns)
    for name in ("Dist", "Model", "Chains", "Run"):
        stub, real = ns[name], getattr(_capi, name)
        assert C.sizeof(stub) == C.sizeof(real), name
        assert [f[0] for f in stub._fields_] == [f[0] for f in real._fields_], name


def test_callback_noise_streams_differ_between_shards():
    """Every rank of a sharded run gets the same seed and its own chain0; the torch generator that draws a callback Model's
    simulator noise (pool rows, MALA gradient estimates) is seeded with both, so shards do not share noise -- and one shard
    (chain0 = 0) keeps the seed's own stream."""
    import inspect
    from glabcmcmc_amd import generic
    seeds = {generic._noise_seed(7, c0) for c0 in (0, 1, 64, 65536, 2 ** 40)}
    assert len(seeds) == 5 and generic._noise_seed(7, 0) == 7
    assert all(0 <= s < 2 ** 63 for s in seeds)
    src = inspect.getsource(generic)
    assert src.count("manual_seed(_noise_seed(") == 2 and "manual_seed(key &" not in src and "manual_seed(self.key &" not in src
