"""glabcmcmc_amd -- MI355X-native drop-in for the hot path of caofff/GL-ABC-MCMC.

Mirrors the export list of the reference package (glabcmcmc/__init__.py:1-13): the
sampler functions, ``MCMCRunner``, ``distribution.*`` and ``esjd``.  The samplers and
``esjd`` execute hand-written gfx950 kernels through the C ABI of ``include/glabc.h``
(``libglabc_hip.so``); importing the package needs neither a GPU nor the library.
"""
from .GlobalMCMC import GlobalMCMC
from .MCMCRunner import MCMCRunner
from .distribution import (
    Uniform,
    Gamma,
    DiagGaussian,
    GaussianMixture,
)
from .GLMALA import GLMALA
from .GLMCMC import GLMCMC
from .AGLMCMC import AGLMCMC
from .GLMCMC_NFs import GLMCMC_NF
from .ESJD import esjd
from .kernel_density import KernelDensity
from .compiled import CompiledModel
from . import _capi, checkpoint, compiled, distribution, engine, flows, generic, parallel, streaming

__all__ = ["GlobalMCMC", "MCMCRunner", "Uniform", "Gamma", "DiagGaussian", "GaussianMixture", "GLMALA", "GLMCMC",
           "AGLMCMC", "GLMCMC_NF", "esjd", "KernelDensity", "distribution", "engine", "CompiledModel"]
