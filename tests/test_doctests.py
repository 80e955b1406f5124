"""Run the doctests of the host-side modules (the reference's CI runs `--doctest-modules`,
pysgmcmc/pytest.ini:1-3); they restate the reference's doctest known answers."""
import doctest
import importlib

import pytest

MODULES = [
    "pysgmcmc_amd.stepsize_schedules", "pysgmcmc_amd.tensor_utils", "pysgmcmc_amd.sampling",
    "pysgmcmc_amd.data_batches", "pysgmcmc_amd.diagnostics.objective_functions",
    "pysgmcmc_amd.diagnostics.sample_chains", "pysgmcmc_amd.samplers.relativistic_sghmc",
]


@pytest.mark.parametrize("name", MODULES)
def test_module_doctests(name):
    mod = importlib.import_module(name)
    result = doctest.testmod(mod, optionflags=doctest.NORMALIZE_WHITESPACE | doctest.ELLIPSIS)
    assert result.attempted > 0 and result.failed == 0, result
