"""The bench workloads (BASELINE.json configs) and the chain they step."""
import os

import torch

from benchlib.common import BATCH, N_DATA

# workloads: the default is BASELINE.json configs[2]; the 50 M ones are configs[4]'s two samplers (HBM-resident working sets,
# burn-in stepsize ramp) and sinc-bnn is configs[1] (the reference's own BNN test case) -- for profiles/, not the headline line.
WORKLOADS = {
    "bnn10m-sghmc": dict(sampler="sghmc", layers=(784, 2048, 2048, 2048)),        # 10 002 434 params
    "bnn50m-sgld": dict(sampler="sgld", layers=(512, 4864, 4864, 4864)),          # 49 826 818 params
    "bnn50m-rsghmc": dict(sampler="rsghmc", layers=(512, 4864, 4864, 4864)),
    # SVGD update path on 16 particles of the same 10 M-parameter model (synthetic gradients)
    "svgd16-10m": dict(sampler="svgd", layers=(784, 2048, 2048, 2048), particles=16),
    # configs[1]: SGHMC on the 3 x 50 tanh sinc BNN (tests/bayesian_neural_network/test_train_predict.py:20-48), 5 252 params
    "sinc-bnn": dict(sampler="sinc", layers=(1, 50, 50, 50)),
}
LAYERS = WORKLOADS["bnn10m-sghmc"]["layers"]


def build_chain(dev, rank, workload="bnn10m-sghmc", burn_in=8, dtype=torch.float32):
    from pysgmcmc_amd.data_batches import Placeholder, generate_batches
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
    from pysgmcmc_amd.samplers import RelativisticSGHMCSampler, SGHMCSampler, SGLDSampler
    from pysgmcmc_amd.stepsize_schedules import BurnInRampStepsizeSchedule, ConstantStepsizeSchedule

    spec = WORKLOADS[workload]
    layers = spec["layers"]
    g = torch.Generator(device=dev).manual_seed(0)             # same synthetic dataset on every rank
    X = torch.randn(N_DATA, layers[0], device=dev, generator=g).to(dtype)       # (the same numbers in either dtype)
    y = torch.randn(N_DATA, device=dev, generator=g).to(dtype)
    xp = Placeholder(dtype=dtype, device=dev, name="X_Minibatch")
    yp = Placeholder(dtype=dtype, device=dev, name="Y_Minibatch")
    params = init_mlp_params(layers[0], hidden=layers[1:], seed=1000 + rank, dtype=dtype, device=dev)
    cost = BNNCost(xp, yp, batch_size=BATCH, n_examples=N_DATA)
    # profiling aid (tools/gpu/r05_prof50m.sh): hidden layers on library products / on the fused launches whatever their tile count
    if os.environ.get("BENCH_FUSED_LAYERS") in ("library", "all"):
        cost.fused_layers = False if os.environ["BENCH_FUSED_LAYERS"] == "library" else "all"
    # profiling aid (tools/gpu/r06_gw_planes.sh): the batched weight gradients on the library / on the bf16 planes whatever their size
    if os.environ.get("BENCH_GW_PLANES") in ("off", "on"):
        cost.gw_on_bf16_planes = os.environ["BENCH_GW_PLANES"] == "on"
    common = dict(params=params, cost_fun=cost,
                  batch_generator=generate_batches(X, y, xp, yp, batch_size=BATCH, seed=rank),
                  session=dev, dtype=dtype, seed=1234 + rank)
    if spec["sampler"] == "sghmc":
        return SGHMCSampler(stepsize_schedule=ConstantStepsizeSchedule(0.01), mdecay=0.05,
                            scale_grad=float(N_DATA),
                            burn_in_steps=burn_in,             # adapted during warmup; timed steps are frozen
                            **common)
    if spec["sampler"] == "sgld":
        # configs[4]: preconditioned SGLD with a burn-in stepsize ramp (a StepsizeSchedule subclass)
        return SGLDSampler(stepsize_schedule=BurnInRampStepsizeSchedule(1e-4, 1e-3, burn_in_steps=burn_in),
                           A=1.0, scale_grad=float(N_DATA), burn_in_steps=burn_in, **common)
    return RelativisticSGHMCSampler(stepsize_schedule=BurnInRampStepsizeSchedule(1e-4, 1e-3, burn_in_steps=burn_in),
                                    mass=1.0, speed_of_light=1.0, D=1.0, Bhat=0.0, **common)


