"""Collect the OUTPUTS OF THE REFERENCE ITSELF that it holds in its own repository for the sampler path, as data:

  * docs/source/notebooks/data/effective_sample_sizes/Relativistic_SGHMC.json -- written by
    docs/source/experiments/compute_ess.py (TensorFlow run by the reference's author): effective sample size of
    RelativisticSGHMCSampler as a function of the stepsize on gmm2 / gmm3 / banana, 5 runs per stepsize, each
    20 consecutive segments x 10 000 kept samples (every 10th of 2e6 sampler steps), pymc3 effective_n, mean over
    the target's variables;
  * docs/source/notebooks/api_quickstart.ipynb -- the printed outputs of cell 13 (first `next(sampler)` of
    SGHMCSampler on the notebook's banana from (0, 0), float32, default arguments) and of cell 19 (pymc3 effective_n
    of 2 fresh RelativisticSGHMCSampler chains x 10 000 samples on the banana from (0, 6), stepsize 0.1).

Run in the build container (reads /root/reference; nothing of the reference travels, only these numbers):

    python3 tests/golden/make_reference_outputs.py
"""
import json
import os
import re

REF = "/root/reference/docs/source/notebooks"
HERE = os.path.dirname(os.path.abspath(__file__))

ess = json.load(open(os.path.join(REF, "data", "effective_sample_sizes", "Relativistic_SGHMC.json")))
curves = {}
for target, by_eps in ess.items():
    if not by_eps:
        continue                                    # gmm1: "no data yet!" (Effective_Sample_Sizes.ipynb cell 1)
    curves[target] = {"%.2f" % float(eps): [float(run[0]) for run in runs]
                      for eps, runs in sorted(by_eps.items(), key=lambda kv: float(kv[0]))}

nb = json.load(open(os.path.join(REF, "api_quickstart.ipynb")))


def cell_text(idx):
    out = []
    for o in nb["cells"][idx].get("outputs", []):
        t = o.get("text") or o.get("data", {}).get("text/plain")
        if t:
            out.append("".join(t))
    return "\n".join(out)


first = cell_text(13)                               # "([-0.0037382236, 0.0019394364], -50.0)"
nums = [float(x) for x in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", first)]
ess_q = cell_text(19)                               # "Effective Sample Sizes:\n{'x:0': 6.0, 'y:0': 3.0}"
ess_vals = dict((k, float(v)) for k, v in re.findall(r"'(\w+:0)': (\d+\.\d+)", ess_q))

doc = {
    "source": "MFreidank/pysgmcmc docs/source/notebooks (data/effective_sample_sizes/Relativistic_SGHMC.json; api_quickstart.ipynb "
              "cells 13 and 19); protocol docs/source/experiments/compute_ess.py:170-246",
    "ess_relativistic_sghmc": {"protocol": {"n_chains": 20, "samples_per_chain": 10000, "keep_every": 10, "dtype": "float32",
                                            "start": {"banana": [0.0, 6.0], "gmm": [0.0]},
                                            "sampler_defaults": {"mass": 1.0, "speed_of_light": 1.0, "D": 1.0, "Bhat": 0.0}},
                               "curves": curves},
    "quickstart_sghmc_first_next": {"sample": nums[:2], "cost": nums[2], "start": [0.0, 0.0], "dtype": "float32",
                                    "cost_fun": "-0.5 * (x**2 / 100 + (y + 0.1 * x**2 - 10)**2)"},
    "quickstart_relativistic_ess": {"values": ess_vals, "n_chains": 2, "samples_per_chain": 10000, "stepsize": 0.1,
                                    "start": [0.0, 6.0]},
}
json.dump(doc, open(os.path.join(HERE, "reference_outputs.json"), "w"), indent=1, sort_keys=True)
print("wrote reference_outputs.json:", {k: len(v) for k, v in curves.items()}, nums, ess_vals)
