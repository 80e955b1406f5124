"""The reference's BNN quick-start (tests/bayesian_neural_network/test_train_predict.py:20-48): fit sinc with a
3x50 tanh BNN sampled by SGHMC, predict mean and variance. Same class and keywords as
`pysgmcmc.models.bayesian_neural_network.BayesianNeuralNetwork`; runs on cuda:0."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import time

import numpy as np

from pysgmcmc_amd.models.bayesian_neural_network import BayesianNeuralNetwork
from pysgmcmc_amd.sampling import Sampler

rng = np.random.RandomState(1)
X = rng.rand(100, 1)
y = np.sinc(X * 10 - 5).sum(axis=1)
X_test = np.linspace(0, 1, 100)[:, None]
y_test = np.sinc(X_test * 10 - 5).sum(axis=1)

bnn = BayesianNeuralNetwork(sampling_method=Sampler.SGHMC, burn_in_steps=1000, sample_steps=100, n_nets=100,
                            batch_size=20, seed=1)
t0 = time.perf_counter()
bnn.train(X, y)
t1 = time.perf_counter()
mean, var = bnn.predict(X_test)
print("trained %d steps in %.2f s (fused small-model kernel: %s); test MSE %.5f, mean predictive variance %.5f"
      % (bnn.sampler.n_iterations, t1 - t0, bnn.used_fused_steps, float(np.mean((mean - y_test) ** 2)), float(var.mean())))

# the same with 10 chains advancing together: 100 networks after a tenth of the sampling iterations
bnn10 = BayesianNeuralNetwork(sampling_method=Sampler.SGHMC, burn_in_steps=1000, sample_steps=100, n_nets=100,
                              batch_size=20, seed=1, n_chains=10)
t0 = time.perf_counter()
bnn10.train(X, y)
t1 = time.perf_counter()
mean, var = bnn10.predict(X_test)
print("10 chains: %d steps per chain in %.2f s; test MSE %.5f, mean predictive variance %.5f"
      % (bnn10.sampler.n_iterations, t1 - t0, float(np.mean((mean - y_test) ** 2)), float(var.mean())))
