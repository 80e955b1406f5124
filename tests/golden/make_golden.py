"""Generate the golden fixtures under tests/golden/ (run in the BUILD container).

    python tests/golden/make_golden.py

Sources of truth
  * bnn_priors.npz ........ DATA copied from the reference's own test fixtures
        /root/reference/pysgmcmc/tests/data/bayesian_neural_network_priors/
        {log_variance,weights,weights_inputs}.npy  (values only; they are the
        reference's golden constants, tests/bayesian_neural_network/test_priors.py:20-81).
  * trajectories.npz ...... produced by the op-by-op numpy restatement of the
        reference's TF graph (oracle/sgmcmc_oracle.py). The reference cannot run
        here (TensorFlow absent) and ships no golden trajectories, so these pin the
        HIP kernels to the oracle, not to reference outputs ("parity unpinned",
        see DESIGN.md).
  * bnn_trajectory.npz .... 12 SGHMC steps (6 burn-in) of the 5 252-parameter 3x50 tanh sinc BNN:
        minibatch windows from RandomState(seed).randint like
        pysgmcmc/data_batches.py:118, gradients by torch-CPU fp64 autograd of the
        NLL restated from pysgmcmc/models/bayesian_neural_network.py:365-388,
        update by the numpy oracle.
  * svgd.npz .............. SVGD particle trajectories (pysgmcmc/samplers/svgd.py:118-181) from the numpy
        restatement oracle/sgmcmc_oracle.py:svgd_step on the reference's toy targets (banana, 3-mode
        mixture), both signs of the kernel-gradient term (+1 = the reference as written), f32 and f64;
        plus kernel matrix, kernel gradients and bandwidth of the initial particles. Oracle-pinned only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import sgmcmc_oracle as O  # noqa: E402

REF_PRIORS = "/root/reference/pysgmcmc/tests/data/bayesian_neural_network_priors"


def make_priors():
    lv = np.load(os.path.join(REF_PRIORS, "log_variance.npy"))
    w = np.load(os.path.join(REF_PRIORS, "weights.npy"))
    wi = np.load(os.path.join(REF_PRIORS, "weights_inputs.npy"), allow_pickle=True, encoding="latin1")
    out = {"log_variance_expected": np.float64(lv), "weights_expected": np.float64(w),
           "log_variance_input": np.full((20, 1), -11.25474104)}
    for k, t in enumerate(wi):
        out["weights_input_%d" % k] = np.asarray(t, np.float64)
    np.savez_compressed(os.path.join(HERE, "bnn_priors.npz"), **out)
    print("bnn_priors.npz: %d tensors, %d params" % (len(wi), sum(np.asarray(t).size for t in wi)))


TARGETS = {
    "gmm1": (np.array([0.0]), O.gmm_cost_grad),              # tests/samplers/sampler_testing.py:15
    "banana": (np.array([0.0, 6.0]), O.banana_cost_grad),    # :16-17
}
N_STEPS = 100


def run_case(sampler, target, dtype, eps, burn):
    theta0, cost_grad = TARGETS[target]
    rng = np.random.default_rng(1234)
    st = O.OpByOpState(theta0, dtype)
    if sampler == "rsghmc":
        st.p[:] = rng.normal(size=st.p.shape).astype(dtype)
    p0 = st.p.copy()
    n = theta0.size
    thetas, costs, grads, xis = [], [], [], []
    frozen = None
    for t in range(N_STEPS):
        cost, g = cost_grad(st.theta.ravel().astype(np.float64))
        g = g.astype(dtype)
        xi = rng.normal(size=n).astype(dtype)
        if sampler == "sghmc":
            adapting = t < burn or burn <= 0
            if adapting:
                O.opbyop_sghmc_step(st, g, eps, 1.0, 0.05, xi)
                frozen = st.minv.copy()
            else:
                O.opbyop_sghmc_step(st, g, eps, 1.0, 0.05, xi, frozen_minv=frozen)
        elif sampler == "sgld":
            adapting = t < burn or burn <= 0
            if adapting:
                O.opbyop_sgld_step(st, g, eps, 1.0, 1.0, xi)
                frozen = st.minv.copy()
            else:
                O.opbyop_sgld_step(st, g, eps, 1.0, 1.0, xi, frozen_minv=frozen)
        else:
            O.opbyop_rsghmc_step(st, g, eps, 1.0, 1.0, 1.0, 0.0, xi)
        thetas.append(st.theta.ravel().copy())
        costs.append(cost)
        grads.append(g)
        xis.append(xi)
    return {"theta": np.array(thetas), "cost": np.array(costs), "grad": np.array(grads), "xi": np.array(xis),
            "p0": p0.ravel(), "V_final": st.V.ravel(), "p_final": st.p.ravel(),
            "minv_frozen": (frozen if frozen is not None else st.minv).ravel()}


def make_trajectories():
    out = {}
    cases = []
    for sampler in ("sghmc", "sgld", "rsghmc"):
        for target in TARGETS:
            for dtype in (np.float32, np.float64):
                for eps in (0.01, 0.1):
                    for burn in ((0, 5, 50) if sampler != "rsghmc" else (0,)):
                        key = "%s|%s|%s|%g|%d" % (sampler, target, np.dtype(dtype).name, eps, burn)
                        with np.errstate(all="ignore"):
                            res = run_case(sampler, target, dtype, eps, burn)
                        for name, arr in res.items():
                            out[key + "|" + name] = arr
                        cases.append(key)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "trajectories.npz"), **out)
    print("trajectories.npz: %d cases" % len(cases))


def make_bnn_trajectory():
    import torch
    seed, n_data, batch_size, n_steps = 1, 100, 20, 12
    rng = np.random.RandomState(seed)
    X = rng.rand(n_data, 1)                                   # tests/.../test_train_predict.py:20-25
    y = np.sinc(X * 10 - 5).sum(axis=1)
    Xn = (X - X.mean(axis=0)) / X.std(axis=0)                  # models/base_model.py:125-133
    yn = (y - y.mean(axis=0)) / y.std(axis=0)
    shapes = [(1, 50), (50,), (50, 50), (50,), (50, 50), (50,), (50, 1), (1,), (1, 1)]
    init = np.random.default_rng(7)
    params0 = []
    for shp in shapes:
        if len(shp) == 2 and shp != (1, 1):
            params0.append(init.normal(size=shp) * np.sqrt(1.0 / shp[0]))
        elif shp == (1, 1):
            params0.append(np.full(shp, np.log(1e-3)))
        else:
            params0.append(np.zeros(shp))
    theta0 = np.concatenate([p.ravel() for p in params0])
    sizes = [int(np.prod(s)) for s in shapes]
    offs = np.concatenate([[0], np.cumsum(sizes)])

    def unflat(v):
        return [v[offs[k]:offs[k + 1]].reshape(shapes[k]) for k in range(len(shapes))]

    def nll_torch(theta, Xb, Yb):
        W1, b1, W2, b2, W3, b3, W4, b4, ob = unflat(theta)
        h = torch.tanh(Xb @ W1 + b1)
        h = torch.tanh(h @ W2 + b2)
        h = torch.tanh(h @ W3 + b3)
        mean = h @ W4 + b4
        log_var = torch.ones_like(mean) * ob
        f_var_inv = 1.0 / (torch.exp(log_var) + 1e-16)
        mse = (Yb - mean) ** 2
        ll = torch.sum(torch.sum(-mse * (0.5 * f_var_inv) - 0.5 * log_var, dim=1)) / batch_size
        d = 2.0 * 0.01
        lvp = torch.mean(torch.sum(-(log_var - np.log(1e-6)) ** 2 / (d + (2 * np.sign(d) * 1e-16 + 1e-16))
                                   - 0.5 * np.log(0.01), dim=1))
        ll = ll + lvp / n_data
        npar = float(theta.numel())
        wp = torch.sum(-0.5 * theta ** 2) / (npar + 3e-16)
        ll = ll + wp / n_data
        return -ll

    brng = np.random.RandomState(seed)                         # data_batches.py:104-105
    out = {"X": Xn, "y": yn, "theta0": theta0, "shapes": np.array([str(s) for s in shapes])}
    for dtype in (np.float32, np.float64):
        st = O.OpByOpState(theta0, dtype)
        brng.seed(seed)
        nrng = np.random.default_rng(4321)     # xi is regenerated from this seed by the tests
        thetas, costs, starts, xis, grads = [], [], [], [], []
        frozen = None
        burn = 6
        for t in range(n_steps):
            start = brng.randint(0, n_data - batch_size + 1)
            Xb = torch.tensor(Xn[start:start + batch_size])
            Yb = torch.tensor(yn[start:start + batch_size].reshape(-1, 1))
            th = torch.tensor(st.theta.ravel().astype(np.float64), requires_grad=True)
            cost = nll_torch(th, Xb, Yb)
            g, = torch.autograd.grad(cost, th)
            # cross-check the torch restatement against the numpy one
            c_np, _ = O.bnn_negative_log_likelihood(unflat(st.theta.ravel().astype(np.float64)),
                                                    Xb.numpy(), Yb.numpy(), batch_size, n_data)
            assert abs(float(cost.detach()) - float(c_np)) < 1e-10 * max(1.0, abs(float(c_np))), (float(cost.detach()), float(c_np))
            g = g.numpy().astype(dtype)
            xi = nrng.normal(size=theta0.size).astype(dtype)
            if t < burn:
                O.opbyop_sghmc_step(st, g, 0.01, float(n_data), 0.05, xi)
                frozen = st.minv.copy()
            else:
                O.opbyop_sghmc_step(st, g, 0.01, float(n_data), 0.05, xi, frozen_minv=frozen)
            thetas.append(st.theta.ravel().copy())
            costs.append(float(cost.detach()))
            starts.append(start)
            xis.append(xi)
            grads.append(g)
        name = np.dtype(dtype).name
        out[name + "|theta"] = np.array(thetas)
        out[name + "|cost"] = np.array(costs)
        out[name + "|grad"] = np.array(grads)
        out["starts"] = np.array(starts)
    np.savez_compressed(os.path.join(HERE, "bnn_trajectory.npz"), **out)
    print("bnn_trajectory.npz: %d params x %d steps" % (theta0.size, n_steps))


def make_svgd():
    out = {}
    cases = []
    rng = np.random.default_rng(2024)
    for target, n, d, steps in (("banana", 10, 2, 30), ("gmm1", 20, 1, 30)):
        x0 = rng.normal(size=(n, d)) * 0.5 + (np.array([0.0, 6.0]) if target == "banana" else 0.0)
        cost_grad = O.banana_cost_grad if target == "banana" else O.gmm_cost_grad
        for dt in (np.float32, np.float64):
            for sign in (1, -1):
                key = "%s|%d|%s|%d" % (target, n, np.dtype(dt).name, sign)
                cases.append(key)
                X = x0.astype(dt)
                H = np.zeros_like(X)
                K, kg, h, D = O.svgd_kernel(X)
                out[key + "|x0"] = X.copy()
                out[key + "|K0"], out[key + "|kgrad0"] = K, kg
                out[key + "|bw0"] = np.array([O.svgd_median(D), h, h * h], dt)
                traj, grads = [], []
                for t in range(steps):
                    G = np.stack([np.asarray(cost_grad(x.astype(np.float64))[1], np.float64).reshape(d) for x in X]).astype(dt)
                    grads.append(G)
                    O.svgd_step(X, G, H, 0.1, 0.9, 1e-6, float(sign))
                    traj.append(X.copy())
                out[key + "|grad"] = np.stack(grads)
                out[key + "|x"] = np.stack(traj)
                out[key + "|hist"] = H.copy()
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "svgd.npz"), **out)
    print("svgd.npz: %d cases" % len(cases))


if __name__ == "__main__":
    if os.path.isdir(REF_PRIORS):
        make_priors()
    else:
        print("reference not mounted: keeping committed bnn_priors.npz")
    make_trajectories()
    make_bnn_trajectory()
    make_svgd()
