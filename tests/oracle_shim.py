"""CPU stand-in for ``pysgmcmc_amd.kernels`` used ONLY by the ``-m "not gpu"`` tests
to exercise the host-side sampler logic (iterator protocol, burn-in switch,
schedule/batch feeding) on a machine without a GPU. Each function has the
signature of its ``kernels`` counterpart and runs the C oracle on the CPU
tensors' memory. The product never imports this (tests/test_boundary.py checks).
"""
import numpy as np
import torch

from oracle import sgmcmc_oracle as O


def _sfx(t):
    return "f32" if t.dtype == torch.float32 else "f64"


def sghmc_step(theta, V, grad, tau, g, v_hat, minv, r, eps, scale_grad, mdecay, adapt, xi=None, seed=0, step=0,
               step_dev=None, stats=None, grad_decay=0.0, launch=None, opts=None):
    lib = O.load_c()
    f = getattr(lib, "oracle_sghmc_step_" + _sfx(theta))
    p = lambda t: None if t is None else t.data_ptr()
    rc = f(p(theta), p(V), p(grad), p(tau), p(g), p(v_hat), p(minv), p(r), theta.numel(),
           float(eps), float(scale_grad), float(mdecay), float(grad_decay), int(bool(adapt)), p(xi), int(seed), int(step))
    assert rc == 0
    calls.append(("sghmc", bool(adapt), float(eps), int(step)))
    _fused_moments(theta, opts)


def sgld_step(theta, grad, tau, g, v_hat, minv, r, eps, A, scale_grad, adapt, xi=None, seed=0, step=0,
              step_dev=None, stats=None, grad_decay=0.0, launch=None, opts=None):
    lib = O.load_c()
    f = getattr(lib, "oracle_sgld_step_" + _sfx(theta))
    p = lambda t: None if t is None else t.data_ptr()
    rc = f(p(theta), p(grad), p(tau), p(g), p(v_hat), p(minv), p(r), theta.numel(),
           float(eps), float(A), float(scale_grad), float(grad_decay), int(bool(adapt)), p(xi), int(seed), int(step))
    assert rc == 0
    calls.append(("sgld", bool(adapt), float(eps), int(step)))
    _fused_moments(theta, opts)


def rsghmc_step(theta, p_, grad_cost, eps, mass, c, D, b_hat, xi=None, seed=0, step=0, step_dev=None, stats=None,
                grad_decay=0.0, launch=None, opts=None):
    lib = O.load_c()
    f = getattr(lib, "oracle_rsghmc_step_" + _sfx(theta))
    p = lambda t: None if t is None else t.data_ptr()
    rc = f(p(theta), p(p_), p(grad_cost), theta.numel(), float(eps), float(mass), float(c), float(D),
           float(b_hat), float(grad_decay), p(xi), int(seed), int(step))
    assert rc == 0
    calls.append(("rsghmc", False, float(eps), int(step)))
    _fused_moments(theta, opts)


def _fused_moments(theta, opts):
    """``opts`` (a dict of StepOpts keywords): the shim honours ``moments`` (K4 folded into the step) and refuses slices."""
    if not opts:
        return
    assert not opts.get("first_element"), "the CPU shim steps whole arenas only"
    if opts.get("moments") is not None:
        mean, m2, count = opts["moments"]
        moments_update(theta, mean, m2, count)


def svgd_workspace(n_particles, like):
    return torch.empty(1, dtype=like.dtype)


def _rows(t, n, dim, ld):
    """Writable [n, dim] numpy view of a flat tensor holding rows of pitch ld."""
    a = t.detach().numpy()
    ld = dim if ld is None else ld
    return np.lib.stride_tricks.as_strided(a, shape=(n, dim), strides=(ld * a.itemsize, a.itemsize))


def svgd_step(particles, grad, hist_grad, n_particles, dim, eps, alpha, fudge_factor, workspace, ld=None,
              repulsion_sign=1):
    X, G, H = (_rows(t, n_particles, dim, ld) for t in (particles, grad, hist_grad))
    Xc, Hc = np.ascontiguousarray(X), np.ascontiguousarray(H)
    O.svgd_step(Xc, np.ascontiguousarray(G), Hc, eps, alpha, fudge_factor, repulsion_sign)
    X[...] = Xc
    H[...] = Hc
    calls.append(("svgd", False, float(eps), int(repulsion_sign)))


def svgd_kernel(particles, n_particles, dim, workspace, ld=None, kernel_gradients=True):
    X = np.ascontiguousarray(_rows(particles, n_particles, dim, ld))
    K, kg, h, D = O.svgd_kernel(X)
    bw = torch.tensor([O.svgd_median(D), h, h * h], dtype=particles.dtype)
    return torch.from_numpy(K), (torch.from_numpy(kg) if kernel_gradients else None), bw


calls = []


def install(monkeypatch):
    """Route pysgmcmc_amd.kernels.*_step to the oracle for the duration of a test."""
    from pysgmcmc_amd import kernels
    del calls[:]
    monkeypatch.setattr(kernels, "sghmc_step", sghmc_step)
    monkeypatch.setattr(kernels, "sgld_step", sgld_step)
    monkeypatch.setattr(kernels, "rsghmc_step", rsghmc_step)
    monkeypatch.setattr(kernels, "svgd_workspace", svgd_workspace)
    monkeypatch.setattr(kernels, "svgd_step", svgd_step)
    monkeypatch.setattr(kernels, "svgd_kernel", svgd_kernel)
    return calls


# ---- diagnostics kernels (numpy stand-ins for K4 / R-hat pack+finish / K6) ----

def moments_update(theta, mean, m2, count, launch=None):
    O.c_moments_update(theta.detach().numpy(), mean.numpy(), m2.numpy(), int(count))


def rhat_pack(mean, m2, count, out3, n_shards=1, shard_len=None):
    out3.numpy()[:] = O.c_rhat_pack(mean.numpy(), m2.numpy(), count, n_shards, shard_len)


def rhat_finish(sum3, n, m_chains, count, rhat, summary_out4=None, summary_workspace=None, ld=None):
    rhat.numpy()[:n] = O.c_rhat_finish(np.ascontiguousarray(sum3.numpy()), m_chains, count, n=n, ld=n if ld is None else ld)
    if summary_out4 is not None:
        summary_out4.copy_(summary(rhat[:n]))


def summary_workspace(device):
    return torch.empty(1, dtype=torch.uint8)


def summary(x, out4=None, workspace=None):
    a = x.detach().numpy().astype(np.float64)
    return torch.tensor([a.sum(), (a * a).sum(), a.min(), a.max()], dtype=torch.float64)


def install_diagnostics(monkeypatch=None):
    from pysgmcmc_amd import kernels
    for name, fn in (("moments_update", moments_update), ("rhat_pack", rhat_pack),
                     ("rhat_finish", rhat_finish), ("summary", summary),
                     ("summary_workspace", summary_workspace)):
        if monkeypatch is not None:
            monkeypatch.setattr(kernels, name, fn)
        else:
            setattr(kernels, name, fn)
