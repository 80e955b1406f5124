"""GPU parity tests of the SVGD kernels (``sgmcmc_svgd_*``, csrc/sgmcmc_svgd.hip) against the numpy
oracle of ``pysgmcmc/samplers/svgd.py`` (oracle/sgmcmc_oracle.py: svgd_kernel / svgd_step).

Sums over columns and particles (tf.norm, tf.reduce_sum, tf.matmul) have no reference rounding order, so
the bar is a floating-point tolerance, written at each assert: f64 kernels against the f64 oracle at
1e-10, f32 kernels against the f64 oracle evaluated on the same f32 inputs at ~1e-5 relative.
"""
import numpy as np
import pytest
import torch

from oracle import sgmcmc_oracle as O
from pysgmcmc_amd import kernels
from pysgmcmc_amd._lib import SgmcmcLibraryError

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = {np.float32: dict(rtol=3e-5, atol=3e-6), np.float64: dict(rtol=1e-10, atol=1e-12)}


def _cloud(n, d, dtype, seed=0, offset=0.0, scale=1.0):
    rng = np.random.default_rng(seed)
    return (offset + scale * rng.normal(size=(n, d))).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,d", [(2, 1), (3, 2), (10, 2), (7, 3), (16, 5252), (50, 100), (64, 1000), (100, 3),
                                 (128, 257), (5, 70001), (33, 64), (1, 4)])
def test_svgd_kernel_matches_oracle(n, d, dtype):
    X = _cloud(n, d, dtype, seed=n * 1000 + d, scale=1.0 / np.sqrt(d))
    K_ref, kg_ref, h_ref, D_ref = O.svgd_kernel(X.astype(np.float64))
    x = torch.from_numpy(X).to(DEV)
    ws = kernels.svgd_workspace(n, x)
    K, kg, bw = kernels.svgd_kernel(x.reshape(-1), n, d, ws)
    torch.cuda.synchronize()
    K, kg, bw = K.cpu().numpy(), kg.cpu().numpy(), bw.cpu().numpy()
    if n == 1:
        # one particle: median 0, h 0, K = exp(-0/0) = nan in the reference too
        assert bw[0] == 0 and bw[1] == 0
        return
    tol = TOL[dtype]
    np.testing.assert_allclose(bw[0], O.svgd_median(D_ref), rtol=tol["rtol"])
    np.testing.assert_allclose(bw[1], h_ref, rtol=tol["rtol"])
    np.testing.assert_allclose(bw[2], h_ref * h_ref, rtol=2 * tol["rtol"])
    np.testing.assert_allclose(K, K_ref, **tol)
    assert np.array_equal(K, K.T) and np.all(np.diag(K) == 1)           # exact structure
    # kernel gradients: a difference of two O(|x| * rowsum) sums, tolerance relative to that scale
    scale = np.abs(X).max() * K_ref.sum(axis=1).max() / (h_ref * h_ref)
    np.testing.assert_allclose(kg, kg_ref, rtol=tol["rtol"], atol=tol["rtol"] * scale * 4)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("sign", [1, -1])
@pytest.mark.parametrize("n,d", [(10, 2), (20, 300), (64, 129), (128, 40), (4, 20000), (17, 1)])
def test_svgd_single_step_matches_oracle(n, d, sign, dtype):
    """One step from a random state (particles, gradients, running squared updates), both signs of the
    kernel-gradient term."""
    rng = np.random.default_rng(n * 31 + d)
    X = _cloud(n, d, dtype, seed=7 + n + d, offset=0.3, scale=1.0 / np.sqrt(d))
    G = rng.normal(size=(n, d)).astype(dtype)
    H = rng.uniform(0.05, 1.0, size=(n, d)).astype(dtype)
    x, g, h = (torch.from_numpy(a.copy()).to(DEV).reshape(-1) for a in (X, G, H))
    ws = kernels.svgd_workspace(n, x)
    eps, alpha, fudge = 0.05, 0.9, 1e-6
    kernels.svgd_step(x, g, h, n, d, eps, alpha, fudge, ws, repulsion_sign=sign)
    Xo, Ho = X.astype(np.float64), H.astype(np.float64)
    O.svgd_step(Xo, G.astype(np.float64), Ho, eps, alpha, fudge, float(sign))
    torch.cuda.synchronize()
    rtol = 2e-5 if dtype == np.float32 else 1e-10
    np.testing.assert_allclose(x.cpu().numpy().reshape(n, d), Xo, rtol=rtol, atol=rtol)
    np.testing.assert_allclose(h.cpu().numpy().reshape(n, d), Ho, rtol=10 * rtol, atol=rtol * 1e-2)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_svgd_free_running_trajectory(dtype):
    """12 steps of the repulsive update on a quadratic cost, starting from zero running statistics."""
    n, d = 12, 3
    X = _cloud(n, d, dtype, seed=21, offset=0.3)
    Xo, Ho = X.astype(np.float64), np.zeros((n, d))
    x = torch.from_numpy(X.copy()).to(DEV).reshape(-1)
    h = torch.zeros_like(x)
    ws = kernels.svgd_workspace(n, x)
    for t in range(12):
        g = (x * 1.5).contiguous()                                          # d/dx 0.75 |x|^2
        kernels.svgd_step(x, g, h, n, d, 0.05, 0.9, 1e-6, ws, repulsion_sign=-1)
        O.svgd_step(Xo, 1.5 * Xo, Ho, 0.05, 0.9, 1e-6, -1.0)
    rtol = 5e-4 if dtype == np.float32 else 1e-9
    np.testing.assert_allclose(x.cpu().numpy().reshape(n, d), Xo, rtol=rtol, atol=rtol)


def test_svgd_elementwise_tail_is_op_for_op():
    """With ONE far-apart pair the sums have a single term each, so the whole step must equal the f32 oracle
    up to the last bit of exp/log (checked at 4 ulp)."""
    X = np.array([[0.0, 1.0], [3.0, -2.0]], np.float32)
    G = np.array([[0.5, -1.0], [2.0, 0.25]], np.float32)
    H = np.array([[0.1, 0.2], [0.3, 0.4]], np.float32)
    x, g, h = (torch.from_numpy(a.copy()).to(DEV).reshape(-1) for a in (X, G, H))
    ws = kernels.svgd_workspace(2, x)
    kernels.svgd_step(x, g, h, 2, 2, 0.1, 0.9, 1e-6, ws, repulsion_sign=1)
    Xo, Ho = X.copy(), H.copy()
    O.svgd_step(Xo, G, Ho, 0.1, 0.9, 1e-6, 1.0)
    np.testing.assert_allclose(x.cpu().numpy().reshape(2, 2), Xo, rtol=5e-7)
    np.testing.assert_allclose(h.cpu().numpy().reshape(2, 2), Ho, rtol=5e-7)


def test_svgd_median_ties_zeros_and_reproducibility():
    # duplicated particles: many exact zeros in D, the median falls inside the tie
    base = _cloud(6, 40, np.float32, seed=5)
    X = np.concatenate([base, base, base[:3]], axis=0)                       # 15 particles, odd n*n
    x = torch.from_numpy(X).to(DEV)
    ws = kernels.svgd_workspace(15, x)
    K, kg, bw = kernels.svgd_kernel(x.reshape(-1), 15, 40, ws)
    _, _, h_ref, D_ref = O.svgd_kernel(X.astype(np.float64))
    np.testing.assert_allclose(bw[0].item(), O.svgd_median(D_ref), rtol=1e-5)
    # identical particles: K = 1 (to Gram-form rounding on the matrix-core path, n >= 9 in f32)
    np.testing.assert_allclose(K.cpu().numpy()[np.arange(6), np.arange(6) + 6], 1.0, atol=2e-6)
    Xs = X[:8]                                                               # n <= 8: difference form, exact zeros
    ws8 = kernels.svgd_workspace(8, x)
    K8, _, _ = kernels.svgd_kernel(torch.from_numpy(Xs).to(DEV).reshape(-1), 8, 40, ws8)
    assert np.all(K8.cpu().numpy()[np.arange(2), np.arange(2) + 6] == 1.0)
    # bit-reproducible run to run (fixed-order partial sums, no atomics) at a multi-workgroup size
    X = _cloud(24, 300000, np.float32, seed=9)
    outs = []
    for _ in range(2):
        x = torch.from_numpy(X.copy()).to(DEV).reshape(-1)
        g = (0.5 * x).contiguous()
        h = torch.zeros_like(x)
        ws = kernels.svgd_workspace(24, x)
        for _ in range(2):
            kernels.svgd_step(x, g, h, 24, 300000, 0.1, 0.9, 1e-6, ws, repulsion_sign=-1)
        outs.append((x.cpu().numpy(), h.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_svgd_row_pitch_and_large_dim_properties():
    """ld > dim leaves the padding untouched; at 16 particles x 2 M parameters the kernel matrix is invariant
    under a common translation of all particles and K of a scaled cloud equals K of the cloud (the median
    bandwidth scales with it)."""
    n, d, ld = 5, 37, 64
    buf = torch.full((n * ld,), 7.0, device=DEV)
    X = _cloud(n, d, np.float32, seed=2)
    buf.view(n, ld)[:, :d] = torch.from_numpy(X).to(DEV)
    g = torch.zeros_like(buf)
    h = torch.zeros_like(buf)
    ws = kernels.svgd_workspace(n, buf)
    kernels.svgd_step(buf, g, h, n, d, 0.1, 0.9, 1e-6, ws, ld=ld, repulsion_sign=-1)
    out = buf.view(n, ld).cpu().numpy()
    assert np.all(out[:, d:] == 7.0) and np.all(h.view(n, ld)[:, d:].cpu().numpy() == 0)
    Xo, Ho = X.astype(np.float64), np.zeros((n, d))
    O.svgd_step(Xo, np.zeros((n, d)), Ho, 0.1, 0.9, 1e-6, -1.0)
    np.testing.assert_allclose(out[:, :d], Xo, rtol=2e-5, atol=2e-6)

    n, d = 16, 2_000_003
    x = torch.randn(n, d, device=DEV) * (1.0 / d ** 0.5)
    ws = kernels.svgd_workspace(n, x)
    K0, _, bw0 = kernels.svgd_kernel(x.reshape(-1), n, d, ws, kernel_gradients=False)
    K1, _, bw1 = kernels.svgd_kernel((x * 4.0).reshape(-1), n, d, ws, kernel_gradients=False)
    K2, _, _ = kernels.svgd_kernel((x + 0.125).reshape(-1), n, d, ws, kernel_gradients=False)
    torch.testing.assert_close(K1, K0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bw1[1], 4.0 * bw0[1], rtol=1e-5, atol=0)
    torch.testing.assert_close(K2, K0, rtol=1e-3, atol=1e-4)
    ref = torch.cdist(x.double(), x.double()) ** 2
    h2 = 0.5 * ref.flatten().sort().values[[n * n // 2 - 1, n * n // 2]].mean() / np.log(n + 1.0)
    torch.testing.assert_close(K0.double(), torch.exp(-ref / h2 / 2), rtol=1e-4, atol=1e-5)


def test_svgd_fails_loudly():
    x = torch.zeros(129 * 4, device=DEV)
    with pytest.raises(ValueError):
        kernels.svgd_workspace(129, x)
    ws = kernels.svgd_workspace(128, x)
    with pytest.raises(SgmcmcLibraryError):
        kernels.svgd_step(x, x.clone(), x.clone(), 129, 4, 0.1, 0.9, 1e-6, ws)
    with pytest.raises(SgmcmcLibraryError):
        kernels.svgd_step(x, x.clone(), x.clone(), 4, 4, 0.1, 0.9, 1e-6, ws, repulsion_sign=0)
    with pytest.raises(SgmcmcLibraryError):
        kernels.svgd_workspace(4, torch.zeros(4))                          # CPU tensor: no CPU path


def test_svgd_sampler_on_gpu_fits_a_gaussian_and_reference_sign_collapses():
    import pysgmcmc_amd.samplers.svgd as svgd_mod
    from pysgmcmc_amd.sampling import Sampler
    rng = np.random.RandomState(0)
    x0 = rng.normal(size=(50, 2)) * 0.1 + 3.0
    mu = torch.tensor([1.0, -2.0], device=DEV)

    def cost(p):
        return 0.5 * ((p - mu) ** 2 / torch.tensor([1.0, 4.0], device=DEV)).sum()

    s = Sampler.get_sampler(Sampler.SVGD, particles=[torch.tensor(r, device=DEV) for r in x0], cost_fun=cost,
                            dtype=torch.float32)
    s.sample_format = "device"
    for _ in range(600):
        sample, costs = next(s)
    P = torch.stack(sample).cpu().numpy()
    assert costs.shape == (50,)
    np.testing.assert_allclose(P.mean(axis=0), [1.0, -2.0], atol=0.15)
    np.testing.assert_allclose(P.std(axis=0), [1.0, 2.0], rtol=0.3)
    K, kg = s.svgd_kernel()
    K_ref, kg_ref, _, _ = O.svgd_kernel(P.astype(np.float64))
    np.testing.assert_allclose(K.cpu().numpy(), K_ref, rtol=1e-4, atol=1e-5)

    s = Sampler.get_sampler(Sampler.SVGD, particles=[torch.tensor(r, device=DEV) for r in x0], cost_fun=cost,
                            dtype=torch.float32)
    s.strict_reference_quirks = True
    s.sample_format = "device"
    for _ in range(600):
        sample, _ = next(s)
    assert torch.stack(sample).std(dim=0).max().item() < 0.3             # quirk Q10: the cloud collapses


def test_svgd_golden_trajectories():
    """Committed oracle trajectories (tests/golden/svgd.npz: banana and 3-mode mixture, both signs, f32/f64):
    every step is replayed from the fixture's state with the fixture's gradients."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "svgd.npz"))
    for key in d["cases"]:
        key = str(key)
        target, n, dtname, sign = key.split("|")
        n, sign, dt = int(n), int(sign), np.dtype(dtname)
        X0 = d[key + "|x0"]
        dim = X0.shape[1]
        x = torch.from_numpy(X0.copy()).to(DEV).reshape(-1)
        h = torch.zeros_like(x)
        ws = kernels.svgd_workspace(n, x)
        K, kg, bw = kernels.svgd_kernel(x, n, dim, ws)
        tol = 3e-5 if dt == np.float32 else 1e-10
        np.testing.assert_allclose(K.cpu().numpy(), d[key + "|K0"], rtol=tol, atol=tol)
        np.testing.assert_allclose(bw.cpu().numpy(), d[key + "|bw0"], rtol=tol)
        for t in range(d[key + "|x"].shape[0]):
            g = torch.from_numpy(d[key + "|grad"][t]).to(DEV).reshape(-1)
            kernels.svgd_step(x, g, h, n, dim, 0.1, 0.9, 1e-6, ws, repulsion_sign=sign)
            want = d[key + "|x"][t]
            got = x.cpu().numpy().reshape(n, dim)
            # the first steps divide by sqrt(0.1 g^2) = a sign function of g: allow the step size on elements
            # whose update direction is numerically undetermined, tight everywhere else
            close = np.isclose(got, want, rtol=30 * tol, atol=30 * tol)
            assert close.mean() > 0.9 and np.abs(got - want).max() < 0.7, (key, t)
            x.copy_(torch.from_numpy(want).to(DEV).reshape(-1))             # re-anchor on the fixture
        torch.cuda.synchronize()


def test_svgd_sampler_vmap_and_hip_graph_modes_agree():
    """The particle-by-particle loop, the automatic vmap batching and the hipGraph replays are the same chain."""
    from pysgmcmc_amd.samplers import SVGDSampler
    x0 = np.random.RandomState(4).normal(size=(20, 3))
    scale = torch.tensor([1.0, 2.0, 0.5], device=DEV)
    cost = lambda p: 0.5 * ((p / scale) ** 2).sum()

    def chain(steps=25, loop=False, graph=False):
        s = SVGDSampler(particles=[torch.tensor(r, device=DEV) for r in x0], cost_fun=cost, dtype=torch.float32)
        s.sample_format = "device"
        if loop:
            s._vmapped = False
        s.use_hip_graph = graph
        for _ in range(steps):
            sample, costs = next(s)
        assert loop or s._vmapped                     # the batched form was accepted
        return torch.stack(sample).cpu().numpy(), costs.cpu().numpy()

    ref, cref = chain(loop=True)
    for kw in (dict(), dict(graph=True), dict(graph="full")):
        got, cgot = chain(**kw)
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(cgot, cref, rtol=2e-4, atol=2e-5)
    a, _ = chain(graph="full")
    b, _ = chain(graph="full")
    assert np.array_equal(a, b)                        # and reproducible bit for bit


def test_full_graph_mode_gives_up_on_a_moving_stepsize():
    """SVGD's kernel takes its stepsize by value, so ``use_hip_graph = "full"`` keeps one graph per stepsize: a schedule that
    keeps moving must not capture one per step -- after 2 * MAX_STEPSIZE_GRAPHS new values in a row the sampler steps with the
    cost graph + direct update, and the chain is the eager chain. The caller's ``use_hip_graph`` is left as configured;
    a CYCLIC schedule of more values than graphs are kept goes on replaying full graphs (least recently used one evicted)."""
    from pysgmcmc_amd.samplers import SVGDSampler
    from pysgmcmc_amd.stepsize_schedules import StepsizeSchedule

    class Decay(StepsizeSchedule):
        def __init__(self):
            self.t, self.initial_value = 0, 0.1

        def __next__(self):
            self.t += 1
            return 0.1 / self.t

        def update(self, *a, **k):
            pass
    x0 = np.random.RandomState(5).normal(size=(16, 3))
    cost = lambda p: 0.5 * (p ** 2).sum()

    def chain(graph):
        s = SVGDSampler(particles=[torch.tensor(r, device=DEV) for r in x0], cost_fun=cost, stepsize_schedule=Decay(), dtype=torch.float32)
        s.sample_format, s.use_hip_graph = "device", graph
        for _ in range(12):
            sample, _ = next(s)
        return torch.stack(sample).cpu().numpy(), s
    ref, _ = chain(False)
    got, s = chain("full")
    assert s.use_hip_graph == "full" and s._full_graph_disabled and sum(1 for k in s._graphs if k[:1] == ("full",)) == 0
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5)

    class Cycle(Decay):
        def __next__(self):
            self.t += 1
            return 0.1 / (1 + self.t % 2)                  # two values: both graphs stay

    def cyc(graph):
        s = SVGDSampler(particles=[torch.tensor(r, device=DEV) for r in x0], cost_fun=cost, stepsize_schedule=Cycle(), dtype=torch.float32)
        s.sample_format, s.use_hip_graph = "device", graph
        for _ in range(12):
            sample, _ = next(s)
        return torch.stack(sample).cpu().numpy(), s
    ref, _ = cyc(False)
    got, s = cyc("full")
    assert s.use_hip_graph == "full" and not s._full_graph_disabled and sum(1 for k in s._graphs if k[:1] == ("full",)) == 2
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("n", [12, 32, 64, 100])
def test_svgd_tight_cloud_far_from_origin(n):
    """Particles 0.01 apart around 100: the matrix-core path forms distances from a Gram matrix, which only works
    because every tile is centred per column first (|x|^2 = 1e4 d would swallow distances of 1e-4 d in f32)."""
    d = 3000
    rng = np.random.default_rng(n)
    X = (100.0 + 0.01 * rng.normal(size=(n, d))).astype(np.float32)
    K_ref, kg_ref, h_ref, _ = O.svgd_kernel(X.astype(np.float64))
    x = torch.from_numpy(X).to(DEV)
    ws = kernels.svgd_workspace(n, x)
    K, kg, bw = kernels.svgd_kernel(x.reshape(-1), n, d, ws)
    np.testing.assert_allclose(bw[1].item(), h_ref, rtol=2e-4)
    np.testing.assert_allclose(K.cpu().numpy(), K_ref, rtol=2e-3, atol=2e-4)
    G = rng.normal(size=(n, d)).astype(np.float32)
    H = np.full((n, d), 0.5, np.float32)
    g, h = torch.from_numpy(G).to(DEV).reshape(-1), torch.from_numpy(H.copy()).to(DEV).reshape(-1)
    xs = x.clone().reshape(-1)
    kernels.svgd_step(xs, g, h, n, d, 1e-3, 0.9, 1e-6, ws, repulsion_sign=-1)
    Xo, Ho = X.astype(np.float64), H.astype(np.float64)
    O.svgd_step(Xo, G.astype(np.float64), Ho, 1e-3, 0.9, 1e-6, -1.0)
    # the update itself is tiny next to 100: compare the displacement
    np.testing.assert_allclose(xs.cpu().numpy().reshape(n, d).astype(np.float64) - X, Xo - X, rtol=0.05, atol=2e-5)


def test_svgd_randomised_shape_sweep():
    """60 random (particles, parameters, row pitch, dtype, sign) combinations, one step each, against the oracle:
    exercises every dispatch range (register / matrix-core / two-pass kernels), aligned and unaligned pitches,
    partial tiles and base pointers that are only element-aligned."""
    rng = np.random.default_rng(20260101)
    for case in range(60):
        n = int(rng.choice([2, 3, 5, 8, 9, 12, 16, 17, 24, 32, 33, 48, 64, 65, 90, 128]))
        d = int(rng.choice([1, 2, 7, 31, 64, 65, 127, 128, 129, 300, 1000, 4099]))
        dtype = np.float32 if rng.random() < 0.5 else np.float64
        sign = int(rng.choice([1, -1]))
        pad = int(rng.choice([0, 0, 1, 3, 4, 64]))
        ld = d + pad
        offset = int(rng.choice([0, 0, 1, 2]))                          # shifts the base pointer by elements
        X = (0.3 + rng.normal(size=(n, d)) / np.sqrt(d)).astype(dtype)
        G = rng.normal(size=(n, d)).astype(dtype)
        H = rng.uniform(0.05, 1.0, size=(n, d)).astype(dtype)

        def dev(a):
            buf = torch.full((offset + n * ld,), 7.0, dtype=torch.from_numpy(a).dtype, device=DEV)
            view = buf[offset:]
            view.view(n, ld)[:, :d] = torch.from_numpy(a).to(DEV)
            return view
        x, g, h = dev(X), dev(G), dev(H)
        ws = kernels.svgd_workspace(n, x)
        kernels.svgd_step(x, g, h, n, d, 0.05, 0.9, 1e-6, ws, ld=ld, repulsion_sign=sign)
        Xo, Ho = X.astype(np.float64), H.astype(np.float64)
        O.svgd_step(Xo, G.astype(np.float64), Ho, 0.05, 0.9, 1e-6, float(sign))
        rtol = 3e-5 if dtype == np.float32 else 1e-10
        got = x.view(n, ld).cpu().numpy()
        np.testing.assert_allclose(got[:, :d], Xo, rtol=rtol, atol=rtol, err_msg=str((case, n, d, ld, offset, dtype, sign)))
        np.testing.assert_allclose(h.view(n, ld).cpu().numpy()[:, :d], Ho, rtol=10 * rtol, atol=rtol * 1e-2,
                                   err_msg=str((case, n, d, ld, offset, dtype, sign)))
        if pad:
            assert np.all(got[:, d:] == 7.0), (case, n, d, ld)           # the padding is never written


def test_svgd_sampler_with_attached_moments():
    """``attach_moments`` on a sampler whose kernel takes no step extras (SVGD) must not raise: the Welford
    pass runs as its own launch (K4) after the step and equals the moments of the visited states."""
    from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments
    from pysgmcmc_amd.sampling import Sampler
    rng = np.random.RandomState(1)
    x0 = rng.normal(size=(12, 3))
    s = Sampler.get_sampler(Sampler.SVGD, particles=[torch.tensor(r, device=DEV) for r in x0],
                            cost_fun=lambda p: 0.5 * (p ** 2).sum(), dtype=torch.float32)
    s.sample_format = "device"
    m = ChainMoments(s.arena.n, DEV)
    s.attach_moments(m, every=2)
    seen = []
    for i in range(8):
        next(s)
        if (i + 1) % 2 == 0:
            seen.append(s.arena.row("theta").double().clone())
    assert m.count == 4
    ref = torch.stack(seen)
    assert torch.allclose(m.mean.double(), ref.mean(dim=0), atol=1e-5)
    assert torch.allclose(m.m2.double(), ((ref - ref.mean(dim=0)) ** 2).sum(dim=0), atol=1e-4)
