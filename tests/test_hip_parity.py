"""GPU parity: the HIP kernels (through the C ABI) against the CPU oracle.

Bar (DESIGN.md "Parity"):
  * injected noise (xi given): BIT-EXACT in f32 and f64 -- every array, every step;
  * Philox words: bit-exact integers;
  * Philox normals: f32 |gpu - oracle| <= 4e-6 * max(1, |z|) except where u is within
    2^-20 of 1 (|z| < 2e-3, hardware log2 is absolute- not relative-accurate there),
    where the bound is 1e-4; f64 <= 1e-12;
  * in-register noise step == injected step fed with the K5 stream (bit-exact).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [(np.float32, torch.float32), (np.float64, torch.float64)]
# ragged sizes: < 1 quad, not a multiple of 4, not a multiple of the block, > 1 grid pass
SIZES = [1, 3, 4, 5, 1023, 5252, 70001]


def _dev(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


class GpuState(object):
    def __init__(self, cst, gpu):
        for name in ("theta", "V", "tau", "g", "v_hat", "minv", "r", "p"):
            setattr(self, name, _dev(getattr(cst, name), gpu))


def _assert_same(dev_t, host_a, what):
    got = dev_t.cpu().numpy()
    if not np.array_equal(got.view(np.uint32 if got.dtype == np.float32 else np.uint64),
                          host_a.view(np.uint32 if host_a.dtype == np.float32 else np.uint64)):
        idx = np.flatnonzero(got != host_a)
        raise AssertionError("%s: %d/%d elements differ, first idx %d gpu=%r oracle=%r" % (
            what, idx.size, got.size, idx[0] if idx.size else -1,
            got[idx[0]] if idx.size else None, host_a[idx[0]] if idx.size else None))


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("npdt,thdt", DTYPES)
def test_sghmc_injected_bit_exact(gpu, oracle, n, npdt, thdt):
    from pysgmcmc_amd import kernels
    rng = np.random.default_rng(1234 + n)
    cst = oracle.CState(rng.normal(size=n), npdt)
    gst = GpuState(cst, gpu)
    burn = 7
    for t in range(15):
        grad = (rng.normal(size=n) * 3).astype(npdt)
        if t == 3:
            grad[:] = 0          # zero-gradient step
        xi = rng.normal(size=n).astype(npdt)
        adapt = t < burn
        oracle.c_sghmc_step(cst, grad, 0.01, 100.0, 0.05, adapt, xi)
        kernels.sghmc_step(gst.theta, gst.V, _dev(grad, gpu), gst.tau, gst.g, gst.v_hat, gst.minv, gst.r,
                           0.01, 100.0, 0.05, adapt, xi=_dev(xi, gpu))
        for name in ("theta", "V", "tau", "g", "v_hat", "minv", "r"):
            _assert_same(getattr(gst, name), getattr(cst, name), "sghmc step %d %s" % (t, name))


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("npdt,thdt", DTYPES)
def test_sgld_injected_bit_exact(gpu, oracle, n, npdt, thdt):
    from pysgmcmc_amd import kernels
    rng = np.random.default_rng(99 + n)
    cst = oracle.CState(rng.normal(size=n), npdt)
    gst = GpuState(cst, gpu)
    for t in range(15):
        grad = (rng.normal(size=n) * 3).astype(npdt)
        xi = rng.normal(size=n).astype(npdt)
        adapt = t < 7
        oracle.c_sgld_step(cst, grad, 0.01, 1.0, 100.0, adapt, xi)
        kernels.sgld_step(gst.theta, _dev(grad, gpu), gst.tau, gst.g, gst.v_hat, gst.minv, gst.r,
                          0.01, 1.0, 100.0, adapt, xi=_dev(xi, gpu))
        for name in ("theta", "tau", "g", "v_hat", "minv", "r"):
            _assert_same(getattr(gst, name), getattr(cst, name), "sgld step %d %s" % (t, name))


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("npdt,thdt", DTYPES)
def test_rsghmc_injected_bit_exact(gpu, oracle, n, npdt, thdt):
    from pysgmcmc_amd import kernels
    rng = np.random.default_rng(7 + n)
    cst = oracle.CState(rng.normal(size=n), npdt)
    cst.p[:] = rng.normal(size=n).astype(npdt)
    gst = GpuState(cst, gpu)
    for t in range(15):
        grad = (rng.normal(size=n) * 3).astype(npdt)
        xi = rng.normal(size=n).astype(npdt)
        oracle.c_rsghmc_step(cst, grad, 0.001, 1.0, 1.0, 1.0, 0.0, xi)
        kernels.rsghmc_step(gst.theta, gst.p, _dev(grad, gpu), 0.001, 1.0, 1.0, 1.0, 0.0, xi=_dev(xi, gpu))
        for name in ("theta", "p"):
            _assert_same(getattr(gst, name), getattr(cst, name), "rsghmc step %d %s" % (t, name))


def test_edge_cases_bit_exact(gpu, oracle):
    """v_hat -> 0 (safe_divide / safe_sqrt guards), negative v_hat, sigma clamp
    (eps_s^4 > 2 eps_s^2 mdecay minv), huge gradients."""
    from pysgmcmc_amd import kernels
    n = 64
    for npdt in (np.float32, np.float64):
        cst = oracle.CState(np.linspace(-1, 1, n), npdt)
        cst.v_hat[:16] = 0.0
        cst.v_hat[16:24] = -1e-16
        cst.v_hat[24:32] = 1e-30
        cst.g[8:20] = 0.0
        gst = GpuState(cst, gpu)
        rng = np.random.default_rng(5)
        for t, (eps, sg, md) in enumerate([(0.01, 1.0, 0.05), (2.0, 1.0, 1e-6), (0.1, 1e6, 0.05), (0.01, 1.0, 0.05)]):
            grad = rng.normal(size=n).astype(npdt)
            if t == 2:
                grad *= npdt(1e18)
            xi = rng.normal(size=n).astype(npdt)
            adapt = t < 3
            with np.errstate(all="ignore"):
                oracle.c_sghmc_step(cst, grad, eps, sg, md, adapt, xi)
            kernels.sghmc_step(gst.theta, gst.V, _dev(grad, gpu), gst.tau, gst.g, gst.v_hat, gst.minv, gst.r,
                               eps, sg, md, adapt, xi=_dev(xi, gpu))
            for name in ("theta", "V", "tau", "g", "v_hat", "minv"):
                got = getattr(gst, name).cpu().numpy()
                want = getattr(cst, name)
                assert np.array_equal(got, want, equal_nan=True), (npdt, t, name)
        assert np.isfinite(cst.minv).all()


def test_misaligned_and_empty(gpu, oracle):
    """Arrays that are not 16-B aligned take the element-wise path: same results. n = 0 is a no-op."""
    from pysgmcmc_amd import kernels
    n = 4099
    rng = np.random.default_rng(3)
    cst = oracle.CState(rng.normal(size=n), np.float32)
    big = {k: torch.zeros(n + 1, dtype=torch.float32, device=gpu) for k in
           ("theta", "V", "tau", "g", "v_hat", "minv", "grad", "xi")}
    views = {k: v[1:] for k, v in big.items()}      # 4-byte offset -> misaligned
    for k in ("theta", "V", "tau", "g", "v_hat", "minv"):
        views[k].copy_(_dev(getattr(cst, k), gpu))
    for t in range(4):
        grad = rng.normal(size=n).astype(np.float32)
        xi = rng.normal(size=n).astype(np.float32)
        views["grad"].copy_(_dev(grad, gpu))
        views["xi"].copy_(_dev(xi, gpu))
        adapt = t < 2
        oracle.c_sghmc_step(cst, grad, 0.01, 10.0, 0.05, adapt, xi)
        kernels.sghmc_step(views["theta"], views["V"], views["grad"], views["tau"], views["g"], views["v_hat"],
                           views["minv"], None, 0.01, 10.0, 0.05, adapt, xi=views["xi"])
        for name in ("theta", "V", "tau", "g", "v_hat", "minv"):
            _assert_same(views[name], getattr(cst, name), "misaligned step %d %s" % (t, name))
    e = torch.empty(0, dtype=torch.float32, device=gpu)
    kernels.sghmc_step(e, e, e, e, e, e, e, None, 0.01, 1.0, 0.05, True)
    torch.cuda.synchronize()


@pytest.mark.parametrize("seed,step", [(0, 0), (1, 0), (87654321, 12345), (2 ** 63 + 5, 2 ** 40 + 3)])
def test_philox_bits_exact(gpu, oracle, seed, step):
    from pysgmcmc_amd import kernels
    n = 100003
    out = torch.empty(n, dtype=torch.int32, device=gpu)
    kernels.philox_bits(out, seed, step)
    got = out.cpu().numpy().view(np.uint32)
    want = oracle.c_philox_bits(seed, step, n)
    assert np.array_equal(got, want)


def test_philox_normal_f32_tolerance(gpu, oracle):
    from pysgmcmc_amd import kernels
    n = 1 << 22
    out = torch.empty(n, dtype=torch.float32, device=gpu)
    kernels.philox_normal(out, 42, 7)
    got = out.cpu().numpy().astype(np.float64)
    want = oracle.c_philox_normal(42, 7, n, np.float32).astype(np.float64)
    err = np.abs(got - want)
    bits = oracle.c_philox_bits(42, 7, n).reshape(-1, 2)
    u_word = np.repeat(bits[:, 0], 2)               # the word that feeds log() for each pair
    near_one = u_word > np.uint32(0xFFFFF000)       # u within 2^-20 of 1
    scale = np.maximum(1.0, np.abs(want))
    assert (err[~near_one] <= 4e-6 * scale[~near_one]).all(), err[~near_one].max()
    assert (err[near_one] <= 1e-4).all()
    # the stream is a standard normal: moments over 4M draws
    assert abs(got.mean()) < 4 / np.sqrt(n)
    assert abs(got.var() - 1.0) < 6 * np.sqrt(2.0 / n)
    assert abs((got ** 4).mean() - 3.0) < 0.03


def test_philox_normal_f64_tolerance(gpu, oracle):
    from pysgmcmc_amd import kernels
    n = 100001
    out = torch.empty(n, dtype=torch.float64, device=gpu)
    kernels.philox_normal(out, 9, 1)
    want = oracle.c_philox_normal(9, 1, n, np.float64)
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-12


@pytest.mark.parametrize("npdt,thdt", DTYPES)
def test_register_noise_equals_injected_stream(gpu, oracle, npdt, thdt):
    """xi == NULL path: the in-register Philox draw equals feeding the K5 stream as xi (bit-exact),
    for the vector path, the ragged tail and every launch geometry."""
    from pysgmcmc_amd import kernels
    n = 70003
    rng = np.random.default_rng(11)
    th0 = rng.normal(size=n).astype(npdt)
    grad = _dev(rng.normal(size=n).astype(npdt), gpu)
    results = []
    try:
        for qpt, blocks, nt in [(1, 2048, 0), (2, 64, 0), (4, 2048, 1), (2, 7, 1)]:
            kernels.set_launch_config(256, qpt, blocks, nt)
            a = GpuState(oracle.CState(th0, npdt), gpu)
            b = GpuState(oracle.CState(th0, npdt), gpu)
            for t in range(3):
                xi = torch.empty(n, dtype=thdt, device=gpu)
                kernels.philox_normal(xi, 2024, t)
                kernels.sghmc_step(a.theta, a.V, grad, a.tau, a.g, a.v_hat, a.minv, None, 0.01, 50.0, 0.05, t < 2,
                                   xi=None, seed=2024, step=t)
                kernels.sghmc_step(b.theta, b.V, grad, b.tau, b.g, b.v_hat, b.minv, None, 0.01, 50.0, 0.05, t < 2,
                                   xi=xi)
                assert torch.equal(a.theta, b.theta) and torch.equal(a.V, b.V)
            results.append(a.theta.clone())
    finally:
        kernels.set_launch_config()           # back to the library defaults (a Python-side default, not library state)
    for r in results[1:]:
        assert torch.equal(results[0], r)     # geometry never changes the samples


def test_concurrent_host_threads_with_their_own_launch_geometry(gpu, oracle):
    """the ABI holds no mutable state, so one host thread per chain is safe. Four threads step four chains
    concurrently, each with its OWN per-call launch geometry (ABI v2: `const sgmcmc_launch_t *launch`) and its own HIP
    stream; every chain must equal the chain stepped alone with the default geometry, and a bad geometry in one
    thread raises there without disturbing the others."""
    import threading
    from pysgmcmc_amd import kernels
    from pysgmcmc_amd._lib import SgmcmcLibraryError
    n, steps = 200003, 25
    rng = np.random.default_rng(5)
    th0 = rng.normal(size=n).astype(np.float32)
    grad = _dev(rng.normal(size=n).astype(np.float32), gpu)

    def run(seed, launch, stream, out, k):
        try:
            st = GpuState(oracle.CState(th0, np.float32), gpu)
            with torch.cuda.stream(stream):
                for t in range(steps):
                    kernels.sghmc_step(st.theta, st.V, grad, st.tau, st.g, st.v_hat, st.minv, None, 0.01, 50.0, 0.05,
                                       t < 5, seed=seed, step=t, launch=launch)
                stream.synchronize()
            out[k] = st.theta
        except Exception as exc:                   # noqa: BLE001 - reported to the main thread
            out[k] = exc
    torch.cuda.synchronize()                       # th0 / grad uploads done before the side streams read them
    alone = {}
    for k in range(4):
        run(100 + k, None, torch.cuda.current_stream(), alone, k)
    geoms = [kernels.LaunchConfig(64, 1, 0, 0), kernels.LaunchConfig(128, 2, 40, 1), kernels.LaunchConfig(256, 4, 7, 0),
             kernels.LaunchConfig(192, 1, 0, 1)]
    together = {}
    threads = [threading.Thread(target=run, args=(100 + k, geoms[k], torch.cuda.Stream(device=gpu), together, k))
               for k in range(4)]
    bad = {}
    threads.append(threading.Thread(target=run, args=(1, kernels.LaunchConfig(100, 0, 0, -1), torch.cuda.Stream(device=gpu), bad, 0)))
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for k in range(4):
        assert isinstance(together[k], torch.Tensor), together[k]
        assert torch.equal(together[k], alone[k]), "chain %d differs when stepped next to other threads" % k
    assert isinstance(bad[0], SgmcmcLibraryError) and "block_threads" in str(bad[0])
    assert kernels.get_launch_config() == {"block_threads": -1, "quads_per_thread": 1, "max_blocks": 1 << 20, "nontemporal": 2}


def test_kernel_timestamp_events(gpu, oracle):
    """``LaunchConfig(events=KernelEvents())``: the launch goes through hipExtLaunchKernel and the events receive the
    kernel's own start/stop timestamps. Results are unchanged (bit-equal to a plain launch); the kernel duration is
    positive, below the duration of a hipEventRecord bracket around the same launch, and in the physically possible
    range for the bytes moved."""
    from pysgmcmc_amd import kernels
    n = 4_000_003
    rng = np.random.default_rng(3)
    th0 = rng.normal(size=n).astype(np.float32)
    grad = _dev(rng.normal(size=n).astype(np.float32), gpu)
    a, b = GpuState(oracle.CState(th0, np.float32), gpu), GpuState(oracle.CState(th0, np.float32), gpu)
    durs, brackets = [], []
    for t in range(12):
        kev = kernels.KernelEvents()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        kernels.sghmc_step(a.theta, a.V, grad, a.tau, a.g, a.v_hat, a.minv, None, 0.01, 50.0, 0.05, t < 4, seed=8, step=t,
                           launch=kernels.LaunchConfig(events=kev))
        e1.record()
        kernels.sghmc_step(b.theta, b.V, grad, b.tau, b.g, b.v_hat, b.minv, None, 0.01, 50.0, 0.05, t < 4, seed=8, step=t)
        torch.cuda.synchronize()
        durs.append(kev.elapsed_us())
        brackets.append(e0.elapsed_time(e1) * 1e3)
    for name in ("theta", "V", "tau", "g", "v_hat", "minv"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    frozen = np.array(durs[4:])
    # 24 B x 4 M = 96 MB: 12 us at 8 TB/s; anything from 10 us (cache-assisted) to 200 us is a sane kernel duration
    assert np.all(frozen > 10.0) and np.all(frozen < 200.0), frozen
    assert np.median(frozen) <= np.median(brackets[4:]) + 0.5, (frozen, brackets)


def test_moments_and_summary(gpu, oracle):
    from pysgmcmc_amd import kernels
    n = 50001
    rng = np.random.default_rng(0)
    mean_h = np.zeros(n, np.float32)
    m2_h = np.zeros(n, np.float32)
    mean_d, m2_d = _dev(mean_h, gpu), _dev(m2_h, gpu)
    for c in range(1, 6):
        x = rng.normal(size=n).astype(np.float32)
        oracle.c_moments_update(x, mean_h, m2_h, c)
        kernels.moments_update(_dev(x, gpu), mean_d, m2_d, c)
    _assert_same(mean_d, mean_h, "welford mean")
    _assert_same(m2_d, m2_h, "welford m2")
    x = rng.normal(size=n).astype(np.float32)
    s = kernels.summary(_dev(x, gpu)).cpu().numpy()
    x64 = x.astype(np.float64)
    assert np.isclose(s[0], x64.sum(), rtol=1e-12, atol=1e-9)
    assert np.isclose(s[1], (x64 * x64).sum(), rtol=1e-12)
    assert s[2] == x64.min() and s[3] == x64.max()
    s2 = kernels.summary(_dev(x, gpu)).cpu().numpy()
    assert np.array_equal(s, s2)              # deterministic


def test_rhat_pack_finish(gpu, oracle):
    from pysgmcmc_amd import kernels
    m, cnt, n = 4, 200, 1000
    rng = np.random.default_rng(1)
    chains = rng.normal(size=(m, cnt, n)).astype(np.float32) + rng.normal(size=(m, 1, n)).astype(np.float32) * 0.1
    total = torch.zeros(3 * n, dtype=torch.float32, device=gpu)
    for c in range(m):
        mean = torch.zeros(n, dtype=torch.float32, device=gpu)
        m2 = torch.zeros(n, dtype=torch.float32, device=gpu)
        for t in range(cnt):
            kernels.moments_update(_dev(chains[c, t], gpu), mean, m2, t + 1)
        out3 = torch.empty(3 * n, dtype=torch.float32, device=gpu)
        kernels.rhat_pack(mean, m2, cnt, out3)
        total += out3                           # stands in for the RCCL all-reduce(SUM)
    rhat = torch.empty(n, dtype=torch.float32, device=gpu)
    kernels.rhat_finish(total, n, m, cnt, rhat)
    want = oracle.gelman_rubin(chains)
    assert np.allclose(rhat.cpu().numpy(), want, rtol=2e-4, atol=1e-5)


def test_full_size_properties(gpu):
    """10 M parameters (BASELINE.json configs[2]): size-independent properties.
    (a) determinism: same (seed, step) -> identical arrays; (b) different step -> different noise;
    (c) linearity in the noise-free limit: with sigma clamped to 1e-8 and V0 = 0 one frozen step
        gives V' ~= -eps^2 * minv * grad; (d) sample moments of the injected noise."""
    from pysgmcmc_amd import kernels
    n = 10_002_434
    g = torch.Generator(device=gpu).manual_seed(0)
    theta0 = torch.randn(n, device=gpu, generator=g)
    grad = torch.randn(n, device=gpu, generator=g) * 0.1
    minv = torch.rand(n, device=gpu, generator=g) * 1.5 + 0.5

    def run(seed, step):
        th, V = theta0.clone(), torch.zeros_like(theta0)
        kernels.sghmc_step(th, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, seed=seed, step=step)
        return th, V
    th1, V1 = run(5, 9)
    th2, V2 = run(5, 9)
    assert torch.equal(th1, th2) and torch.equal(V1, V2)
    th3, V3 = run(5, 10)
    assert not torch.equal(V1, V3)
    assert torch.equal(th1, theta0 + V1)
    eps = 0.01
    eps_s = eps / np.sqrt(1e5)
    sigma = torch.sqrt(torch.clamp(2 * eps_s ** 2 * 0.05 * minv - eps_s ** 4, min=1e-16))
    z = (V1 + eps ** 2 * minv * grad) / sigma
    assert abs(z.mean().item()) < 5 / np.sqrt(n)
    assert abs(z.var().item() - 1.0) < 0.01
    assert z.abs().max().item() < 7.0


@pytest.mark.parametrize("n", [3, 1023, 70001, 3_000_001])
def test_fused_step_statistics(gpu, oracle, n):
    """stats_out = {sum theta'^2, sum V'^2, sum minv, sum minv^2} reduced inside the step kernel
    (wave shuffles -> LDS -> block partials -> fixed-order final): equal to a float64 reduction of the
    arrays the kernel wrote, deterministic, and the step itself is unchanged by asking for them."""
    from pysgmcmc_amd import kernels
    rng = np.random.default_rng(n)
    th0 = rng.normal(size=n).astype(np.float32)
    grad = _dev(rng.normal(size=n).astype(np.float32), gpu)
    for adapt in (True, False):
        a = GpuState(oracle.CState(th0, np.float32), gpu)
        b = GpuState(oracle.CState(th0, np.float32), gpu)
        a.minv.copy_(torch.rand(n, device=gpu) + 0.5); b.minv.copy_(a.minv)
        st = kernels.StepStats(n, gpu)
        kernels.sghmc_step(a.theta, a.V, grad, a.tau, a.g, a.v_hat, a.minv, None, 0.01, 50.0, 0.05, adapt,
                           seed=3, step=1, stats=st)
        kernels.step_stats_finish(st)
        kernels.sghmc_step(b.theta, b.V, grad, b.tau, b.g, b.v_hat, b.minv, None, 0.01, 50.0, 0.05, adapt,
                           seed=3, step=1)
        assert torch.equal(a.theta, b.theta) and torch.equal(a.V, b.V) and torch.equal(a.minv, b.minv)
        got = st.out.cpu().numpy()
        want = [(a.theta.double() ** 2).sum().item(), (a.V.double() ** 2).sum().item(),
                a.minv.double().sum().item(), (a.minv.double() ** 2).sum().item()]
        assert np.allclose(got, want, rtol=5e-7), (adapt, got, want)   # quad-level sums in f32, totals in f64
        first = got.copy()
        c = GpuState(oracle.CState(th0, np.float32), gpu)
        c.minv.copy_(b.minv if not adapt else torch.ones(n, device=gpu))
        if not adapt:
            kernels.sghmc_step(c.theta, c.V, grad, None, None, None, c.minv, None, 0.01, 50.0, 0.05, False,
                               seed=3, step=1, stats=st)
            kernels.step_stats_finish(st)
            assert np.array_equal(st.out.cpu().numpy(), first)      # bit-reproducible
    # SGLD and relativistic variants fill their slots
    a = GpuState(oracle.CState(th0, np.float32), gpu)
    st = kernels.StepStats(n, gpu)
    kernels.sgld_step(a.theta, grad, a.tau, a.g, a.v_hat, a.minv, None, 0.01, 1.0, 50.0, True, seed=1, step=0, stats=st)
    kernels.step_stats_finish(st)
    got = st.out.cpu().numpy()
    assert np.isclose(got[0], (a.theta.double() ** 2).sum().item(), rtol=5e-7) and got[1] == 0.0
    assert np.isclose(got[2], a.minv.double().sum().item(), rtol=5e-7)
    kernels.rsghmc_step(a.theta, a.p, grad, 0.001, 1.0, 1.0, 1.0, 0.0, seed=1, step=0, stats=st)
    kernels.step_stats_finish(st)
    got = st.out.cpu().numpy()
    assert np.isclose(got[0], (a.theta.double() ** 2).sum().item(), rtol=5e-7)
    assert np.isclose(got[1], (a.p.double() ** 2).sum().item(), rtol=5e-7)


@pytest.mark.parametrize("npdt,thdt", DTYPES)
def test_grad_decay_bit_exact(gpu, oracle, npdt, thdt):
    """grad_decay != 0: the kernel forms grad + grad_decay * theta in registers; bit-equal to the
    oracle for all three samplers, burn-in and frozen."""
    from pysgmcmc_amd import kernels
    n = 10007
    rng = np.random.default_rng(21)
    wd = 3.7e-4
    for sampler in ("sghmc", "sgld", "rsghmc"):
        cst = oracle.CState(rng.normal(size=n), npdt)
        cst.p[:] = rng.normal(size=n).astype(npdt)
        gst = GpuState(cst, gpu)
        for t in range(8):
            grad = (rng.normal(size=n) * 2).astype(npdt)
            xi = rng.normal(size=n).astype(npdt)
            adapt = t < 4
            if sampler == "sghmc":
                oracle.c_sghmc_step(cst, grad, 0.01, 100.0, 0.05, adapt, xi, grad_decay=wd)
                kernels.sghmc_step(gst.theta, gst.V, _dev(grad, gpu), gst.tau, gst.g, gst.v_hat, gst.minv, None,
                                   0.01, 100.0, 0.05, adapt, xi=_dev(xi, gpu), grad_decay=wd)
                names = ("theta", "V", "tau", "g", "v_hat", "minv")
            elif sampler == "sgld":
                oracle.c_sgld_step(cst, grad, 0.01, 1.0, 100.0, adapt, xi, grad_decay=wd)
                kernels.sgld_step(gst.theta, _dev(grad, gpu), gst.tau, gst.g, gst.v_hat, gst.minv, None,
                                  0.01, 1.0, 100.0, adapt, xi=_dev(xi, gpu), grad_decay=wd)
                names = ("theta", "tau", "g", "v_hat", "minv")
            else:
                oracle.c_rsghmc_step(cst, grad, 0.001, 1.0, 1.0, 1.0, 0.0, xi, grad_decay=wd)
                kernels.rsghmc_step(gst.theta, gst.p, _dev(grad, gpu), 0.001, 1.0, 1.0, 1.0, 0.0, xi=_dev(xi, gpu),
                                    grad_decay=wd)
                names = ("theta", "p")
            for name in names:
                _assert_same(getattr(gst, name), getattr(cst, name), "%s grad_decay step %d %s" % (sampler, t, name))


def test_large_array_64bit_indexing(gpu, oracle):
    """268 435 459 parameters (2^28 + 3: > 2^30 bytes per array, ragged tail, > 2^20-block grid-stride):
    slices at the start, around 2^27 and at the very end are bit-equal to the oracle (injected noise), and
    the in-register Philox stream at element indices ~2^28 equals the oracle's stream."""
    from pysgmcmc_amd import kernels
    n = (1 << 28) + 3
    g = torch.Generator(device=gpu).manual_seed(1)
    theta = torch.randn(n, device=gpu, generator=g)
    V = torch.randn(n, device=gpu, generator=g) * 0.01
    grad = torch.randn(n, device=gpu, generator=g)
    minv = torch.rand(n, device=gpu, generator=g) + 0.5
    xi = torch.randn(n, device=gpu, generator=g)
    windows = [(0, 1001), ((1 << 27) - 500, (1 << 27) + 501), (n - 1003, n)]
    before = [{k: t[a:b].cpu().numpy().copy() for k, t in (("theta", theta), ("V", V), ("grad", grad), ("minv", minv),
                                                          ("xi", xi))} for a, b in windows]
    kernels.sghmc_step(theta, V, grad, None, None, None, minv, None, 0.01, 1e5, 0.05, False, xi=xi)
    for (a, b), old in zip(windows, before):
        st = oracle.CState(old["theta"], np.float32)
        st.V[:] = old["V"]; st.minv[:] = old["minv"]
        oracle.c_sghmc_step(st, old["grad"], 0.01, 1e5, 0.05, False, old["xi"])
        assert np.array_equal(theta[a:b].cpu().numpy(), st.theta) and np.array_equal(V[a:b].cpu().numpy(), st.V)
    del xi, grad, V, minv
    kernels.philox_normal(theta, 77, 5)
    for a, b in windows:
        want = oracle.c_philox_normal_range(77, 5, a, b - a, np.float32).astype(np.float64)
        got = theta[a:b].cpu().numpy().astype(np.float64)
        assert np.abs(got - want).max() < 1e-4 and np.abs(got - want).mean() < 1e-6


def test_randomised_parity_sweep(gpu, oracle):
    """60 random configurations (sampler, dtype, size incl. ragged tails, stepsize, scale_grad, mdecay/A,
    grad_decay, phase, state magnitudes): injected-noise steps are bit-equal to the oracle."""
    from pysgmcmc_amd import kernels
    rng = np.random.default_rng(2024)
    for case in range(60):
        npdt = (np.float32, np.float64)[case % 2]
        sampler = ("sghmc", "sgld", "rsghmc")[case % 3]
        n = int(rng.integers(1, 40000))
        eps = float(10 ** rng.uniform(-4, -0.5))
        sg = float(10 ** rng.uniform(0, 6))
        md = float(rng.uniform(0.001, 0.5))
        wd = float(rng.choice([0.0, 10 ** rng.uniform(-6, -2)]))
        cst = oracle.CState(rng.normal(size=n) * 10 ** rng.uniform(-3, 2), npdt)
        cst.p[:] = rng.normal(size=n).astype(npdt)
        cst.V[:] = (rng.normal(size=n) * 0.01).astype(npdt)
        cst.v_hat[:] = (10 ** rng.uniform(-6, 4, size=n)).astype(npdt)
        cst.g[:] = rng.normal(size=n).astype(npdt)
        cst.tau[:] = (1 + rng.random(n) * 50).astype(npdt)
        cst.minv[:] = (10 ** rng.uniform(-2, 2, size=n)).astype(npdt)
        gst = GpuState(cst, gpu)
        for t in range(3):
            grad = (rng.normal(size=n) * 10 ** rng.uniform(-3, 3)).astype(npdt)
            xi = rng.normal(size=n).astype(npdt)
            adapt = bool(rng.integers(0, 2))
            with np.errstate(all="ignore"):
                if sampler == "sghmc":
                    oracle.c_sghmc_step(cst, grad, eps, sg, md, adapt, xi, grad_decay=wd)
                    kernels.sghmc_step(gst.theta, gst.V, _dev(grad, gpu), gst.tau, gst.g, gst.v_hat, gst.minv, gst.r,
                                       eps, sg, md, adapt, xi=_dev(xi, gpu), grad_decay=wd)
                    names = ("theta", "V", "tau", "g", "v_hat", "minv")
                elif sampler == "sgld":
                    oracle.c_sgld_step(cst, grad, eps, md * 4, sg, adapt, xi, grad_decay=wd)
                    kernels.sgld_step(gst.theta, _dev(grad, gpu), gst.tau, gst.g, gst.v_hat, gst.minv, gst.r,
                                      eps, md * 4, sg, adapt, xi=_dev(xi, gpu), grad_decay=wd)
                    names = ("theta", "tau", "g", "v_hat", "minv")
                else:
                    oracle.c_rsghmc_step(cst, grad, eps, 1.0 + md, 0.5 + md, 1.0, md * 0.1, xi, grad_decay=wd)
                    kernels.rsghmc_step(gst.theta, gst.p, _dev(grad, gpu), eps, 1.0 + md, 0.5 + md, 1.0, md * 0.1,
                                        xi=_dev(xi, gpu), grad_decay=wd)
                    names = ("theta", "p")
            for name in names:
                got = getattr(gst, name).cpu().numpy()
                assert np.array_equal(got, getattr(cst, name), equal_nan=True), (case, sampler, npdt, n, t, name)


def test_rhat_pack_refuses_a_buffer_of_another_dtype(gpu):
    """the pack kernel is chosen from the moments' dtype; an f64 pack into an f32 buffer of the same element
    count would write twice the allocation. Both the kernel front end and RhatExchange.start refuse it."""
    from pysgmcmc_amd import kernels
    from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments, RhatExchange
    n = 1000
    mean, m2 = torch.zeros(n, dtype=torch.float64, device=gpu), torch.ones(n, dtype=torch.float64, device=gpu)
    with pytest.raises(TypeError, match="share a dtype"):
        kernels.rhat_pack(mean, m2, 5, torch.empty(3 * n, dtype=torch.float32, device=gpu))
    with pytest.raises(TypeError, match="share a dtype"):
        kernels.rhat_pack(mean, m2.float(), 5, torch.empty(3 * n, dtype=torch.float64, device=gpu))
    kernels.rhat_pack(mean, m2, 5, torch.empty(3 * n, dtype=torch.float64, device=gpu))
    ex = RhatExchange(n, gpu)                                     # default dtype float32, mode all-reduce: no group needed to build
    mom = ChainMoments(n, gpu, dtype=torch.float64)
    mom.count = 5
    try:
        ex.start(mom)
        raise AssertionError("RhatExchange.start accepted float64 moments for a float32 exchange")
    except TypeError as exc:
        assert "float64" in str(exc)
    except RuntimeError as exc:                                   # no process group here: the dtype check must come first
        raise AssertionError("dtype must be checked before the process group: %s" % exc)
