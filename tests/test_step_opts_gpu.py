"""ABI v3 step extras (``sgmcmc_step_opts_t``): slice launches, statistics selection, fused Welford moments, the
minv-store skip and device-resident scalars -- each must leave the update arithmetic of every element untouched."""
from itertools import islice

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def _state(n, dt, dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    mk = lambda sc: torch.randn(n, dtype=dt, device=dev, generator=g) * sc
    st = dict(theta=mk(0.5), V=mk(0.1), grad=mk(0.3), tau=torch.rand(n, dtype=dt, device=dev, generator=g) + 1.0,
              g=mk(0.2), v_hat=torch.rand(n, dtype=dt, device=dev, generator=g) + 0.1,
              minv=torch.rand(n, dtype=dt, device=dev, generator=g) + 0.5)
    return st


def _call(kind, st, adapt, sl=None, **kw):
    from pysgmcmc_amd import kernels
    r = (lambda t: t) if sl is None else (lambda t: t[sl])
    if kind == "sghmc":
        kernels.sghmc_step(r(st["theta"]), r(st["V"]), r(st["grad"]), r(st["tau"]), r(st["g"]), r(st["v_hat"]), r(st["minv"]),
                           None, 0.01, 50.0, 0.05, adapt, **kw)
    elif kind == "sgld":
        kernels.sgld_step(r(st["theta"]), r(st["grad"]), r(st["tau"]), r(st["g"]), r(st["v_hat"]), r(st["minv"]), None,
                          0.01, 1.0, 50.0, adapt, **kw)
    else:
        kernels.rsghmc_step(r(st["theta"]), r(st["V"]), r(st["grad"]), 0.01, 1.0, 1.0, 1.0, 0.0, **kw)


CASES = [("sghmc", True), ("sghmc", False), ("sgld", True), ("sgld", False), ("rsghmc", False)]


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("kind,adapt", CASES)
def test_slice_launches_equal_the_single_launch(gpu, dt, kind, adapt):
    """A step issued as launches over consecutive slices (first_element = the slice's start) draws the Philox quads the
    single launch draws: every array bit-equal, in any slice order, incl. a ragged last slice; the slices' statistics
    records land side by side in one workspace and add up to the single launch's statistics."""
    from pysgmcmc_amd import kernels
    n = 70_003
    ref = _state(n, dt, gpu)
    got = {k: v.clone() for k, v in ref.items()}
    st_ref, st_got = kernels.StepStats(n, gpu), kernels.StepStats(n, gpu)
    _call(kind, ref, adapt, seed=11, step=7, stats=st_ref)
    cfg = kernels.LaunchConfig(block_threads=128)
    cuts = [0, 4 * 1000, 4 * 9001, 4 * 9002, n]
    spans = list(zip(cuts[:-1], cuts[1:]))
    blocks = [kernels.step_stats_records(hi - lo, cfg) for lo, hi in spans]
    bases = np.concatenate([[0], np.cumsum(blocks)[:-1]])
    for i in (2, 0, 3, 1):                                  # any order
        lo, hi = spans[i]
        _call(kind, got, adapt, sl=slice(lo, hi), seed=11, step=7, stats=st_got, launch=cfg,
              opts=dict(first_element=lo, stats_base=int(bases[i]), stats_total=int(sum(blocks))))
    for k in ref:
        assert torch.equal(ref[k], got[k]), (kind, adapt, k)
    a = kernels.step_stats_finish(st_ref).cpu().numpy()
    b = kernels.step_stats_finish(st_got).cpu().numpy()
    assert np.allclose(a, b, rtol=2e-6 if dt == torch.float32 else 1e-13), (a, b)
    assert a[0] > 0
    with pytest.raises(Exception, match="multiple of 4"):
        _call(kind, got, adapt, sl=slice(2, 10), seed=1, step=0, opts=dict(first_element=2))


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("kind,adapt", CASES)
def test_fused_moments_equal_the_separate_welford_pass(gpu, dt, kind, adapt):
    """K4 folded into the step launch: same mean / m2 bits as sgmcmc_moments_update_* after the step, same chain; on
    the vector path (fused variant), on a misaligned array (the library runs K4 itself) and with injected noise."""
    from pysgmcmc_amd import kernels
    n = 40_001
    for variant in ("vec", "misaligned", "inject"):
        off = 1 if variant == "misaligned" else 0
        base = _state(n + off, dt, gpu, seed=3)
        a = {k: v.clone()[off:] for k, v in base.items()}
        b = {k: v.clone()[off:] for k, v in base.items()}
        xi = torch.randn(n, dtype=dt, device=gpu) if variant == "inject" else None
        mean0, m20 = torch.randn(n + off, dtype=dt, device=gpu)[off:], torch.rand(n + off, dtype=dt, device=gpu)[off:]
        ma, m2a, mb, m2b = mean0.clone(), m20.clone(), mean0.clone(), m20.clone()
        if variant == "misaligned":
            buf = torch.zeros(2 * (n + 1), dtype=dt, device=gpu)
            ma, m2a = buf[1:n + 1], buf[n + 2:2 * n + 2]
            ma.copy_(mean0); m2a.copy_(m20)
        for cnt in (5, 6):
            _call(kind, a, adapt, seed=2, step=cnt, xi=xi, opts=dict(moments=(ma, m2a, cnt)))
            _call(kind, b, adapt, seed=2, step=cnt, xi=xi)
            kernels.moments_update(b["theta"], mb, m2b, cnt)
        for k in a:
            assert torch.equal(a[k], b[k]), (variant, k)
        assert torch.equal(ma, mb) and torch.equal(m2a, m2b), variant


@pytest.mark.parametrize("kind,adapt", CASES)
def test_theta_sq_only_statistics(gpu, kind, adapt):
    """stats_select = THETA_SQ: the same sum theta'^2 bits as the full reduction, the other statistics read 0, the
    chain unchanged; one 32-byte record per block (block-major)."""
    from pysgmcmc_amd import kernels
    n = 123_457
    a, b = _state(n, torch.float32, gpu, 5), _state(n, torch.float32, gpu, 5)
    sa, sb = kernels.StepStats(n, gpu), kernels.StepStats(n, gpu)
    _call(kind, a, adapt, seed=4, step=1, stats=sa)
    _call(kind, b, adapt, seed=4, step=1, stats=sb, opts=dict(theta_sq_only=True))
    for k in a:
        assert torch.equal(a[k], b[k])
    fa, fb = kernels.step_stats_finish(sa).cpu().numpy(), kernels.step_stats_finish(sb).cpu().numpy()
    assert fa[0] == fb[0] and fb[0] > 0 and np.all(fb[1:] == 0.0)
    assert np.isclose(fa[0], float((a["theta"].double() ** 2).sum()), rtol=5e-7)
    recs = sb.workspace.view(torch.float64)
    nrec = int(sb.workspace.view(torch.int64)[0])
    assert nrec == (n // 4 + 255) // 256                          # one record per block
    rec = recs[4:4 + 4 * nrec].view(nrec, 4)
    assert float(rec[:, 0].sum()) == pytest.approx(fb[0], rel=1e-12) and float(rec[:, 1:].abs().sum()) == 0.0


@pytest.mark.parametrize("kind", ["sghmc", "sgld"])
def test_skip_minv_store(gpu, kind):
    """SGMCMC_STEP_SKIP_MINV_STORE: a burn-in step that leaves minv untouched and everything else as usual."""
    n = 10_007
    a, b = _state(n, torch.float32, gpu, 6), _state(n, torch.float32, gpu, 6)
    minv0 = b["minv"].clone()
    _call(kind, a, True, seed=4, step=1)
    _call(kind, b, True, seed=4, step=1, opts=dict(skip_minv_store=True))
    assert torch.equal(b["minv"], minv0) and not torch.equal(a["minv"], minv0)
    for k in a:
        if k != "minv":
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("kind,adapt", CASES)
def test_device_scalars_override_the_by_value_scalars(gpu, dt, kind, adapt):
    """scalars_dev filled by sgmcmc_*_scalars_* for stepsize e gives the step a by-value stepsize e gives, whatever
    stepsize is passed by value next to it (that is what a replayed graph does)."""
    from pysgmcmc_amd import kernels
    n = 5_003
    a, b = _state(n, dt, gpu, 8), _state(n, dt, gpu, 8)
    sc = torch.zeros(8, dtype=dt, device=gpu)
    for eps in (0.01, 0.0037):
        args = {"sghmc": (eps, 50.0, 0.05), "sgld": (eps, 1.0, 50.0), "rsghmc": (eps, 1.0, 1.0, 1.0, 0.0)}[kind]
        kernels.step_scalars(kind, sc, *args)
        # b is called with the WRONG by-value stepsize 0.01 (from _call) but the right device block
        _call(kind, b, adapt, seed=3, step=2, opts=dict(scalars_dev=sc))
        from pysgmcmc_amd import kernels as K
        if kind == "sghmc":
            K.sghmc_step(a["theta"], a["V"], a["grad"], a["tau"], a["g"], a["v_hat"], a["minv"], None, eps, 50.0, 0.05, adapt,
                         seed=3, step=2)
        elif kind == "sgld":
            K.sgld_step(a["theta"], a["grad"], a["tau"], a["g"], a["v_hat"], a["minv"], None, eps, 1.0, 50.0, adapt, seed=3, step=2)
        else:
            K.rsghmc_step(a["theta"], a["V"], a["grad"], eps, 1.0, 1.0, 1.0, 0.0, seed=3, step=2)
        for k in a:
            assert torch.equal(a[k], b[k]), (kind, eps, k)


def _bnn_chain(gpu, ctor, graph, moments_every=0, fused_moments=True, steps=14, **kw):
    from pysgmcmc_amd.data_batches import Placeholder, generate_batches
    from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments
    from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
    from pysgmcmc_amd.profiling import UpdateKernelTimer
    from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
    rng = np.random.RandomState(0)
    X, y = rng.rand(400, 16), rng.rand(400)
    xp, yp = Placeholder(dtype=torch.float32, device=gpu), Placeholder(dtype=torch.float32, device=gpu)
    params = init_mlp_params(16, hidden=(96, 128, 64), seed=5, dtype=torch.float32, device=gpu)
    cost = BNNCost(xp, yp, batch_size=32, n_examples=400)
    s = ctor(params=params, cost_fun=cost, batch_generator=generate_batches(X, y, xp, yp, batch_size=32, seed=2),
             stepsize_schedule=ConstantStepsizeSchedule(0.01), session=gpu, dtype=torch.float32, seed=9, **kw)
    s.sample_format = "view"
    s.use_hip_graph = graph
    s.collect_stats = "theta_sq"
    s.kernel_timer = UpdateKernelTimer()
    s.kernel_timer.enabled = True
    m = ChainMoments(s.arena.n, gpu)
    if moments_every and fused_moments:
        s.attach_moments(m, moments_every)
    costs = []
    for i in range(steps):
        costs.append(float(next(s)[1]))
        if moments_every and not fused_moments and (i + 1) % moments_every == 0:
            m.update(s.arena.row("theta"))
    torch.cuda.synchronize()
    return s, m, costs


def test_burn_in_without_minv_stores_gives_the_same_chain(gpu):
    """``store_minv_every_step = False``: only the LAST burn-in step writes minv (it is the one value the frozen steps
    consume, pysgmcmc/samplers/base_classes.py:449-454). Same chain bit for bit through the switch, in every stepping
    mode; ``.minv`` read during burn-in comes from v_hat."""
    from pysgmcmc_amd.samplers import SGHMCSampler, SGLDSampler
    for ctor in (SGHMCSampler, SGLDSampler):
        for graph in (False, True, "full"):
            kw = dict(burn_in_steps=6, scale_grad=400.0)
            a, _, ca = _bnn_chain(gpu, ctor, graph=graph, steps=11, **kw)
            b = None

            def chain_b():
                from pysgmcmc_amd.data_batches import Placeholder, generate_batches
                from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
                from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
                rng = np.random.RandomState(0)
                X, y = rng.rand(400, 16), rng.rand(400)
                xp, yp = Placeholder(dtype=torch.float32, device=gpu), Placeholder(dtype=torch.float32, device=gpu)
                params = init_mlp_params(16, hidden=(96, 128, 64), seed=5, dtype=torch.float32, device=gpu)
                s = ctor(params=params, cost_fun=BNNCost(xp, yp, batch_size=32, n_examples=400),
                         batch_generator=generate_batches(X, y, xp, yp, batch_size=32, seed=2),
                         stepsize_schedule=ConstantStepsizeSchedule(0.01), session=gpu, dtype=torch.float32, seed=9, **kw)
                s.sample_format = "view"
                s.use_hip_graph = graph
                s.collect_stats = "theta_sq"
                s.store_minv_every_step = False
                return s
            b = chain_b()
            minv0 = b.arena.row("minv").clone()
            for i in range(11):
                next(b)
                if i == 3:
                    assert torch.equal(b.arena.row("minv"), minv0)                     # burn-in steps left minv alone ...
                    want = 1.0 / torch.sqrt(b.arena.row("v_hat"))
                    got = torch.from_numpy(np.concatenate([m.ravel() for m in b.minv])).to(gpu)
                    assert torch.allclose(got, want, rtol=1e-6)                          # ... and .minv comes from v_hat
            for row in ("theta", "minv", "tau", "g", "v_hat"):
                assert torch.equal(a.arena.row(row), b.arena.row(row)), (ctor.__name__, graph, row)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("kind,adapt", CASES)
def test_window_gather_rides_in_the_step_launch(gpu, dt, kind, adapt):
    """``opts.gather_*`` (round 6): the NEXT step's minibatch window -- ``sgmcmc_window_gather_*``'s job,
    pysgmcmc/data_batches.py:118-123 -- copied by extra workgroups in front of the update's. The update of every element, the
    statistics records and the noise stream are those of the plain launch bit for bit; the window lands in the pitched feed buffer
    with the columns behind the data untouched; launch variants without a fused form (grid-capped, element-wise) still deliver it."""
    from pysgmcmc_amd import kernels
    n, B, D, N = 70_003, 32, 12, 500
    g = torch.Generator(device=gpu).manual_seed(3)
    X = torch.randn(N, D, dtype=dt, device=gpu, generator=g)
    y = torch.randn(N, dtype=dt, device=gpu, generator=g)
    for variant in ("single pass", "grid-capped", "element-wise"):
        ref = _state(n + 1, dt, gpu)
        got = {k: v.clone() for k, v in ref.items()}
        sl = slice(1, n + 1) if variant == "element-wise" else slice(0, n)          # a misaligned slice takes the element-wise path
        launch = kernels.LaunchConfig(max_blocks=40) if variant == "grid-capped" else None
        st_ref, st_got = kernels.StepStats(n, gpu), kernels.StepStats(n, gpu)
        xbuf = torch.full((B, D + 4), -7.0, dtype=dt, device=gpu)
        ybuf = torch.full((B,), -7.0, dtype=dt, device=gpu)
        start = 100 if dt == torch.float64 else 96                                   # 16-byte aligned source windows
        _call(kind, ref, adapt, sl=sl, seed=11, step=7, stats=st_ref, launch=launch)
        _call(kind, got, adapt, sl=sl, seed=11, step=7, stats=st_got, launch=launch,
              opts=dict(gather=(X, y, start, xbuf[:, :D], ybuf)))
        for k in ref:
            assert torch.equal(ref[k], got[k]), (variant, k)
        assert torch.equal(kernels.step_stats_finish(st_ref), kernels.step_stats_finish(st_got)), variant
        assert torch.equal(xbuf[:, :D], X[start:start + B]) and torch.all(xbuf[:, D:] == -7.0), variant
        assert torch.equal(ybuf, y[start:start + B]), variant
    # what cannot ride is refused, loudly: rows that are no multiple of 16 bytes, a misaligned window, another dtype
    st = _state(4096, dt, gpu)
    bad_rows = torch.randn(N, 3, dtype=dt, device=gpu)
    with pytest.raises(ValueError):
        _call(kind, st, adapt, opts=dict(gather=(bad_rows, y, 0, torch.empty(B, 3, dtype=dt, device=gpu), ybuf)))
    assert kernels.gather_fits_step_launch(X, y, 1, xbuf[:, :D], ybuf)              # (rows of 48 / 96 bytes: every window is aligned)
    assert not kernels.gather_fits_step_launch(X.view(-1)[1:1 + 40 * D].view(40, D), y[:40], 0, xbuf[:, :D], ybuf)   # a dataset that starts off a 16-byte boundary
    assert not kernels.gather_fits_step_launch(X, y, N - 3, xbuf[:, :D], ybuf)      # runs off the end of the dataset
    other = torch.float64 if dt == torch.float32 else torch.float32
    assert not kernels.gather_fits_step_launch(X.to(other), y.to(other), 0, xbuf[:, :D], ybuf)
