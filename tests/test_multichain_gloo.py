"""The N > 1 path on CPU: world_size-2 `gloo` process group, one chain per rank, no
collective on the data path, ONE all-reduce for R-hat and one all-gather for ESS.
The HIP kernels are replaced by the oracle shim in each rank (no GPU here); what is
tested is the sharding, the exchange protocol and the formulas."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_shim
    from pysgmcmc_amd import kernels
    kernels.sghmc_step = oracle_shim.sghmc_step
    oracle_shim.install_diagnostics()
    from itertools import islice
    from pysgmcmc_amd.diagnostics.sampler_diagnostics import ChainMoments, RhatExchange, cross_chain_rhat, ess_across_ranks
    from pysgmcmc_amd.samplers import SGHMCSampler
    from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

    n = 37
    # independent chains: same target, seed = base + rank, over-dispersed start per chain
    x = torch.full((n,), 3.0 * (rank - 0.5), dtype=torch.float32)
    s = SGHMCSampler(params=[x], cost_fun=lambda p: 0.5 * (p[0] ** 2).sum(), burn_in_steps=50, session="cpu",
                     stepsize_schedule=ConstantStepsizeSchedule(0.1), dtype=torch.float32, seed=100 + rank)
    s.sample_format = "view"
    mom = ChainMoments(n, "cpu")
    kept, trace = [], []
    for t, (sample, cost) in enumerate(islice(s, 650)):
        if t >= 50 and t % 3 == 0:
            mom.update(s.arena.row("theta"))
            kept.append(sample.clone().numpy())
            trace.append([float(cost), float(sample[0]), float(sample[1])])
    rhat, summ = cross_chain_rhat(mom)
    # the non-blocking form gives the same numbers, and the chain may move on in between
    ex = RhatExchange(n, "cpu")
    ex.start(mom)
    next(s)
    rhat2, none = ex.finish()              # in-loop default: no host read of the summary
    assert none is None and ex.exchanges == 1
    summ2 = ex.summary.as_dict()           # read when asked for
    assert torch.equal(rhat, rhat2) and summ == summ2 and not ex.pending
    # SURVEY 8(e) "reduce-scatter + all-gather": parameter-sharded exchange (gloo has no reduce-scatter: the class
    # all-reduces the sharded pack and keeps its own chunk -- same layout, same finish on the shard, same summary)
    rs = RhatExchange(n, "cpu", mode="reduce_scatter")
    assert rs.n_shards == world and rs.shard_len == 20 and rs.n_valid == (20 if rank == 0 else 17)
    rs.start(mom)
    shard, none = rs.finish()
    assert none is None and shard.numel() == 20
    lo = rank * rs.shard_len
    assert torch.equal(shard[:rs.n_valid], rhat[lo:lo + rs.n_valid])        # this rank's shard of the same R-hat
    assert torch.equal(rs.gather(), rhat)                                    # all-gather of the shards = full vector
    ssum = rs.summary.as_dict()
    assert np.isclose(ssum["mean"], summ["mean"], rtol=1e-12) and ssum["max"] == summ["max"]
    # two chains PER RANK (chains that share a GPU): the local packs are added before the one collective, R-hat is over
    # world x 2 chains, in both exchange layouts
    x_b = torch.full((n,), -2.0 + rank, dtype=torch.float32)
    s_b = SGHMCSampler(params=[x_b], cost_fun=lambda p: 0.5 * (p[0] ** 2).sum(), burn_in_steps=50, session="cpu",
                       stepsize_schedule=ConstantStepsizeSchedule(0.1), dtype=torch.float32, seed=500 + rank)
    s_b.sample_format = "view"
    mom_a, mom_b, kept_b = ChainMoments(n, "cpu"), ChainMoments(n, "cpu"), []
    for arr in kept:
        mom_a.update(torch.from_numpy(arr))
    for t, (sample, cost) in enumerate(islice(s_b, 650)):
        if t >= 50 and t % 3 == 0:
            mom_b.update(s_b.arena.row("theta"))
            kept_b.append(sample.clone().numpy())
    ex.start([mom_a, mom_b])
    rhat4 = ex.finish()[0].clone()
    rs.start([mom_a, mom_b])
    shard4 = rs.finish()[0]
    assert torch.equal(shard4[:rs.n_valid], rhat4[lo:lo + rs.n_valid]) and torch.equal(rs.gather(), rhat4)
    with pytest.raises(ValueError):
        short = ChainMoments(n, "cpu")
        short.update(torch.zeros(n))
        ex.start([mom_a, short])                                            # unequal sample counts
    ess = ess_across_ranks(torch.tensor(trace, dtype=torch.float32))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), kept=np.array(kept), rhat=rhat.numpy(),
             kept_b=np.array(kept_b), rhat4=rhat4.numpy(),
             rhat_mean=summ["mean"], rhat_max=summ["max"], ess=np.array(ess), trace=np.array(trace),
             final=s.arena.row("theta").numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_chains_rhat_and_ess_over_gloo(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(world)]
    # chains are independent (different seeds and starts) ...
    assert not np.array_equal(r[0]["final"], r[1]["final"])
    # ... every rank ends with the same R-hat, equal to the oracle's on the gathered chains
    assert np.array_equal(r[0]["rhat"], r[1]["rhat"])
    chains = np.stack([r[0]["kept"], r[1]["kept"]])                    # (m, n_samples, P)
    want = oracle.gelman_rubin(chains)
    assert np.allclose(r[0]["rhat"], want, rtol=2e-3, atol=1e-4)
    assert np.isclose(r[0]["rhat_mean"], want.mean(), rtol=2e-3) and np.isclose(r[0]["rhat_max"], want.max(), rtol=2e-3)
    assert 0.9 < want.mean() < 1.3
    # two chains per rank: R-hat over all four chains
    assert np.array_equal(r[0]["rhat4"], r[1]["rhat4"])
    want4 = oracle.gelman_rubin(np.stack([r[0]["kept"], r[0]["kept_b"], r[1]["kept"], r[1]["kept_b"]]))
    assert np.allclose(r[0]["rhat4"], want4, rtol=2e-3, atol=1e-4) and not np.allclose(r[0]["rhat4"], r[0]["rhat"], rtol=1e-3)
    # ESS: identical on both ranks, equal to the oracle's variogram estimate
    assert np.array_equal(r[0]["ess"], r[1]["ess"])
    traces = np.stack([r[0]["trace"], r[1]["trace"]])                   # (m, n, K)
    for k in range(traces.shape[2]):
        want_ess = oracle.effective_n(traces[:, :, k])
        assert abs(int(r[0]["ess"][k]) - want_ess) <= max(2, 0.01 * want_ess), (k, r[0]["ess"][k], want_ess)


def test_effective_n_and_gelman_rubin_match_oracle(oracle):
    from pysgmcmc_amd.diagnostics.sampler_diagnostics import effective_n, gelman_rubin_from_chains as gelman_rubin
    rng = np.random.default_rng(0)
    iid = rng.normal(size=(4, 1500))
    ar = np.zeros((3, 1500))
    for t in range(1, 1500):
        ar[:, t] = 0.8 * ar[:, t - 1] + rng.normal(size=3)
    for x in (iid, ar):
        assert abs(effective_n(torch.tensor(x)) - oracle.effective_n(x)) <= 1
    assert 50 < effective_n(torch.tensor(ar[:1])) < 600           # one chain: B = 0, still defined
    ch = rng.normal(size=(3, 200, 11)) + rng.normal(size=(3, 1, 11)) * 0.3
    assert np.allclose(gelman_rubin(torch.tensor(ch)).numpy(), oracle.gelman_rubin(ch), rtol=1e-12)


def test_trace_container_and_reference_diagnostics_entry_points(monkeypatch):
    """PYSGMCMCTrace / multitrace / effective_sample_sizes / gelman_rubin with the reference's signatures
    (diagnostics/sample_chains.py, sampler_diagnostics.py:47,118), chains run by the oracle shim."""
    import oracle_shim
    oracle_shim.install(monkeypatch)
    from pysgmcmc_amd.diagnostics import sample_chains, sampler_diagnostics
    from pysgmcmc_amd.diagnostics.objective_functions import banana_log_likelihood, to_negative_log_likelihood
    from pysgmcmc_amd.samplers import SGHMCSampler
    seeds = iter(range(100))

    def get_sampler(session=None):
        s = SGHMCSampler(params=[torch.tensor(0.), torch.tensor(6.)], session="cpu", dtype=torch.float32,
                         cost_fun=to_negative_log_likelihood(banana_log_likelihood), seed=next(seeds), burn_in_steps=20)
        s.param_names = ["x:0", "y:0"]
        return s
    trace = sample_chains.PYSGMCMCTrace.from_sampler(chain_id=0, sampler=get_sampler(), n_samples=50)
    assert len(trace) == 50 and trace.varnames == ["x:0", "y:0"] and trace.get_values("y:0").shape == (50,)
    assert set(trace.point(3)) == {"x:0", "y:0"} and len(trace[10:20]) == 10
    mt = sample_chains.pymc3_multitrace(get_sampler, n_chains=3, samples_per_chain=40)
    assert mt.nchains == 3 and len(mt) == 40 and mt.get_values("x:0").shape == (120,)
    ess = sampler_diagnostics.effective_sample_sizes(get_sampler=get_sampler, n_chains=2, samples_per_chain=200)
    rhat = sampler_diagnostics.gelman_rubin(get_sampler=get_sampler, n_chains=2, samples_per_chain=200)
    assert set(ess) == {"x:0", "y:0"} and set(rhat) == {"x:0", "y:0"}
    assert all(1 <= int(v) <= 400 * 1.5 for v in ess.values()) and all(np.isfinite(v) and v > 0.5 for v in rhat.values())
    with pytest.raises(ValueError):
        trace.get_values("nope")
