"""TEST INFRASTRUCTURE ONLY -- CPU restatement of adaptive rejection sampling for the initial relativistic
momenta. Only tests/ may import this; the product never does.

What the reference does (``pysgmcmc/samplers/relativistic_sghmc.py:143-223``): it draws one initial momentum per
parameter tensor (``n_params=len(self.params)``, ``:108-113``) from the 1-D law

    p(p) ~ exp(-m c^2 sqrt(p^2 / (m^2 c^2) + 1))                              (``:208-216``)

with ``arspy.ars.adaptive_rejection_sampling(logpdf, a=-10.0, b=10.0, domain=(-inf, inf), n_samples, seed)``
(``:219-223``). ``arspy`` (``requirements.txt:10``, unpinned, un-vendored, not installable here) implements the
DERIVATIVE-FREE adaptive rejection sampler of

    W. R. Gilks & P. Wild, "Adaptive rejection sampling for Gibbs sampling", Appl. Statist. 41 (1992) 337-348,
    W. R. Gilks, "Derivative-free adaptive rejection sampling for Gibbs sampling", Bayesian Statistics 4 (1992),

which is restated below from the published algorithm:

  * abscissae S (sorted) with log-density values; initial mesh from the two start points ``a < b`` (for an
    unbounded domain the log-density must rise at ``a`` and fall at ``b``);
  * LOWER hull = the chords between consecutive abscissae (a concave function lies above its chords);
  * UPPER hull on [S_i, S_{i+1}] = min of the two neighbouring chords extended into the interval (a concave
    function lies below the extension of a chord outside the chord's own interval); the outermost chords extended
    to the domain ends cover the tails;
  * draw x from the normalised exp(upper hull) (piecewise exponential: pick a piece by its mass, invert its CDF),
    draw U ~ U(0,1): accept if log U <= lower(x) - upper(x) (squeeze test, no density evaluation), else evaluate
    the log-density, accept if log U <= logpdf(x) - upper(x); whenever the density was evaluated, x joins S.

**Parity unpinned** (SURVEY.md 8(c)): neither arspy nor any of its outputs is in the reference repository, so its
random-number consumption cannot be reproduced; what CAN be checked, and is (tests/test_relativistic_momentum.py),
is that this sampler, the reference's target law (quadrature CDF) and the product's inverse-CDF sampler agree in
distribution -- ARS is an exact sampler, so every correct implementation has exactly this law.
"""
import math

import numpy as np

__all__ = ["relativistic_logpdf", "adaptive_rejection_sampling", "sample_relativistic_momentum",
           "relativistic_cdf_table"]


def relativistic_logpdf(m, c):
    """``relativistic_sghmc.py:208-216``."""
    def logpdf(p):
        return -m * c ** 2 * np.sqrt(p ** 2 / (m ** 2 * c ** 2) + 1.0)
    return logpdf


class _Piece(object):
    """One linear piece ``slope * x + icpt`` of the upper hull on ``[left, right]``."""
    __slots__ = ("left", "right", "slope", "icpt", "logmass")

    def __init__(self, left, right, slope, icpt):
        self.left, self.right, self.slope, self.icpt = left, right, slope, icpt
        self.logmass = self._logmass()

    def _logmass(self):
        # log of integral_left^right exp(slope x + icpt) dx
        a, b, s, i = self.left, self.right, self.slope, self.icpt
        if abs(s) < 1e-12:
            return i + math.log(b - a)
        hi, lo = (b, a) if s > 0 else (a, b)                      # exp(s*hi) is the larger term
        if math.isinf(lo):
            return i + s * hi - math.log(abs(s))
        return i + s * hi + math.log1p(-math.exp(s * (lo - hi))) - math.log(abs(s))

    def invert(self, u):
        """x with (mass of [left, x]) = u * (mass of the piece)."""
        a, b, s = self.left, self.right, self.slope
        if abs(s) < 1e-12:
            return a + u * (b - a)
        if math.isinf(a):                                          # s > 0: cdf(x) = exp(s (x - b))
            return b + math.log(u) / s if u > 0.0 else b - 745.0 / s
        if math.isinf(b):                                          # s < 0: 1 - cdf(x) = exp(s (x - a))
            return a + math.log1p(-u) / s
        # finite piece: exp(s x) = exp(s a) + u (exp(s b) - exp(s a)), written stably around the larger end
        if s > 0:
            return b + math.log(u + (1.0 - u) * math.exp(s * (a - b))) / s
        return a + math.log((1.0 - u) + u * math.exp(s * (b - a))) / s


def _chord(x0, f0, x1, f1):
    s = (f1 - f0) / (x1 - x0)
    return s, f0 - s * x0


def _upper_hull(S, fS, domain):
    """Pieces of the derivative-free upper hull over ``domain`` for sorted abscissae S (len >= 3)."""
    n = len(S)
    chords = [_chord(S[i], fS[i], S[i + 1], fS[i + 1]) for i in range(n - 1)]
    pieces = []
    lo, hi = domain
    if lo < S[0]:
        s, i = chords[0]
        if math.isinf(lo) and s <= 0:
            raise ValueError("log-density must increase at the left start point for a domain unbounded to the left")
        pieces.append(_Piece(lo, S[0], s, i))
    for k in range(n - 1):
        left, right = S[k], S[k + 1]
        cands = []
        if k - 1 >= 0:
            cands.append(chords[k - 1])                            # chord to the left, extended rightwards
        if k + 1 <= n - 2:
            cands.append(chords[k + 1])                            # chord to the right, extended leftwards
        if len(cands) == 1:
            pieces.append(_Piece(left, right, *cands[0]))
            continue
        (s0, i0), (s1, i1) = cands
        if abs(s0 - s1) < 1e-14:
            pieces.append(_Piece(left, right, s0, min(i0, i1)))
            continue
        x = (i1 - i0) / (s0 - s1)                                  # where the two extended chords cross
        if not (left < x < right):
            # numerically degenerate (nearly collinear points): the lower of the two at the midpoint bounds the interval
            mid = 0.5 * (left + right)
            pieces.append(_Piece(left, right, *(cands[0] if s0 * mid + i0 <= s1 * mid + i1 else cands[1])))
            continue
        pieces.append(_Piece(left, x, s0, i0))
        pieces.append(_Piece(x, right, s1, i1))
    if hi > S[-1]:
        s, i = chords[-1]
        if math.isinf(hi) and s >= 0:
            raise ValueError("log-density must decrease at the right start point for a domain unbounded to the right")
        pieces.append(_Piece(S[-1], hi, s, i))
    return pieces


def _eval_upper(pieces, x):
    for pc in pieces:
        if pc.left <= x <= pc.right:
            return pc.slope * x + pc.icpt
    raise AssertionError("x outside the hull")


def _eval_lower(S, fS, x):
    if x < S[0] or x > S[-1]:
        return -math.inf
    k = min(max(int(np.searchsorted(S, x, side="right")) - 1, 0), len(S) - 2)
    s, i = _chord(S[k], fS[k], S[k + 1], fS[k + 1])
    return s * x + i


def adaptive_rejection_sampling(logpdf, a, b, domain, n_samples, seed=None, stats=None):
    """``n_samples`` exact draws from ``exp(logpdf)`` (log-concave) -- the call of ``relativistic_sghmc.py:219-223``.

    ``stats`` (optional dict) receives ``{"evaluations", "proposals", "abscissae"}``."""
    assert callable(logpdf)
    assert len(domain) == 2 and domain[1] >= domain[0]
    assert n_samples >= 0
    if a >= b or math.isinf(a) or math.isinf(b) or a < domain[0] or b > domain[1]:
        raise ValueError("invalid start points a, b")
    rng = np.random.RandomState(seed)
    h = 1e-3 * (b - a)
    S = sorted(set([a, a + h] + list(np.linspace(a + h, b - h, 5)) + [b - h, b]))
    fS = [float(logpdf(s)) for s in S]
    evaluations = len(S)
    proposals = 0
    pieces = _upper_hull(S, fS, domain)
    samples = []
    while len(samples) < n_samples:
        logm = np.array([pc.logmass for pc in pieces])
        w = np.exp(logm - logm.max())
        w /= w.sum()
        k = int(np.searchsorted(np.cumsum(w), rng.rand(), side="right"))
        k = min(k, len(pieces) - 1)
        x = pieces[k].invert(rng.rand())
        proposals += 1
        upper = _eval_upper(pieces, x)
        log_u = math.log(rng.rand())
        if log_u <= _eval_lower(S, fS, x) - upper:                 # squeeze test
            samples.append(x)
            continue
        fx = float(logpdf(x))
        evaluations += 1
        if log_u <= fx - upper:                                    # rejection test
            samples.append(x)
        if x not in S:
            pos = int(np.searchsorted(S, x))
            S.insert(pos, x)
            fS.insert(pos, fx)
            pieces = _upper_hull(S, fS, domain)
    if stats is not None:
        stats.update(evaluations=evaluations, proposals=proposals, abscissae=len(S))
    return samples


def sample_relativistic_momentum(m, c, n_params, bounds=(float("-inf"), float("inf")), seed=None):
    """``_sample_relativistic_momentum`` of the reference (``relativistic_sghmc.py:143-223``): a list of
    ``n_params`` floats, one per parameter TENSOR at the reference's call site (``:108-113``)."""
    assert isinstance(m, float)
    assert isinstance(c, float)
    return adaptive_rejection_sampling(logpdf=relativistic_logpdf(m, c), a=-10.0, b=10.0, domain=bounds,
                                       n_samples=n_params, seed=seed)


def relativistic_cdf_table(m, c, p_max=60.0, knots=400001):
    """(p, cdf) of the target law by trapezoid quadrature on a fine grid -- the analytic reference the samplers
    are KS-tested against (tail mass beyond |p| = 60 at m = c = 1 is < 1e-25)."""
    p = np.linspace(-p_max, p_max, knots)
    pdf = np.exp(relativistic_logpdf(m, c)(p))
    cdf = np.concatenate([[0.0], np.cumsum(0.5 * (pdf[1:] + pdf[:-1]))])
    return p, cdf / cdf[-1]
