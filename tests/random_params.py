"""Random parameter sets in the reference file format (SURVEY.md 8(d) "random-params stress"):
weights Bernoulli / ternary, thresholds spread around the accumulator's distribution so that both
outcomes of every compare are exercised, unsorted threshold pairs, and a sprinkling of extreme
values (never / always fire).  Written with bnn/params_io.py."""
import numpy as np

from bnn import params_io


def make(directory, network, seed, neg2=0.0):
    """neg2: fraction of the 2-bit weight fields set to -2 (0b10), the value bit flips create"""
    rng = np.random.default_rng(seed)
    cnv = network.startswith("cnv")
    a2 = network.endswith("A2")
    weights, thresholds = [], []
    for l, L in enumerate(params_io.layout(network)):
        mh, mw = L["mh"], L["mw"]
        if L["wbits"] == 1:
            W = rng.choice(np.array([-1, 1], np.int8), size=(mh, mw))
        else:
            W = rng.choice(np.array([-1, 0, 1, -2], np.int8), size=(mh, mw), p=[0.35 - neg2 / 2, 0.3 - neg2 / 2, 0.35, neg2])
        nthr = max(L["nthr"], 1)
        if cnv and l == 0:
            centre, spread, lo, hi = 0.0, 1800.0, -(1 << 23), (1 << 23) - 1     # 2*sum(+-q), 2^-8 units
        elif not a2 or (network == "lfcW1A2" and l == 0 and False):
            centre, spread, lo, hi = mw / 2.0, 2.5 * np.sqrt(mw), -32768, 32767  # popcount of matches
        else:
            centre, spread, lo, hi = 0.0, 2.5 * np.sqrt(mw), -32768, 32767       # signed sums
        if network == "lfcW1A2" and l == 0:
            centre = 0.0                                                          # +-1 products, signed
        T = np.rint(rng.uniform(centre - spread, centre + spread, size=(mh, nthr))).astype(np.int64)
        extreme = rng.random((mh, nthr)) < 0.03
        T = np.where(extreme, rng.choice(np.array([lo, hi, lo + 1, hi - 1]), size=(mh, nthr)), T)
        weights.append(W)
        thresholds.append(np.clip(T, lo, hi))
    params_io.write_params(directory, network, weights, thresholds, classes=[str(i) for i in range(10)])
    return weights, thresholds
