"""The chunk-resolve step of the NMS kernel (yfv2_post.hip, step 4): inside a chunk of 64 score-sorted candidates member j is
kept iff it is alive and no KEPT earlier member overlaps it.  The kernel no longer walks the members in order; it iterates
K <- alive & !(S & K) from K = alive with one ballot per round until K reproduces itself.  This file pins the claim the
kernel relies on, on the CPU: for every overlap matrix S (bits only below the diagonal) and every alive mask, the fixed
point exists, is unique, equals the sequential greedy walk, and is reached within 64 rounds - members 0..t-1 are final
after t rounds.  (torchvision.ops.nms's loop, utils/utils.py:271, is the sequential walk.)"""
import numpy as np


def _walk(S, alive):
    kept = np.zeros(64, bool)
    for j in range(64):
        kept[j] = alive[j] and not (S[j] & kept).any()
    return kept


def _fixed_point(S, alive):
    K = alive.copy()
    prefixes_final = []
    for rounds in range(1, 66):
        K2 = alive & ~((S & K[None, :]).any(1))
        prefixes_final.append(K2.copy())
        if (K2 == K).all():
            return K2, rounds, prefixes_final
        K = K2
    raise AssertionError("no fixed point within 65 rounds")


def _cases(rng):
    tri = np.tril(np.ones((64, 64), bool), -1)
    yield np.zeros((64, 64), bool), np.ones(64, bool)                                  # nothing overlaps
    yield tri.copy(), np.ones(64, bool)                                                # everything overlaps everything earlier
    chain = np.zeros((64, 64), bool); chain[np.arange(1, 64), np.arange(0, 63)] = True  # j overlaps j-1 only: the longest dependency chain
    yield chain, np.ones(64, bool)
    for p in (0.02, 0.1, 0.3, 0.7):
        for _ in range(60):
            yield (rng.random((64, 64)) < p) & tri, rng.random(64) < rng.choice([0.3, 0.9, 1.0])
    for _ in range(40):                                                                 # banded overlap (neighbouring boxes), partial chunks
        w = int(rng.integers(1, 6))
        S = np.zeros((64, 64), bool)
        for j in range(64):
            S[j, max(0, j - w):j] = rng.random(min(w, j)) < 0.8
        alive = np.arange(64) < int(rng.integers(1, 65))
        yield S, alive


def test_fixed_point_equals_the_sequential_walk():
    rng = np.random.default_rng(11)
    worst = 0
    for S, alive in _cases(rng):
        want = _walk(S, alive)
        got, rounds, hist = _fixed_point(S, alive)
        assert (got == want).all()
        worst = max(worst, rounds)
        for t, Kt in enumerate(hist, start=1):            # after t rounds members 0 .. t-1 already hold their final value
            assert (Kt[:t] == want[:t]).all()
    assert worst <= 64                                    # the kernel's loop bound; the one-step chain needs the most rounds
