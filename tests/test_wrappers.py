"""rbc_gym.wrappers against the formulas of the reference's wrappers:
rbc_normalize_observation.py:66-74, rbc_normalize_reward.py:27-32, rbc_reward_shaping.py:72-140."""
import itertools

import numpy as np
import pytest

from rbc_gym import wrappers as W
from rbc_gym._gym import gym


class _Dummy2D(gym.Env):
    dim, ra, temperature_difference, state_shape = 2, 1e4, [1, 2], [64, 96]

    def __init__(self, seed=0):
        self.rng = np.random.default_rng(seed)
        self.observation_space = gym.spaces.Box(-np.inf, np.inf, (3, 8, 48), np.float32)
        self.action_space = gym.spaces.Box(-1, 1, (12,), np.float32)

    def _obs(self):
        o = self.rng.normal(size=(3, 8, 48)).astype(np.float32)
        o[0] += 1.5
        return o

    def _state(self):
        x = np.linspace(0, 2 * np.pi, 96, endpoint=False)
        s = np.zeros((3, 64, 96), np.float32)
        s[2] = (np.sin(2 * x + self.rng.uniform(0, 6)) + 0.4 * np.sin(3 * x + self.rng.uniform(0, 6)))[None, :]
        return s

    def reset(self, *, seed=None, options=None):
        return self._obs(), {"state": self._state()}

    def step(self, action):
        return self._obs(), -float(self.rng.uniform(1, 4)), False, False, {"state": self._state()}


class _Dummy3D(_Dummy2D):
    dim, ra = 3, 2500.0


def test_normalize_observation_formula_dtype_and_space():
    env = W.RBCNormalizeObservation(_Dummy2D(1), heater_limit=0.75)
    raw = _Dummy2D(1)
    obs, _ = env.reset()
    r, _ = raw.reset()
    assert obs.dtype == np.float32 and obs.shape == (3, 8, 48)
    lo, hi = [1, -1.3, -1.3], [2 + 0.75, 1.3, 1.3]
    for c in range(3):
        want = 1 * (2 * (r[c] - lo[c]) / (hi[c] - lo[c]) - 1)          # float32 array, python-float bounds
        assert want.dtype == np.float32 and np.array_equal(obs[c], want)
    assert env.observation_space.shape == (3, 8, 48) and np.allclose(env.observation_space.high, 1.3)
    assert env.unwrapped.ra == 1e4 and env.action_space.shape == (12,)
    clipped = W.RBCNormalizeObservation(_Dummy2D(1), heater_limit=0.75, u_limit=0.5, clip=True)
    o2, *_ = clipped.step(np.zeros(12, np.float32))
    assert np.abs(o2).max() <= 1.0
    with pytest.raises(ValueError):
        W.RBCNormalizeObservation(_Dummy2D(), heater_limit=0.75, u_limit=None)
    e3 = W.RBCNormalizeObservation(_Dummy3D(), heater_limit=0.75, u_limit=None)
    n = 1.06741877
    assert abs(e3.max_vals[1] - 0.96549382 * 2500.0**n / (2500.0**n + 654.37063331**n)) < 1e-15 and e3.min_vals[3] == -e3.max_vals[3]


def test_normalize_reward():
    e2, e3 = W.RBCNormalizeReward(_Dummy2D(2)), W.RBCNormalizeReward(_Dummy3D(2))
    assert abs(e2.scale - 0.1 * 1e4**0.4) < 1e-12 and abs(e3.scale - 0.22 * 2500.0**0.27) < 1e-12
    raw = _Dummy2D(2)
    _, r, *_ = e2.step(None)
    _, r0, *_ = raw.step(None)
    assert r == (r0 + e2.scale) / (e2.scale - 1)
    assert abs(e2.reward(-1.0) - 1.0) < 1e-12 and abs(e2.reward(-e2.scale)) < 1e-12      # Nu = 1 -> 1, Nu = scale -> 0


def test_find_peaks_matches_scipy():
    sig = pytest.importorskip("scipy.signal")
    rng = np.random.default_rng(0)
    for _ in range(2000):
        x = np.round(rng.normal(size=rng.integers(3, 40)) * rng.choice([1, 3, 0.002]), rng.integers(0, 4))
        assert np.array_equal(W.find_peaks_min_height(x, 0.001), sig.find_peaks(x, height=0.001)[0])


def _cell_distance_restated(uy):
    """rbc_reward_shaping.py:85-140 written once more, with scipy's peak finder"""
    from scipy.signal import find_peaks
    pk = find_peaks(uy, height=0.001)[0]
    xs = np.linspace(0, 2 * np.pi, len(uy), endpoint=False)
    ds = []
    for i, j in itertools.combinations(pk, 2):
        d1 = abs(xs[j] - xs[i])
        d2 = 2 * np.pi - d1
        same = np.all(uy[i:j] > 0) if d1 < d2 else (np.all(uy[j:] > 0) and np.all(uy[:i] > 0))
        ds.append(0 if same else min(d1, d2))
    return max(ds) if ds else 0


def test_cell_distance_cases_and_random_signals():
    pytest.importorskip("scipy.signal")
    x = np.linspace(0, 2 * np.pi, 96, endpoint=False)
    assert W.cell_distance(np.sin(x)) == 0                                  # one cell
    assert abs(W.cell_distance(np.sin(2 * x)) - np.pi) < 1e-12              # two cells half a domain apart
    two_bumps = 1.0 + 0.2 * np.sin(4 * x)                                   # four maxima, never negative: one cell
    assert W.cell_distance(two_bumps) == 0
    assert W.cell_distance(np.zeros(96)) == 0
    rng = np.random.default_rng(3)
    batch = []
    for _ in range(300):
        k = rng.integers(1, 5, size=3)
        uy = sum(rng.normal() * np.sin(kk * x + rng.uniform(0, 6)) for kk in k) + 0.3 * rng.normal()
        batch.append(uy.astype(np.float32))
        assert W.cell_distance(batch[-1]) == _cell_distance_restated(batch[-1])
    batch = np.stack(batch)
    assert np.array_equal(W.cell_distances(batch), np.array([_cell_distance_restated(r) for r in batch], dtype=np.float64))


def test_reward_shaping_wrapper():
    pytest.importorskip("scipy.signal")
    env, raw = W.RBCRewardShaping(_Dummy2D(5), shaping_weight=0.25), _Dummy2D(5)
    env.reset(); raw.reset()
    for _ in range(5):
        _, r, term, trunc, info = env.step(None)
        _, r0, _, _, i0 = raw.step(None)
        cd = _cell_distance_restated(i0["state"][2][31])
        assert info["cell_dist"] == cd and abs(r - (0.75 * r0 + 0.25 * (np.pi - cd) / np.pi)) < 1e-15
    assert env.compute_cell_distances(i0["state"], use_avg=True) == _cell_distance_restated(i0["state"][2].mean(axis=0))


class _DummyVec(gym.vector.VectorEnv):
    dim, ra, temperature_difference, state_shape = 2, 1e4, [1, 2], [64, 96]

    def __init__(self, n, seed):
        self.num_envs = n
        self.envs = [_Dummy2D(seed + e) for e in range(n)]
        self.single_observation_space = self.envs[0].observation_space
        self.single_action_space = self.envs[0].action_space

    def reset(self, *, seed=None, options=None):
        o, i = zip(*[e.reset() for e in self.envs])
        return np.stack(o), {"state": np.stack([x["state"] for x in i])}

    def step(self, actions):
        o, r, *_, i = zip(*[e.step(None) for e in self.envs])
        return np.stack(o), np.array(r), np.zeros(self.num_envs, bool), np.zeros(self.num_envs, bool), {"state": np.stack([x["state"] for x in i])}


def test_vector_wrappers_equal_the_single_env_wrappers():
    pytest.importorskip("scipy.signal")
    n = 5
    vec = W.VectorRBCRewardShaping(W.VectorRBCNormalizeReward(W.VectorRBCNormalizeObservation(_DummyVec(n, 9), heater_limit=0.75)), 0.1)
    singles = [W.RBCRewardShaping(W.RBCNormalizeReward(W.RBCNormalizeObservation(_Dummy2D(9 + e), heater_limit=0.75)), 0.1) for e in range(n)]
    assert vec.num_envs == n and vec.unwrapped.ra == 1e4
    vo, _ = vec.reset()
    so = np.stack([s.reset()[0] for s in singles])
    assert np.array_equal(vo, so) and vec.single_observation_space.shape == (3, 8, 48)
    vo, vr, _, _, vi = vec.step(np.zeros((n, 12), np.float32))
    outs = [s.step(None) for s in singles]
    assert np.array_equal(vo, np.stack([o[0] for o in outs]))
    assert np.allclose(vr, [o[1] for o in outs], rtol=0, atol=1e-15)
    assert np.array_equal(vi["cell_dist"], [o[4]["cell_dist"] for o in outs])
