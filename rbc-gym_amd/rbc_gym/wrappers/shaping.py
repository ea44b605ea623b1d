"""RBCRewardShaping (reference: wrappers/rbc_reward_shaping.py): blends the reward with a measure of how far
apart the Benard cells (up-welling plumes on the mid-height line of the vertical velocity) are."""
import logging

import numpy as np

from .._gym import gym
from ..envs._common import RBCField


def find_peaks_min_height(x, height):
    """Indices of the local maxima of a 1D signal that reach `height`: what scipy.signal.find_peaks(x, height=h)
    returns (strict rise before, strict fall after; a flat top counts once, at its middle sample rounded down;
    the end samples are never peaks)."""
    x = np.asarray(x)
    n = x.size
    out = []
    i = 1
    while i < n - 1:
        if x[i - 1] < x[i]:
            j = i + 1
            while j < n - 1 and x[j] == x[i]:
                j += 1
            if x[j] < x[i]:
                out.append((i + j - 1) // 2)
                i = j
        i += 1
    pk = np.asarray(out, dtype=np.intp)
    return pk[x[pk] >= height] if pk.size else pk


def cell_distance(uy, lx=2 * np.pi, height=0.001):
    """compute_cell_distances (rbc_reward_shaping.py:85-140) on one mid-line signal uy[nx]: the largest periodic
    distance between any two peaks, where a pair counts as 0 if the signal stays positive between them -- over
    [p_i, p_j) when the direct way is the shorter one, else over [p_j, end) and [0, p_i)."""
    uy = np.asarray(uy)
    peaks = find_peaks_min_height(uy, height)
    if peaks.size <= 1:
        return 0
    x = np.linspace(0, lx, uy.size, endpoint=False)
    best = None
    for a in range(peaks.size):
        for b in range(a + 1, peaks.size):
            i, j = peaks[a], peaks[b]
            direct = np.abs(x[j] - x[i])
            around = lx - direct
            d = min(direct, around)
            if direct < around:
                if np.all(uy[i:j] > 0):
                    d = 0
            elif np.all(uy[j:] > 0) and np.all(uy[:i] > 0):
                d = 0
            best = d if best is None else max(best, d)
    return best


def cell_distances(uy_batch, lx=2 * np.pi, height=0.001):
    """cell_distance for every row of uy_batch[B, nx] -> float64[B].  Rows without an interior strict local maximum
    above `height` (the common converged single-cell... and rest states) are settled in one vectorised pass."""
    uy = np.asarray(uy_batch)
    mid = uy[:, 1:-1]
    cand = (uy[:, :-2] <= mid) & (mid >= uy[:, 2:]) & (mid >= height)       # superset of the peaks (keeps flat tops)
    out = np.zeros(uy.shape[0])
    for e in np.nonzero(cand.sum(axis=1) >= 2)[0]:
        out[e] = cell_distance(uy[e], lx, height)
    return out


class RBCRewardShaping(gym.Wrapper):
    """reward <- (1 - w) reward + w (pi - cell_dist) / pi; info["cell_dist"] is added."""

    def __init__(self, env, shaping_weight, debug_cell_dist=False):
        super().__init__(env)
        self.logger = logging.getLogger(__name__)
        self.shaping_weight = shaping_weight
        self.debug_cell_dist = debug_cell_dist      # the reference animates the mid-line here; this build only logs
        self.size_state = env.unwrapped.state_shape

    def step(self, action):
        obs, reward, closed, truncated, info = self.env.step(action)
        cd = self.compute_cell_distances(info["state"])
        w = self.shaping_weight
        reward = (1 - w) * reward + w * ((-cd + np.pi) / np.pi)
        if np.isnan(reward):
            self.logger.error("Reward is NaN")
        info["cell_dist"] = cd
        return obs, reward, closed, truncated, info

    def compute_cell_distances(self, state, use_avg=False):
        uy = state[RBCField.UY].mean(axis=0) if use_avg else state[RBCField.UY][int(self.size_state[0] / 2) - 1]
        cd = cell_distance(uy)
        if self.debug_cell_dist:
            self.logger.info("Distance between cells: %s", cd)
        return cd
