"""RBCNormalizeObservation / RBCNormalizeReward (reference: wrappers/rbc_normalize_observation.py,
wrappers/rbc_normalize_reward.py)."""
import numpy as np

from .._gym import gym
from ..envs.rbc3D import RayleighBenardConvection3DEnv


def u_limit_3d(ra):
    """Saturating fit of max |w| over Ra from the 3D flow statistics (rbc_normalize_observation.py:76-80)."""
    w_inf, ra_c, n = 0.96549382, 654.37063331, 1.06741877
    return w_inf * ra**n / (ra**n + ra_c**n)


def normalization_bounds(unwrapped, heater_limit, u_limit):
    """Per-channel (min, max): temperature on [T_cold, T_hot + heater_limit], velocities on [-u_limit, u_limit]
    (rbc_normalize_observation.py:40-57).  u_limit=None is only meaningful for the 3D env."""
    is3d = isinstance(unwrapped, RayleighBenardConvection3DEnv) or getattr(unwrapped, "dim", 2) == 3
    if u_limit is None:
        if not is3d:
            raise ValueError("u_limit must be provided for 2D RBC.")
        u_limit = u_limit_3d(unwrapped.ra)
    t = unwrapped.temperature_difference
    lo = [t[0], -u_limit, -u_limit, -u_limit]
    hi = [t[1] + heater_limit, u_limit, u_limit, u_limit]
    return lo, hi


def normalize_channels(obs, lo, hi, maxval, channel_axis=0):
    """obs[c] <- maxval * (2 * (obs[c] - lo[c]) / (hi[c] - lo[c]) - 1) in place, in the array's own precision and in
    that operation order (rbc_normalize_observation.py:68-69).  All channels of obs must have bounds."""
    n = obs.shape[channel_axis]
    if n > len(lo):
        raise IndexError(f"observation has {n} channels but only {len(lo)} have normalisation bounds")
    ch = np.moveaxis(obs, channel_axis, 0)
    for c in range(n):
        ch[c] = maxval * (2 * (ch[c] - lo[c]) / (hi[c] - lo[c]) - 1)
    return obs


class RBCNormalizeObservation(gym.ObservationWrapper):
    """Normalize the observation to approximately lie in [-maxval, maxval]."""

    def __init__(self, env, heater_limit, maxval=1, u_limit=1.3, eps=0.3, clip=False):
        super().__init__(env)
        self.heater_limit = heater_limit
        self.clip = clip
        self.maxval = maxval
        self.excursion_eps = eps
        self.min_vals, self.max_vals = normalization_bounds(env.unwrapped, heater_limit, u_limit)
        limit = maxval * (1 + eps)
        self.observation_space = gym.spaces.Box(low=-limit, high=limit, shape=env.observation_space.shape, dtype=np.float32)

    def observation(self, obs):
        obs = normalize_channels(obs, self.min_vals, self.max_vals, self.maxval)
        if self.clip:
            obs = np.clip(obs, -self.maxval, self.maxval)
        if np.any(np.abs(obs) > (1 + self.excursion_eps) * self.maxval):
            print(f"Warning: observation exceeds maxval {self.maxval}, namely: {np.max(np.abs(obs))} is the max observed value.")
        return obs


def reward_scale(unwrapped):
    """Nu_max ~ s Ra^a: (0.1, 0.4) in 2D, (0.22, 0.27) in 3D (rbc_normalize_reward.py:13-27)."""
    if isinstance(unwrapped, RayleighBenardConvection3DEnv) or getattr(unwrapped, "dim", 2) == 3:
        s, a = 0.22, 0.27
    else:
        s, a = 0.1, 0.4
    return s * (np.asarray(unwrapped.ra, dtype=np.float64)**a if np.ndim(unwrapped.ra) else unwrapped.ra**a)


class RBCNormalizeReward(gym.RewardWrapper):
    """Normalize the reward (= -Nu, Nu in [1, scale]) to about [0, 1]."""

    def __init__(self, env):
        super().__init__(env)
        self.scale = reward_scale(env.unwrapped)

    def reward(self, reward):
        return (reward + self.scale) / (self.scale - 1)
