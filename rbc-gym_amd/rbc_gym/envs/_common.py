"""Shared machinery of the two single-env classes.

The reference implements its 2D and 3D envs as two independent classes talking to Julia (src/rbc_gym/envs/rbc2D.py,
rbc3D.py).  Here both are thin descriptions (kwargs -> simulation config, spaces, info keys) on top of one base that
owns the episode logic: seeding quirk, checkpoint episodes, stepping a batch-of-one native simulation, truncation,
error behaviour.  Every behaviour cites the reference line it reproduces.
"""
import logging
import warnings
from enum import IntEnum
from pathlib import Path

import numpy as np

from .._gym import gym
from ..checkpoint import read_checkpoint


class RBCField(IntEnum):
    """channel order of observations / states (rbc2D.py:21-26)"""
    T = 0
    UX = 1
    UY = 2
    P = 3


def pick_checkpoint_episode(num_episodes, seed):
    """The reference draws `idx = rand(1:n)` from Julia's global RNG right after `Random.seed!(seed)`
    (rbc_sim2D_api.jl:44, rbc_sim2D.jl:176-177).  That stream cannot be reproduced outside Julia; this is the
    documented stand-in: deterministic in the seed."""
    return int(np.random.default_rng(int(seed) % (2**63)).integers(0, num_episodes))


def checkpoint_index(checkpoint_idx, num_episodes, seed):
    """0-based row of the episode to load.  `checkpoint_idx` is the reference's kwarg and keeps its meaning: it goes
    straight into Julia's `read(h5, "b")[idx, :, :, :]` (rbc_sim3D.jl:186-192), i.e. it is 1-BASED (eval_sarl.py:45 uses
    checkpoint_idx=1 for the first episode); 0 or > num_episodes is Julia's BoundsError, an IndexError here.  None =
    the seeded draw that stands in for `rand(1:n)`."""
    if checkpoint_idx is None:
        return pick_checkpoint_episode(num_episodes, seed)
    idx = int(checkpoint_idx)
    if not 1 <= idx <= int(num_episodes):
        raise IndexError(f"checkpoint_idx={checkpoint_idx} out of range: episodes are numbered 1..{int(num_episodes)} "
                         "(1-based, as in the reference's Julia reader)")
    return idx - 1


def check_checkpoint_grid(path, data, fields, state_shape):
    """the arrays of a checkpoint go to the device as raw pointers: refuse a file written on another grid"""
    cen = tuple(int(n) for n in state_shape)
    for f in fields:
        want = (cen[0] + 1,) + cen[1:] if f == "w" else cen
        got = tuple(data[f].shape[1:])
        if got != want:
            raise ValueError(f"{path}: dataset {f!r} has grid {got}, this env needs {want} (state_shape={list(cen)})")


def box_stack(shape, bounds):
    """Box over `len(bounds)` channels of `shape`, channel c bounded by bounds[c] = (low, high)."""
    shape = tuple(int(s) for s in shape)
    low = np.stack([np.full(shape, lo, dtype=np.float32) for lo, _ in bounds])
    high = np.stack([np.full(shape, hi, dtype=np.float32) for _, hi in bounds])
    return gym.spaces.Box(low, high, shape=(len(bounds),) + shape, dtype=np.float32)


def temperature_image(field, vmin, vmax):
    """(nz, nx) temperature -> (nx, nz, 3) uint8 through matplotlib's turbo map with z flipped so that the bottom plate
    is at the bottom of a top-left-origin canvas (what rbc2D.py:236-240 hands to pygame)."""
    import matplotlib
    scaled = (np.flip(np.transpose(field), axis=1) - vmin) / (vmax - vmin)
    return matplotlib.colormaps["turbo"](scaled, bytes=True)[:, :, :3]


class _Window:
    """pygame window of the 'human' render mode (rbc2D.py:222-255); created lazily, closed with the env."""

    def __init__(self, size, fps, caption="Rayleigh Benard Convection"):
        self.size, self.fps, self.caption = size, fps, caption
        self.screen = None
        self.clock = None

    def show(self, image, native_size):
        try:
            import pygame
        except ImportError as e:
            raise RuntimeError("render_mode='human' needs pygame, which is not installed") from e
        if self.screen is None:
            pygame.init()
            pygame.display.init()
            self.screen = pygame.display.set_mode(self.size)
            pygame.display.set_caption(self.caption)
            self.clock = pygame.time.Clock()
        canvas = pygame.Surface(native_size)
        pygame.surfarray.blit_array(canvas, image)
        self.screen.blit(pygame.transform.scale(canvas, self.size), (0, 0))
        pygame.event.pump()
        self.clock.tick(self.fps)
        pygame.display.flip()

    def close(self):
        if self.screen is not None:
            import pygame
            pygame.display.quit()
            pygame.quit()
            self.screen = None


class NativeEnvBase(gym.Env):
    """One env = a batch-of-one simulation handle.  Subclasses provide `_fields` (dataset names of a checkpoint
    episode), `_observe()`, `_info()` and `_reward()`."""

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 10}
    _fields = ("b", "u", "w")

    def _setup(self, sim, episode_length, checkpoint, render_mode, logger):
        self.sim = sim
        self.episode_length = episode_length
        self.checkpoint = checkpoint
        self.render_mode = render_mode
        self.logger = logger
        self.closed = False
        self._episodes = None
        logger.info(f"Using Rayleigh number Ra={self.ra}")
        logger.info(f"Using episode length {self.episode_length} timesteps")

    # ---- reset -----------------------------------------------------------------------------------
    def _checkpoint_episode(self, seed):
        path = Path(self.checkpoint)
        self.logger.info(f"Using checkpoint file {path.absolute()}")
        if not path.exists():
            raise FileNotFoundError(f"Checkpoint file {path} does not exist. Please provide a valid checkpoint directory.")
        key = str(path.absolute())
        if self._episodes is None or self._episodes[0] != key:
            self._episodes = (key, read_checkpoint(path))
        data = self._episodes[1]
        missing = [f for f in self._fields if f not in data]
        if missing:
            raise ValueError(f"{path}: checkpoint lacks dataset(s) {missing} needed by this env")
        idx = checkpoint_index(getattr(self, "checkpoint_idx", None), data["num_episodes"], seed)
        check_checkpoint_grid(path, data, self._fields, self.state_shape)
        return [data[f][idx:idx + 1] for f in self._fields]

    def reset(self, seed=None, options=None):
        super().reset(seed=seed)
        # quirk kept from the reference (rbc2D.py:150): reset(seed=None) re-uses the env's seed, so
        # every un-seeded reset of one env object starts the same episode
        seed = self.np_random_seed
        if self.checkpoint:
            self.sim.reset_from_arrays(*self._checkpoint_episode(seed))
        else:
            self.sim.reset(np.array([seed % (2**64)], dtype=np.uint64))
        self.last_action = np.zeros(self.action_space.shape, dtype=self.action_space.dtype)
        return self._observe(), self._info()

    # ---- step ------------------------------------------------------------------------------------
    def _checked_action(self, action):
        return np.asarray(action, dtype=np.float32).reshape((1,) + tuple(self.action_space.shape))

    def step(self, action=None):
        if action is None:                                     # rbc2D.py:164-166
            warnings.warn("No action provided, using zero action")
            action = np.zeros(self.action_space.shape, dtype=np.float32)
        if not self.sim.step(self._checked_action(action)):    # NaN scan of step_simulation (rbc_sim2D_api.jl:91-94)
            self.logger.error("Simulation step failed, probably NaN values in the simulation.")
            raise RuntimeError("Error in simulation step, probably NaN values")
        self.last_obs, self.last_reward, self.last_info = self._observe(), self._reward(), self._info()
        truncated = bool(self.last_info["t"] >= self.episode_length)
        return self.last_obs, self.last_reward, False, truncated, self.last_info      # there is no terminal state

    # ---- teardown --------------------------------------------------------------------------------
    def close(self):
        window = getattr(self, "_window", None)
        if window is not None:
            window.close()
        if getattr(self, "sim", None) is not None:             # shutdown_simulation (rbc_sim3D_api.jl:164-172)
            self.sim.close()
            self.sim = None
        self.closed = True


def env_logger(name, log_dir=None, env_id=None):
    """module logger, optionally with the per-env file handler of rbc3D.py:84-99"""
    logger = logging.getLogger(name if env_id is None else f"{name}.{env_id}")
    if log_dir is not None:
        handler = logging.FileHandler(str(Path(log_dir) / f"env_{env_id}.log"))
        handler.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
        logger.addHandler(handler)
    return logger
