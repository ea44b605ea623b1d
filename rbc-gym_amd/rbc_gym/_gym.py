"""`gym` namespace used by rbc_gym: the real gymnasium when it is installed, otherwise a small
from-scratch stand-in with the API subset the reference touches (rbc2D.py, __init__.py,
example/run_2D.py, example/run_vectorized.py): Env, spaces.Box, vector.VectorEnv,
register / make / make_vec, logger.warn.  The stand-in exists because the build and GPU images
ship without gymnasium and nothing can be installed there; semantics follow gymnasium 1.1.x
(the version the reference pins, uv.lock:321-322).
"""
import importlib
import warnings

import numpy as np

try:                                    # pragma: no cover - depends on the image
    import gymnasium as gym             # noqa: F401
    HAVE_GYMNASIUM = True
except ImportError:                     # the stand-in
    HAVE_GYMNASIUM = False

    class _Space:
        def __init__(self, shape, dtype, seed=None):
            self._shape = None if shape is None else tuple(int(s) for s in shape)
            self.dtype = np.dtype(dtype)
            self._np_random = None
            self._seed = seed

        @property
        def shape(self):
            return self._shape

        @property
        def np_random(self):
            if self._np_random is None:
                self.seed(self._seed)
            return self._np_random

        def seed(self, seed=None):
            self._np_random = np.random.default_rng(seed)
            return seed

    class Box(_Space):
        """Continuous box, bounded or not (gymnasium.spaces.Box subset)."""

        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            super().__init__(shape, dtype, seed)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), self.shape).astype(dtype)
                self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), self.shape).astype(dtype)
            self.bounded_below = np.isfinite(self.low)
            self.bounded_above = np.isfinite(self.high)

        def sample(self):
            rng = self.np_random
            out = np.empty(self.shape, dtype=np.float64)
            both = self.bounded_below & self.bounded_above
            neither = ~self.bounded_below & ~self.bounded_above
            lo_only = self.bounded_below & ~self.bounded_above
            hi_only = ~self.bounded_below & self.bounded_above
            out[both] = rng.uniform(self.low[both], self.high[both])
            out[neither] = rng.normal(size=int(neither.sum()))
            out[lo_only] = self.low[lo_only] + rng.exponential(size=int(lo_only.sum()))
            out[hi_only] = self.high[hi_only] - rng.exponential(size=int(hi_only.sum()))
            return out.astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

        def __contains__(self, x):
            return self.contains(x)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

        def __eq__(self, other):
            return (isinstance(other, Box) and self.shape == other.shape and self.dtype == other.dtype
                    and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))

    class Env:
        """gymnasium.Env subset: seeding contract of reset(seed=...) and np_random_seed."""
        metadata = {"render_modes": []}
        render_mode = None
        spec = None
        action_space = None
        observation_space = None
        _np_random = None
        _np_random_seed = None

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random = np.random.default_rng(seed)
                self._np_random_seed = int(seed)

        @property
        def np_random_seed(self):
            if self._np_random_seed is None:
                self._np_random_seed = int(np.random.SeedSequence().entropy % (2**31))
                self._np_random = np.random.default_rng(self._np_random_seed)
            return self._np_random_seed

        @property
        def np_random(self):
            if self._np_random is None:
                _ = self.np_random_seed
            return self._np_random

        @property
        def unwrapped(self):
            return self

        def step(self, action):
            raise NotImplementedError

        def render(self):
            raise NotImplementedError

        def close(self):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            self.close()
            return False

    class _VectorEnv:
        """gymnasium.vector.VectorEnv subset."""
        metadata = {}
        render_mode = None
        spec = None
        closed = False
        num_envs = 0
        single_observation_space = None
        single_action_space = None
        observation_space = None
        action_space = None
        _np_random = None
        _np_random_seed = None

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random = np.random.default_rng(seed)
                self._np_random_seed = seed

        def step(self, actions):
            raise NotImplementedError

        def close(self, **kw):
            self.closed = True

        @property
        def unwrapped(self):
            return self

    class Wrapper(Env):
        """gymnasium.Wrapper subset (1.x semantics: attributes are reached through `unwrapped` /
        `get_wrapper_attr`, spaces and metadata are forwarded unless the wrapper overrides them)."""

        def __init__(self, env):
            self.env = env
            self._observation_space = None
            self._action_space = None

        @property
        def observation_space(self):
            return self.env.observation_space if self._observation_space is None else self._observation_space

        @observation_space.setter
        def observation_space(self, space):
            self._observation_space = space

        @property
        def action_space(self):
            return self.env.action_space if self._action_space is None else self._action_space

        @action_space.setter
        def action_space(self, space):
            self._action_space = space

        @property
        def metadata(self):
            return self.env.metadata

        @property
        def render_mode(self):
            return self.env.render_mode

        @property
        def spec(self):
            return self.env.spec

        @property
        def np_random(self):
            return self.env.np_random

        @property
        def np_random_seed(self):
            return self.env.np_random_seed

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def get_wrapper_attr(self, name):
            if name in self.__dict__ or hasattr(type(self), name):
                return getattr(self, name)
            if hasattr(self.env, "get_wrapper_attr"):
                return self.env.get_wrapper_attr(name)
            return getattr(self.env, name)

        def reset(self, *, seed=None, options=None):
            return self.env.reset(seed=seed, options=options)

        def step(self, action):
            return self.env.step(action)

        def render(self):
            return self.env.render()

        def close(self):
            return self.env.close()

    class ObservationWrapper(Wrapper):
        def reset(self, *, seed=None, options=None):
            obs, info = self.env.reset(seed=seed, options=options)
            return self.observation(obs), info

        def step(self, action):
            obs, reward, terminated, truncated, info = self.env.step(action)
            return self.observation(obs), reward, terminated, truncated, info

        def observation(self, observation):
            raise NotImplementedError

    class RewardWrapper(Wrapper):
        def step(self, action):
            obs, reward, terminated, truncated, info = self.env.step(action)
            return obs, self.reward(reward), terminated, truncated, info

        def reward(self, reward):
            raise NotImplementedError

    class _Forwarded:
        """attribute of a wrapper that reads through to the wrapped env until the wrapper assigns its own"""

        def __set_name__(self, owner, name):
            self.name = name

        def __get__(self, obj, objtype=None):
            if obj is None:
                return self
            own = obj.__dict__.get("_own_" + self.name, None)
            return getattr(obj.env, self.name) if own is None else own

        def __set__(self, obj, value):
            obj.__dict__["_own_" + self.name] = value

    class _VectorWrapper(_VectorEnv):
        """gymnasium.vector.VectorWrapper subset."""
        num_envs = _Forwarded()
        single_observation_space = _Forwarded()
        single_action_space = _Forwarded()
        observation_space = _Forwarded()
        action_space = _Forwarded()
        metadata = _Forwarded()
        render_mode = _Forwarded()
        spec = _Forwarded()
        closed = _Forwarded()

        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):          # spaces, num_envs, ... of the wrapped vector env
            if name.startswith("_") or name == "env":
                raise AttributeError(name)
            return getattr(self.env, name)

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def reset(self, *, seed=None, options=None):
            return self.env.reset(seed=seed, options=options)

        def step(self, actions):
            return self.env.step(actions)

        def close(self, **kw):
            return self.env.close(**kw)

    class _Registry(dict):
        pass

    class _Spec:
        def __init__(self, id, entry_point, kwargs, vector_entry_point=None, max_episode_steps=None):
            self.id, self.entry_point, self.kwargs = id, entry_point, dict(kwargs or {})
            self.vector_entry_point = vector_entry_point
            self.max_episode_steps = max_episode_steps

    class _Logger:
        @staticmethod
        def warn(msg, *a):
            warnings.warn(msg % a if a else msg)

    def _load(entry_point):
        if callable(entry_point):
            return entry_point
        mod, name = entry_point.split(":")
        return getattr(importlib.import_module(mod), name)

    class _Error:
        class NameNotFound(KeyError):
            pass

    class _GymNamespace:
        """Looks like the `gymnasium` module for the symbols rbc_gym and its users need."""
        Env = Env
        Wrapper = Wrapper
        ObservationWrapper = ObservationWrapper
        RewardWrapper = RewardWrapper
        logger = _Logger()
        error = _Error
        registry = _Registry()

        class spaces:
            Box = Box
            Space = _Space

        class vector:
            VectorEnv = _VectorEnv
            VectorWrapper = _VectorWrapper

        class envs:
            class registration:
                pass

        @classmethod
        def register(cls, id, entry_point=None, kwargs=None, vector_entry_point=None, max_episode_steps=None, **_):
            cls.registry[id] = _Spec(id, entry_point, kwargs, vector_entry_point, max_episode_steps)

        @classmethod
        def spec(cls, id):
            if id not in cls.registry:
                raise _Error.NameNotFound(f"Environment `{id}` doesn't exist.")
            return cls.registry[id]

        @classmethod
        def make(cls, id, **kwargs):
            sp = cls.spec(id)
            env = _load(sp.entry_point)(**{**sp.kwargs, **kwargs})
            env.spec = sp
            return env

        @classmethod
        def make_vec(cls, id, num_envs=1, vectorization_mode=None, vector_kwargs=None, wrappers=None, **kwargs):
            sp = cls.spec(id)
            if sp.vector_entry_point is None:
                raise ValueError(f"{id} has no vector entry point")
            env = _load(sp.vector_entry_point)(num_envs=num_envs, **{**sp.kwargs, **kwargs})
            env.spec = sp
            return env

    _GymNamespace.envs.registration.register = _GymNamespace.register
    gym = _GymNamespace
