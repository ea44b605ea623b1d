"""CPU: host logic of the drop-in layer -- registry, spaces, checkpoint reader, gym stand-in."""
import os

import numpy as np
import pytest

import rbc_gym
from rbc_gym._gym import gym, HAVE_GYMNASIUM
from rbc_gym.checkpoint import read_checkpoint, write_checkpoint_npz
from rbc_gym.envs.rbc2D import build_spaces, pick_checkpoint_episode, sim_kwargs


def test_registry_ids_and_default_kwargs():
    """src/rbc_gym/__init__.py:4-38"""
    sp = gym.spec("rbc_gym/RayleighBenardConvection2D-v0")
    assert sp.kwargs == {"rayleigh_number": 10_000, "episode_length": 300, "observation_shape": (8, 48),
                         "state_shape": (64, 96), "heater_segments": 12, "heater_limit": 0.75,
                         "heater_duration": 1.5, "checkpoint": None, "use_gpu": False, "render_mode": None}
    sp3 = gym.spec("rbc_gym/RayleighBenardConvection3D-v0")
    assert sp3.kwargs["state_shape"] == (16, 32, 32) and sp3.kwargs["heater_segments"] == 8


def test_spaces_match_reference():
    """rbc2D.py:74-108"""
    a, o = build_spaces([8, 48], 12, 0.75, False)
    assert a.shape == (12,) and a.dtype == np.float32 and a.low.min() == -1 and a.high.max() == 1
    assert o.shape == (3, 8, 48) and o.dtype == np.float32
    assert np.all(o.low[0] == 1) and np.all(o.high[0] == np.float32(2.75))
    assert np.all(np.isneginf(o.low[1:])) and np.all(np.isposinf(o.high[1:]))
    _, o5 = build_spaces([64, 96], 12, 0.75, True)
    assert o5.shape == (5, 64, 96)
    s = a.sample()
    assert s.shape == (12,) and s.dtype == np.float32 and np.all(np.abs(s) <= 1)


def test_julia_order_kwargs():
    """rbc2D.py:145-146: shapes travel reversed (x, z)"""
    k = sim_kwargs(1e5, [8, 48], [64, 96], 12, 0.75, 1.5)
    assert (k["nx"], k["nz"], k["obs_nx"], k["obs_nz"]) == (96, 64, 48, 8) and k["dt_control"] == 1.5 and k["ra"] == 1e5


def test_make_fails_loudly_without_gpu():
    from rbc_gym import _native
    if _native.load_library().rbc_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_native.RbcError):
        gym.make("rbc_gym/RayleighBenardConvection2D-v0")
    with pytest.raises(_native.RbcError):
        gym.make("rbc_gym/RayleighBenardConvection3D-v0")


def test_minimal_hdf5_reader(golden_dir, ckpt_ra1e4):
    ck = read_checkpoint(os.path.join(golden_dir, "ckpt2d_small.h5"))
    assert ck["num_episodes"] == 2 and ck["start_seed"] == 42
    assert ck["b"].shape == (2, 64, 96) and ck["w"].shape == (2, 65, 96)
    for k in "buw":
        assert np.array_equal(ck[k], ckpt_ra1e4[k][:2])
    with pytest.raises(FileNotFoundError):
        read_checkpoint(os.path.join(golden_dir, "missing.h5"))


@pytest.mark.skipif(not os.path.exists("/root/reference/data/checkpoints"), reason="reference data only exists in the build container")
def test_reader_on_the_reference_files(ckpt_ra1e4):
    ck = read_checkpoint("/root/reference/data/checkpoints/train/ckpt_ra10000.h5")   # compact-group (Link message) layout
    assert ck["num_episodes"] == 20 and ck["start_seed"] == 42
    for k in "buw":
        assert np.array_equal(ck[k][:3], ckpt_ra1e4[k])


def test_npz_checkpoint_roundtrip(tmp_path, ckpt_ra1e4):
    p = str(tmp_path / "c.npz")
    write_checkpoint_npz(p, ckpt_ra1e4["b"], ckpt_ra1e4["u"], ckpt_ra1e4["w"], start_seed=7)
    ck = read_checkpoint(p)
    assert ck["num_episodes"] == 3 and ck["start_seed"] == 7 and np.array_equal(ck["u"], ckpt_ra1e4["u"])


def test_checkpoint_episode_choice_is_deterministic():
    assert pick_checkpoint_episode(20, 42) == pick_checkpoint_episode(20, 42)
    picks = {pick_checkpoint_episode(20, s) for s in range(200)}
    assert picks == set(range(20))


@pytest.mark.skipif(HAVE_GYMNASIUM, reason="real gymnasium present")
def test_gym_stand_in_seeding_contract():
    class E(gym.Env):
        pass
    e = E()
    s0 = e.np_random_seed
    e.reset()
    assert e.np_random_seed == s0                  # reset(seed=None) never reseeds (the quirk rbc2D.py:150 relies on)
    e.reset(seed=11)
    assert e.np_random_seed == 11


def test_reference_clock_kwarg_is_validated_before_any_device_call():
    """`reference_clock` ("documented" | "recorded") on both env classes and both vector envs: a wrong
    name is a ValueError whether or not a GPU is there; the two names map onto the C ABI's RBC_CLOCK_* values"""
    from rbc_gym import _native
    from rbc_gym.envs import RayleighBenardConvection2DEnv, RayleighBenardConvection3DEnv
    from rbc_gym.vector import RayleighBenardConvection2DVectorEnv, RayleighBenardConvection3DVectorEnv
    assert _native.CLOCKS == {"documented": 0, "recorded": 1} and _native.clock_code("recorded") == 1 and _native.clock_code(0) == 0
    for make in (lambda: RayleighBenardConvection2DEnv(reference_clock="julia"), lambda: RayleighBenardConvection3DEnv(reference_clock="julia"),
                 lambda: RayleighBenardConvection2DVectorEnv(num_envs=2, reference_clock="julia"),
                 lambda: RayleighBenardConvection3DVectorEnv(num_envs=2, reference_clock="julia")):
        with pytest.raises(ValueError, match="reference_clock"):
            make()


def test_output_pool_never_hands_out_an_array_the_caller_still_holds(monkeypatch):
    """_native.PinnedPool: large outputs come from page-locked arrays that are reused only when nothing but the pool references
    them -- the caller's own name, a view, a dict entry or a list slot all keep an array out of circulation, so what the env returned
    is never written again while it can still be read (the reference returns a new array from every step: rbc2D.py:185-196)."""
    from rbc_gym import _native
    made = []

    def fake_pinned(shape, dtype=np.float32):            # no GPU here: ordinary memory, same bookkeeping
        made.append(1)
        return np.empty(shape, dtype)
    monkeypatch.setattr(_native, "pinned_empty", fake_pinned)
    pool = _native.PinnedPool((4, 3), cap=3)
    a = pool.take()
    ida = id(a)
    b = pool.take()
    assert b is not a                                    # `a` is still held
    del a
    c = pool.take()
    assert id(c) == ida and len(made) == 2               # dropped: back in circulation, nothing new allocated
    view = c[1]                                          # a view keeps its base alive -- and out of the pool
    del c
    d = pool.take()
    assert id(d) != ida and d is not b and len(made) == 3
    info = {"state": d}
    del d
    assert pool.take() is None                           # b, the view's base and the dict's array: all three are held
    del info
    e = pool.take()
    assert e is not None and e is not b and len(made) == 3
    del view, b, e
    assert id(pool.take()) in {id(x) for x in pool._items}

    def no_memory(shape, dtype=np.float32):
        raise MemoryError
    monkeypatch.setattr(_native, "pinned_empty", no_memory)
    empty = _native.PinnedPool((2, 2))
    assert empty.take() is None and empty.cap == 0       # no page-locked memory: the getter falls back to an ordinary array
