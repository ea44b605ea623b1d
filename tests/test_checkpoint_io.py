"""Checkpoint files in the reference's on-disk format: dependency-free HDF5 writer and reader."""
import os
import subprocess

import numpy as np
import pytest

from rbc_gym import checkpoint as ck


def test_hdf5_writer_round_trips_2d_and_matches_reference_layout(tmp_path, golden_dir):
    d = np.load(os.path.join(golden_dir, "ckpt2d_ra10000.npz"))
    p = tmp_path / "ckpt_ra10000.h5"
    ck.write_checkpoint(p, d["b"], d["u"], d["w"], start_seed=42)
    r = ck.read_checkpoint(p)
    assert r["num_episodes"] == d["b"].shape[0] and r["start_seed"] == 42
    for k in "buw":
        assert np.array_equal(r[k], d[k])
    # the layout facts of the reference's own files (h5dump -pH of data/checkpoints/train/ckpt_ra10000.h5):
    # datasets (Nz[+1], 1, Nx, E) float64 little-endian, contiguous, first data block at byte 2048, no gaps
    h = ck._MiniHDF5(str(p))
    E = d["b"].shape[0]
    assert h.read("b").shape == (64, 1, 96, E) and h.read("w").shape == (65, 1, 96, E)
    raw = open(p, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and raw[8] == 0
    first = np.frombuffer(raw, "<f8", 64 * 96 * E, 2048).reshape(64, 1, 96, E)
    assert np.array_equal(np.moveaxis(first, -1, 0)[:, :, 0], d["b"])
    assert len(raw) == 2048 + 8 * E * 96 * (64 + 64 + 65)


def test_hdf5_writer_round_trips_3d(tmp_path):
    rng = np.random.default_rng(0)
    b = rng.normal(size=(2, 8, 6, 10)); u = rng.normal(size=b.shape); v = rng.normal(size=b.shape)
    w = rng.normal(size=(2, 9, 6, 10))
    p = tmp_path / "ckpt3.h5"
    ck.write_checkpoint(p, b, u, w, start_seed=7, v=v)
    r = ck.read_checkpoint(p)
    assert r["num_episodes"] == 2 and r["start_seed"] == 7
    for k, a in (("b", b), ("u", u), ("v", v), ("w", w)):
        assert np.array_equal(r[k], a)


def test_npz_suffix_keeps_the_npz_container(tmp_path):
    b = np.ones((1, 4, 6)); w = np.zeros((1, 5, 6))
    ck.write_checkpoint(tmp_path / "c.npz", b, b, w, start_seed=3)
    r = ck.read_checkpoint(tmp_path / "c.npz")
    assert r["start_seed"] == 3 and np.array_equal(r["w"], w)


@pytest.mark.skipif(not os.path.exists("/opt/conda/bin/python3.9"), reason="no interpreter with h5py on this machine")
def test_written_file_opens_with_libhdf5(tmp_path, golden_dir):
    """libhdf5 (through h5py in the conda interpreter, where present) reads what the writer wrote."""
    d = np.load(os.path.join(golden_dir, "ckpt2d_ra10000.npz"))
    p = tmp_path / "x.h5"
    ck.write_checkpoint(p, d["b"], d["u"], d["w"], start_seed=62)
    code = ("import h5py, numpy as np, sys\n"
            "f = h5py.File(sys.argv[1], 'r')\n"
            "print(int(f.attrs['num_episodes']), int(f.attrs['start_seed']), f['b'].shape, f['w'].shape, repr(float(f['u'][...].sum())))\n")
    try:
        out = subprocess.run(["/opt/conda/bin/python3.9", "-c", code, str(p)], capture_output=True, text=True, timeout=120)
    except (OSError, subprocess.TimeoutExpired):
        pytest.skip("conda interpreter not runnable")
    if out.returncode != 0 and "No module named" in out.stderr:
        pytest.skip("h5py not importable")
    assert out.returncode == 0, out.stderr
    E = d["b"].shape[0]
    n, s = out.stdout.split()[:2]
    assert (int(n), int(s)) == (E, 62)
    assert f"(64, 1, 96, {E})" in out.stdout and f"(65, 1, 96, {E})" in out.stdout
    assert abs(float(out.stdout.strip().rsplit(" ", 1)[1]) - float(d["u"].sum())) < 1e-9


@pytest.mark.gpu
def test_generate_checkpoints_on_device_and_reset_from_them(tmp_path):
    from rbc_gym.generate import generate_checkpoints_2d
    import rbc_gym
    from rbc_gym._gym import gym
    p = generate_checkpoints_2d(str(tmp_path / "train"), ra=1e4, random_inits=4, seed=42, duration=6.0)
    assert os.path.basename(p) == "ckpt_ra10000.h5"
    r = ck.read_checkpoint(p)
    assert r["num_episodes"] == 4 and r["start_seed"] == 42 and r["b"].shape == (4, 64, 96) and r["w"].shape == (4, 65, 96)
    assert np.all(r["w"][:, 0] == 0) and np.all(r["w"][:, -1] == 0)
    div = (np.roll(r["u"], -1, axis=2) - r["u"]) / (2 * np.pi / 96) + (r["w"][:, 1:] - r["w"][:, :-1]) / (2.0 / 64)
    assert np.abs(div).max() < 1e-12                              # P1: states on file are discretely divergence-free
    assert len({r["b"][e].tobytes() for e in range(4)}) == 4      # independent initialisations
    env = gym.make("rbc_gym/RayleighBenardConvection2D-v0", checkpoint=p)
    _, info = env.reset(seed=1)
    st = info["state"]
    assert any(np.allclose(st[0], r["b"][e].astype(np.float32)) for e in range(4))
    env.close()


@pytest.mark.gpu
def test_generate_3d_checkpoints_and_reset_from_them(tmp_path):
    from rbc_gym.generate import generate_checkpoints_3d
    import rbc_gym  # noqa: F401
    from rbc_gym._gym import gym
    p = generate_checkpoints_3d(str(tmp_path / "train"), ra=2500, random_inits=3, seed=42, n=(16, 16, 8), duration=2.0)
    assert os.path.basename(p) == "3D_ckpt_ra2500.h5"
    r = ck.read_checkpoint(p)
    assert r["num_episodes"] == 3 and r["start_seed"] == 42
    assert r["b"].shape == (3, 8, 16, 16) and r["v"].shape == (3, 8, 16, 16) and r["w"].shape == (3, 9, 16, 16)
    assert np.all(r["w"][:, 0] == 0) and np.all(r["w"][:, -1] == 0)
    dx = dy = 4 * np.pi / 16
    div = ((np.roll(r["u"], -1, axis=3) - r["u"]) / dx + (np.roll(r["v"], -1, axis=2) - r["v"]) / dy
           + (r["w"][:, 1:] - r["w"][:, :-1]) / (2.0 / 8))
    assert np.abs(div).max() < 1e-11
    env = gym.make("rbc_gym/RayleighBenardConvection3D-v0", state_shape=(8, 16, 16), rayleigh_number=2500, checkpoint=p, checkpoint_idx=3)
    obs, _ = env.reset(seed=0)       # checkpoint_idx is 1-based like the reference's Julia reader (rbc_sim3D.jl:186-192): 3 = the last episode
    assert np.array_equal(obs[0], r["b"][2].astype(np.float32)) and np.array_equal(obs[3], r["w"][2][:8].astype(np.float32))
    env.close()
