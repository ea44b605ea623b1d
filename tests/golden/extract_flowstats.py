#!/usr/bin/env python3
"""Extract the per-step series of the reference's 3D flow-statistics run into a plain .npz fixture.

Source (DATA, not code): /root/reference/experiments/flowstats/flowstats_ra.pkl, written by
experiments/flowstats/flowstats_ra.py:82-91 as {str(Ra): {"nusselt_step", "uv_max_step", "uw_max_step",
"uz_max_step"}} with 300 float64 samples each (one per env.step of the 32x64x64 zero-action run, :27-66).

The file is a pickle, so it is NOT unpickled: nothing from it is executed and no object is constructed.
`pickletools.genops` only tokenises the byte stream into (opcode, literal argument) pairs; this script keeps
the string literals (dict keys, also when they come back through the memo) and the raw little-endian float64
payloads (BINBYTES literals of 8*300 bytes) and pairs them up by order of appearance.  The dtype strings
('f8', '<') seen in the stream are checked, not interpreted.

Output: tests/golden/flowstats_ref_series.npz  (ra[14], nusselt[14,300], umax[14,300], vmax[14,300], wmax[14,300])
  umax/vmax/wmax = max|observation[1]|, max|observation[2]|, max|observation[3]| = the x, y and z velocity of
  the float32 state (rbc3D.py:229-232 channel order b,u,v,w; the reference's key names uv/uw/uz are historical).
"""
import os
import pickletools
import sys

import numpy as np

SRC = "/root/reference/experiments/flowstats/flowstats_ra.pkl"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flowstats_ref_series.npz")
FIELDS = ("nusselt_step", "uv_max_step", "uw_max_step", "uz_max_step")


def walk(data):
    """-> {ra_string: {field: float64[300]}} from the opcode stream alone."""
    memo, n_memo = {}, 0            # memo index -> string literal (only strings are tracked)
    last = None                     # the literal the previous opcode pushed, if it was a string
    ra, field, out = None, None, {}
    seen_f8 = False
    for op, arg, _pos in pickletools.genops(data):
        name = op.name
        if name == "MEMOIZE":
            if last is not None:
                memo[n_memo] = last
            n_memo += 1
            continue                 # keeps `last`
        pushed = None
        if name in ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8"):
            pushed = arg
        elif name in ("BINGET", "LONG_BINGET"):
            pushed = memo.get(arg)
        elif name in ("BINBYTES", "BINBYTES8") and len(arg) == 2400:
            if ra is None or field is None:
                raise SystemExit("payload before its keys")
            out.setdefault(ra, {})[field] = np.frombuffer(arg, dtype="<f8").copy()
            field = None
        if pushed is not None:
            if pushed.isdigit():
                ra = pushed
            elif pushed in FIELDS:
                field = pushed
            elif pushed == "f8":
                seen_f8 = True
        last = pushed
    if not seen_f8:
        raise SystemExit("no float64 dtype marker in the stream")
    return out


def main():
    with open(SRC, "rb") as f:
        series = walk(f.read())
    ras = sorted(series, key=int)
    for r in ras:
        missing = [k for k in FIELDS if k not in series[r]]
        if missing:
            raise SystemExit(f"Ra={r}: missing {missing}")
    arr = {k: np.stack([series[r][k] for r in ras]) for k in FIELDS}
    np.savez_compressed(OUT, ra=np.array([float(r) for r in ras]), nusselt=arr["nusselt_step"], umax=arr["uv_max_step"],
                        vmax=arr["uw_max_step"], wmax=arr["uz_max_step"])
    print(f"{OUT}: {len(ras)} Rayleigh numbers x {arr['nusselt_step'].shape[1]} steps")
    for i, r in enumerate(ras):
        nu = arr["nusselt_step"][i]
        print(f"  Ra={r:>8}: Nu[0..2] = {nu[0]:.8f} {nu[1]:.8f} {nu[2]:.8f}   mean(last 100) = {nu[-100:].mean():.4f}")


if __name__ == "__main__":
    sys.exit(main())
