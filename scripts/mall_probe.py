#!/usr/bin/env python3
"""Streaming-copy rate against buffer size: does a working set below the 256 MB Infinity Cache run faster than HBM?
(rbc_copy_ceiling: device-to-device copy kernel, bytes read + written, best of five grid sizes)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rbc-gym_amd"))
from rbc_gym import _native
for mb in (16, 32, 64, 96, 128, 192, 256, 512, 1024):
    r = _native.copy_ceiling(0, mb << 20, 20)
    print(f"{mb:5d} MiB buffer (working set {2 * mb} MiB): kernel {r['kernel_gbs']:.0f} GB/s, memcpy {r['memcpy_d2d_gbs']:.0f} GB/s", flush=True)
