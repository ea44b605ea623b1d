#!/bin/bash
for r in 8 16 32; do for t in 128 256 512; do echo -n "rows=$r threads=$t: "; RBC_EXPERIMENT_FFT_ROWS=$r RBC_EXPERIMENT_FFT2D_THREADS=$t timeout -k 10 100 python scripts/stream2d_timing.py 1024 3 128 64 || exit 1; done; done
