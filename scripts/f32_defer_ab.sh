for r in 1 2 3; do for v in "RBC_DEFER_W=0" "RBC_DEFER_W=1"; do echo -n "$v: "; env $v python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, "rbc-gym_amd")
from rbc_gym import _native
import torch
B=32
sim = _native.NativeSim3D(batch=B, shape=(32, 48, 48), ra=1e4, precision=1)
sim.reset(np.arange(B, dtype=np.uint64) + 1234)
g=torch.Generator(device="cuda"); g.manual_seed(1)
act = (torch.rand((B,8,8), device="cuda", generator=g)*2-1).contiguous(); torch.cuda.synchronize()
for _ in range(5): sim.step_dev(act.data_ptr())
sim.synchronize(); t0=time.perf_counter()
for _ in range(100): sim.step_dev(act.data_ptr())
sim.synchronize(); dt=(time.perf_counter()-t0)/100
print(f"{B/dt:.0f} env-steps/s")
PY
done; done
