# A/B timing of the 3D tile-kernel variants at configs[4] (B = 32, graph replay): RBC_DEFER_W x precision, seeded actions (so the
# mean Nusselt number printed at the end doubles as a parity check across the variants).  Run on the GPU box from the repo root.
for v in "RBC_DEFER_W=0" "RBC_DEFER_W=1"; do
  echo "== $v"; env $v python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, "rbc-gym_amd")
from rbc_gym import _native
import torch
for prec in (0, 1):
    B=32
    sim = _native.NativeSim3D(batch=B, shape=(32, 48, 48), ra=1e4, precision=prec)
    sim.reset(np.arange(B, dtype=np.uint64) + 1234)
    g=torch.Generator(device="cuda"); g.manual_seed(1)
    act = (torch.rand((B,8,8), device="cuda", generator=g)*2-1).contiguous(); torch.cuda.synchronize()
    for _ in range(3): sim.step_dev(act.data_ptr())
    sim.synchronize(); t0=time.perf_counter()
    for _ in range(60): sim.step_dev(act.data_ptr())
    sim.synchronize(); dt=(time.perf_counter()-t0)/60
    print("prec", prec, f"{B/dt:.0f} env-steps/s", "nan", int(sim.get_flags().sum()), "Nu", float(sim.get_nusselt().mean()))
    sim.close()
PY
done
