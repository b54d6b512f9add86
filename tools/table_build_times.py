"""Host against device builder of the delay table (awpu_hip_build_delay_table / _device): run on an MI355X box."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import importlib, time
pkg = importlib.import_module("beamforming-lk_amd")
S = pkg.synthetic
for name in ("headline", "c4"):
    spec = S.WORKLOADS[name]; xyz = S.geometry(spec)
    pkg.build_delay_table_device(xyz, 8, 8)  # (context creation outside the clock)
    t0 = time.perf_counter(); a = pkg.build_delay_table(xyz, spec.res, spec.res, spec.fov); t1 = time.perf_counter()
    b = pkg.build_delay_table_device(xyz, spec.res, spec.res, spec.fov); t2 = time.perf_counter()
    print(name, "host %.3f s, device %.3f s, equal %s" % (t1 - t0, t2 - t1, (a[0] == b[0]).all() and (a[1].view('u4') == b[1].view('u4')).all()))
