import os, sys, tempfile, logging, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.basicConfig(level=logging.INFO)
import numpy as np, torch
import radar_processor_amd as rg
from radar_processor_amd import synthetic, _native
from radar_processor_amd.grid_geometry import CompactCSR
cfg = synthetic.CONFIGS["C4"]
el, az, r = synthetic.sweep_geometry(cfg["n_elev"], cfg["n_az"], cfg["n_gates"])
gx, gy, gz = synthetic.gate_coordinates(el, az, r)
t0 = time.time()
try:
    with tempfile.TemporaryDirectory() as tmp:
        geom = rg.compute_grid_geometry(gx, gy, gz, cfg["grid_shape"], cfg["grid_limits"], tmp, layout="compact")
    torch.cuda.synchronize()
    csr = geom.device_csr()
    print("built in", time.time() - t0, "gate_indices is None:", csr.gate_indices is None)
    c = geom.device_compact()
    print("window_cap", c.window_cap, "max_dict", c.max_dict, "n_dict", c.n_dict, "dict B/pair", 4 * c.n_dict / csr.n_pairs)
    print({w: round(c.fallback_fraction(w), 5) for w in (1024, 1536, 2048, 2560, 3072, 4096, 8192)})
    print("free GB", torch.cuda.mem_get_info()[0] / 1e9)
except Exception as e:
    import traceback; traceback.print_exc()
