"""(GPU box) cost of the 'typed' paths (a scene with special particles: obstacles, or ghost copies in slab mode): stage
times of the settled 1 M dam-break with and without ONE obstacle particle."""
import importlib.util, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pbf_sph_amd", os.path.join(ROOT, "pbf-sph_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "pbf-sph_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["pbf_sph_amd"] = pkg; spec.loader.exec_module(pkg)
sc, side = pkg.scene_dambreak(1 << 20, False)
for obstacle in (False, True):
    s = pkg.Solver(h=0.1, flags=pkg.FLAG_STAGE_TIMING)
    sc2 = {k: v.copy() for k, v in sc.items()}
    if obstacle:
        sc2["type"][0] = 1
    s.upload(**sc2); p = pkg.default_params(4, side)
    s.set_option("timing_mask", 0); s.steps(p, 200); s.sync()
    s.set_option("timing_mask", 0xFFFF); s.reset_stage_times(); s.steps(p, 10); s.sync()
    t = s.stage_times()
    print(json.dumps({"obstacle": obstacle, "ms_per_step": {k: round(v[0] / 10, 4) for k, v in t.items()}}))
