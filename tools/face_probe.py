import importlib.util, os, sys, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
spec = importlib.util.spec_from_file_location("pbf_sph_amd", os.path.join(ROOT, "pbf-sph_amd", "__init__.py"), submodule_search_locations=[os.path.join(ROOT, "pbf-sph_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["pbf_sph_amd"] = pkg; spec.loader.exec_module(pkg)
sc, side = pkg.scene_dambreak(1<<20, False)
s = pkg.Solver(h=0.1); s.upload(**sc); p = pkg.default_params(4, side)
s.steps(p, 205); s.stage("predict", p).stage("sort", p)
k = s.keys().astype(np.uint64); e,_ = s.extent(); tn = len(s.table())
def c10(v):
    v = v & 0x09249249; v=(v|(v>>2))&0x030C30C3; v=(v|(v>>4))&0x0300F00F; v=(v|(v>>8))&0x030000FF; v=(v|(v>>16))&0x3FF; return v
x,y,z = c10(k), c10(k>>1), c10(k>>2)
print("extent", e, "tableN", tn, "n", len(k))
for nm,a,ex in (("x",x,e[0]),("y",y,e[1]),("z",z,e[2])):
    print(nm, "min", a.min(), "max", a.max(), "==0:", (a==0).sum(), ">=ext-1:", (a+1>=ex).sum())
print("key+1>=tableN", (k+1>=tn).sum())
