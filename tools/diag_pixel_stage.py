import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
import gsplat_amd as gs
from oracle import pyoracle as orc
from util import SeamRaster, make_scene, view_parts, oracle_forward
P,M,D,W,H,seed = 800,1,0,250,130,11
s,cams,views = make_scene(P,M,seed,W,H,n_cams=2)
vp = view_parts(views[1])
sr = SeamRaster(); out,R = sr.forward(s,D,M,vp,W,H)
r,oout,oR = oracle_forward(orc,s,D,M,vp,W,H)
rng = np.random.default_rng(seed); dpix = rng.uniform(-1,1,(3,H,W)).astype(np.float32)
g = sr.backward(dpix); og = r.backward(dpix, want_abs=True)
r64 = orc.Rasterizer(np.float64)
r64.forward(D,M,vp["bg"],W,H,s["loc"],s["sh"],s["opac"],s["scale"],1.0,s["rot"],vp["view"],vp["proj"],vp["campos"],vp["tanx"],vp["tany"])
g64 = r64.backward(dpix)
abs9 = og["abs9"]
for name,(q,stride,c) in {"mean2D.x":(3,3,0),"mean2D.y":(4,3,1),"conic.x":(5,4,0),"conic.y":(6,4,1),"conic.w":(7,4,3),"opac":(8,1,0),"col0":(0,3,0)}.items():
    key = {"mean2D":"dL_dmean2D","conic":"dL_dconic","opac":"dL_dopacity","col0":"dL_dcolor"}[name.split(".")[0]]
    got = g[key].reshape(P,stride)[:,c].astype(np.float64); want = og[key].reshape(P,stride)[:,c].astype(np.float64); w64 = g64[key].reshape(P,stride)[:,c]
    a = np.maximum(abs9[:,q], 1e-3*abs9[:,q].max()+1e-30)
    e = np.abs(got-want)/a; e64 = np.abs(got-w64)/a; eo = np.abs(want-w64)/a
    scale = np.abs(want).max()
    er = np.abs(got-want)/np.maximum(np.abs(want), 1e-3*scale)
    print(f"{name:9s} err/abs9: max {e.max():.2e} #>1e-4 {int((e>1e-4).sum())} | gpu-vs-f64 max {e64.max():.2e} oracle32-vs-f64 max {eo.max():.2e} | rel-to-value max {er.max():.2e} #>1e-4 {int((er>1e-4).sum())}")
key="dL_dmean2D"; got=g[key].reshape(P,3)[:,0].astype(np.float64); want=og[key].reshape(P,3)[:,0].astype(np.float64)
a = np.maximum(abs9[:,3], 1e-3*abs9[:,3].max()+1e-30)
bad = np.nonzero(np.abs(got-want)/a > 1e-4)[0]
sm = r.get("splat_margin"); radii = r.get("radii"); co = r.get("conic_opacity").reshape(P,4); m2 = r.get("means2D").reshape(P,2); tt = r.get("tiles_touched")
fT = r.get("final_T"); nc = r.get("n_contrib")
gfT = sr.get_final_T() if hasattr(sr, "get_final_T") else None
print("image max diff", np.abs(out-oout).max(), "n px > 1e-5:", int((np.abs(out-oout)>1e-5).sum()))
for i in bad:
    print(i, "margin", sm[i], "radius", radii[i], "conic_op", co[i], "mean2D", m2[i], "tiles", tt[i], "got", got[i], "want", want[i], "w64", g64[key].reshape(P,3)[i,0], "abs", abs9[i,3])
    print("   opac got/want/64", g["dL_dopacity"][i], og["dL_dopacity"][i], g64["dL_dopacity"][i], "abs", abs9[i,8])
