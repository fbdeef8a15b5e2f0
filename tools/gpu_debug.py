import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context
from oracle.binding import Oracle
sc = synth.make_scene(20000, seed=7)
ctx = Context(0); ctx.upload_scene(sc); orc = Oracle(sc)
w,h = 96,54
for md in (1,2,3):
    p = abi.default_params(w, h, spp=1, rr_q=0.0, max_depth=md, flags=abi.FLAG_FIXED_AREA_PATTERN)
    f,_ = ctx.render(p); ref = orc.render(p, threads=8)
    diff = np.abs(f-ref).max(axis=2)
    bad = np.argwhere(diff>1e-6)
    print('max_depth', md, 'bad', len(bad), bad[:10].tolist(), diff.max())
# shadow attenuation at camera hit points
eye = np.asarray(sc.camera.eye); look=np.asarray(sc.camera.look); u=np.asarray(sc.camera.u); v=np.asarray(sc.camera.v)
jj,ii = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
x = ii.ravel()/w-0.5; y = jj.ravel()/h-0.5
d = look[None]+x[:,None]*u[None]+y[:,None]*v[None]; d/=np.linalg.norm(d,axis=1)[:,None]
o = np.repeat(eye[None], len(d), 0)
t,prim,n = orc.intersect(o,d); hit = prim>=0
P = o[hit]+t[hit,None]*d[hit]; fire = P + n[hit]*1e-8*3.0
p = abi.default_params(w, h, spp=1, rr_q=0.0, max_depth=1, flags=abi.FLAG_FIXED_AREA_PATTERN)
for li in range(len(sc.lights)):
    a = ctx.shadow_attenuation(p, li, fire); b = orc.shadow_attenuation(p, li, fire)
    dd = np.abs(a-b).max(axis=1); bad = np.argwhere(dd>1e-9).ravel()
    print('light', li, 'shadow mismatches', len(bad), 'of', len(fire))
    for k in bad[:5]:
        print('   fire', fire[k].tolist(), 'gpu', a[k], 'orc', b[k], 'prim', prim[hit][k], 'n', n[hit][k])
