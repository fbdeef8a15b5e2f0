import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context
sc = synth.make_scene(1000000, seed=0xC4)
ctx = Context(0); ctx.upload_scene(sc)
w, h = 1920, 1080
d = torch.zeros(3*w*h, dtype=torch.float64, device="cuda:0")
p = abi.default_params(w, h, spp=16, batch_samples=1 << 28)
for _ in range(2): ctx.render_accumulate(p, d.data_ptr(), 0)
torch.cuda.synchronize()
lib = ctx.lib
lib.mipt_debug_tail.argtypes = [C.c_void_p, C.c_int]
a = np.zeros(8192*8, dtype=np.uint64)
print("rc", lib.mipt_debug_tail(a.ctypes.data, a.size))
a = a.reshape(-1, 8).astype(np.float64)
a = a[a[:,1] > 0]
t0 = a[:,0].min(); start = (a[:,0]-t0)/100.0; dur = (a[:,1]-a[:,0])/100.0  # us at 100 MHz
end = (a[:,1]-t0)/100.0
print("waves", len(a), "launch span us", end.max(), "wave dur us: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (dur.mean(), *np.percentile(dur,[50,90,99]), dur.max()))
print("wave start us: p50 %.0f p90 %.0f max %.0f; end us p50 %.0f p90 %.0f p99 %.0f max %.0f" % (*np.percentile(start,[50,90]), start.max(), *np.percentile(end,[50,90,99]), end.max()))
print("rounds: mean %.1f p50 %.0f p99 %.0f max %.0f" % (a[:,2].mean(), *np.percentile(a[:,2],[50,99]), a[:,2].max()))
print("nodes per wave: mean %.1f max %.0f ; nodes per round %.2f" % (a[:,3].mean(), a[:,3].max(), a[:,3].sum()/a[:,2].sum()))
print("walk us per round %.1f  node us per round %.1f  trav_rounds per round %.1f" % (a[:,4].sum()/100/a[:,2].sum(), a[:,5].sum()/100/a[:,2].sum(), a[:,6].sum()/a[:,2].sum()))
print("share of wave time: walk %.2f node %.2f" % (a[:,4].sum()/ (a[:,1]-a[:,0]).sum(), a[:,5].sum()/(a[:,1]-a[:,0]).sum()))
print("stopped waves", int(a[:,7].sum()))
i = np.argsort(dur)[-5:]
print("slowest waves: dur", dur[i], "rounds", a[i,2], "nodes", a[i,3])
