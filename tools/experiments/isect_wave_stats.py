"""The closest-hit kernel wave by wave (a build with -DMIPT_ANY_DEBUG=1): start, end, chunks drawn and hardware wave slot per
wave of the LAST k_intersect launch of a call (DEPTH=2: level 1, the widest that traces every ray)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context
sc = synth.make_scene(1000000, seed=0xC4)
ctx = Context(0); ctx.upload_scene(sc)
w, h = 1920, 1080
d = torch.zeros(3*w*h, dtype=torch.float64, device="cuda:0")
p = abi.default_params(w, h, spp=int(os.environ.get("SPP", "128")), batch_samples=1 << 28, max_depth=int(os.environ.get("DEPTH", "2")))
for _ in range(2): ctx.render_accumulate(p, d.data_ptr(), 0)
torch.cuda.synchronize()
lib = ctx.lib
lib.mipt_debug_isect.argtypes = [C.c_void_p, C.c_int]
a = np.zeros(16384*4, dtype=np.uint64)
assert lib.mipt_debug_isect(a.ctypes.data, a.size) == 0
a = a.reshape(-1, 4); a = a[a[:,1] > 0]
chunks = ((a[:,2] >> 8) & 0xffffffff).astype(np.float64); hw = (a[:,2] >> 40).astype(np.int64); slot = hw & 0xf
t0 = a[:,0].min(); st = (a[:,0].astype(np.float64)-t0)/100.0; en = (a[:,1].astype(np.float64)-t0)/100.0; span = en.max()
print(f"{len(a)} waves, {a[0,3]} rays, launch span {span:.0f} us; wave start p50 {np.median(st):.0f} max {st.max():.0f}; end min {en.min():.0f} p50 {np.median(en):.0f} max {en.max():.0f}")
print("idle at the end %.2f %% of slots x span, at the start %.2f %%" % (100*(span-en).sum()/(span*len(a)), 100*st.sum()/(span*len(a))))
print("chunks per wave: min %.0f p50 %.0f max %.0f; by wave slot (median): " % (chunks.min(), np.median(chunks), chunks.max()) + ", ".join(f"{v}: {np.median(chunks[slot == v]):.0f}" for v in sorted(set(slot.tolist()))))
