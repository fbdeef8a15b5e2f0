"""The shadow kernel wave by wave (a build with -DMIPT_ANY_DEBUG=1): when each wave of ONE launch starts and ends, and on
which XCD — how much of the launch's wave slots idle while the last waves finish.  Renders one level only (max_depth 1:
the camera level's hits and their shadow rays) so that the last launch recorded is the big one."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from path_tracer_amd import abi, synth
from path_tracer_amd.raytracer import Context
sc = synth.make_scene(1000000, seed=0xC4)
ctx = Context(0); ctx.upload_scene(sc)
w, h = 1920, 1080
d = torch.zeros(3*w*h, dtype=torch.float64, device="cuda:0")
for spp in (int(x) for x in os.environ.get("SPP", "16").split(",")):
    p = abi.default_params(w, h, spp=spp, batch_samples=1 << 28, max_depth=int(os.environ.get('DEPTH', '1')))
    for _ in range(2): ctx.render_accumulate(p, d.data_ptr(), 0)
    torch.cuda.synchronize()
    lib = ctx.lib
    lib.mipt_debug_any.argtypes = [C.c_void_p, C.c_int]
    a = np.zeros(16384*4, dtype=np.uint64)
    assert lib.mipt_debug_any(a.ctypes.data, a.size) == 0
    a = a.reshape(-1, 4); a = a[a[:,1] > 0]
    xcd = (a[:,2] & 0xf).astype(np.int64); ndyn = ((a[:,2] >> 8) & 0xffffffff).astype(np.float64); tdyn = a[:,3].astype(np.float64)
    hw = (a[:,2] >> 40).astype(np.int64)
    slot, simd, cu, sh, se = hw & 0xf, (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    widx = np.arange(len(a))
    a = a.astype(np.float64); a[:,2] = xcd
    t0 = a[:,0].min(); st = (a[:,0]-t0)/100.0; en = (a[:,1]-t0)/100.0
    span = en.max()
    print(f"{spp} spp, last level of max_depth {os.environ.get('DEPTH', '1')}: {len(a)} waves, {a[0,3]:.0f} shadow rays, launch span {span:.0f} us")
    print("  wave start us: p50 %.0f p99 %.0f max %.0f;  wave end us: min %.0f p1 %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f" % (*np.percentile(st,[50,99]), st.max(), en.min(), *np.percentile(en,[1,10,50,90]), en.max()))
    print("  wave-slot time idle at the end: %.2f %% of slots x span;  at the start %.2f %%" % (100*(span-en).sum()/(span*len(a)), 100*st.sum()/(span*len(a))))
    td = (tdyn - t0)/100.0
    print("  end of the static rounds us: min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f;  dynamic chunks per wave: min %.0f p50 %.0f max %.0f" % (td.min(), *np.percentile(td,[10,50,90]), td.max(), ndyn.min(), np.median(ndyn), ndyn.max()))
    late = en > np.percentile(en, 95)
    print("  the 5 %% of waves that end last: static rounds ended at p50 %.0f us, dynamic chunks p50 %.0f, end p50 %.0f" % (np.median(td[late]), np.median(ndyn[late]), np.median(en[late])))
    for name, key in (("wave slot", slot), ("SIMD", simd), ("CU id", cu), ("SE", se), ("wave in block", widx % 4), ("block order (launch index // 1024)", widx // 1024)):
        vals = sorted(set(key.tolist()))
        print(f"  static rounds end (us, median) by {name}: " + ", ".join(f"{v}: {np.median(td[key == v]):.0f}" for v in vals[:16]))
    for x in range(8):
        m = a[:,2] == x
        if m.any(): print(f"  XCD {x}: {m.sum()} waves, static ends p50 {np.median(td[m]):.0f} max {td[m].max():.0f}, dynamic chunks p50 {np.median(ndyn[m]):.0f}, median end {np.median(en[m]):.0f}, last end {en[m].max():.0f} us, idle at the end {100*(span-en[m]).sum()/(span*m.sum()):.2f} %")
