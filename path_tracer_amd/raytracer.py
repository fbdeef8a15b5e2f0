"""Host-side mirror of the reference's `RayTracer` render interface
(ray/src/RayTracer.h:32-56), backed by libmipt.so through the C ABI of include/mipt.h.

`Context` is a thin, explicit wrapper of the C entry points; `RayTracer` keeps the
reference's method names and call order (`loadScene` -> `traceSetup` -> `traceImage`
-> `waitRender` -> `getBuffer`, ui/CommandLineUI.cpp:76-96).  There is no CPU path:
every render call goes to the HIP library or raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import abi
from .scene import Scene, load_json_scene


class MiptError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mipt error {code}: {msg}")
        self.code = code


class Context:
    """One rendering context on one GPU (mipt_create .. mipt_destroy)."""

    def __init__(self, device: int = 0):
        self.lib = abi.load_library()
        h = C.c_void_p()
        rc = self.lib.mipt_create(C.byref(h), device)
        if rc != 0:
            raise MiptError(rc, self.lib.mipt_last_error(None).decode())
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mipt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise MiptError(rc, self.lib.mipt_last_error(self._h).decode())

    def set_option(self, name: str, value: int):
        """mipt_set_option: "builder", "lds_stack", ... (include/mipt.h); takes effect at the next upload."""
        self._check(self.lib.mipt_set_option(self._h, name.encode(), int(value)))

    def upload_scene(self, scene: Scene):
        desc, keep = scene.to_c()
        self._check(self.lib.mipt_upload_scene(self._h, C.byref(desc)))
        del keep

    def render(self, params: abi.Params, want_f32=True, want_rgb8=True):
        """mipt_render: whole frame to host buffers. Returns (f32 (h,w,3) or None, u8 (h,w,3) or None)."""
        w, h = params.width, params.height
        f = np.zeros((h, w, 3), dtype=np.float32) if want_f32 else None
        b = np.zeros((h, w, 3), dtype=np.uint8) if want_rgb8 else None
        self._check(self.lib.mipt_render(self._h, C.byref(params), f.ctypes.data if want_f32 else None, b.ctypes.data if want_rgb8 else None))
        return f, b

    def render_accumulate(self, params: abi.Params, d_sum_ptr: int, stream: int = 0):
        """mipt_render_accumulate into a device buffer of 3*w*h doubles (pointer as int)."""
        self._check(self.lib.mipt_render_accumulate(self._h, C.byref(params), C.c_void_p(d_sum_ptr), C.c_void_p(stream)))

    def render_accumulate_static(self, params: abi.Params, d_sum_ptr: int, stream: int = 0):
        """mipt_render_accumulate_static: the same work out of the library's static device workspace, no allocation."""
        self._check(self.lib.mipt_render_accumulate_static(self._h, C.byref(params), C.c_void_p(d_sum_ptr), C.c_void_p(stream)))

    def resolve(self, d_sum_ptr: int, width: int, height: int, spp: int, d_f32_ptr: int = 0, d_rgb8_ptr: int = 0, stream: int = 0):
        self._check(self.lib.mipt_resolve(self._h, C.c_void_p(d_sum_ptr), width, height, spp, C.c_void_p(d_f32_ptr or None),
                                          C.c_void_p(d_rgb8_ptr or None), C.c_void_p(stream)))

    def trace_pixel(self, params: abi.Params, i: int, j: int) -> np.ndarray:
        out = (C.c_double * 3)()
        self._check(self.lib.mipt_trace_pixel(self._h, C.byref(params), i, j, out))
        return np.array(out[:])

    def intersect(self, org, dir, t_max: float = 1000.0):
        org = np.ascontiguousarray(org, dtype=np.float64).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, dtype=np.float64).reshape(-1, 3)
        n = org.shape[0]
        t = np.empty(n, dtype=np.float64)
        prim = np.empty(n, dtype=np.int32)
        nrm = np.empty((n, 3), dtype=np.float64)
        self._check(self.lib.mipt_intersect(self._h, n, org.ctypes.data, dir.ctypes.data, t_max, t.ctypes.data, prim.ctypes.data, nrm.ctypes.data))
        return t, prim, nrm

    def hit_material(self, org, dir, t_max: float = 1000.0):
        """(uv, kd, ka) at the closest hit of every ray; zeros on a miss."""
        org = np.ascontiguousarray(org, dtype=np.float64).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, dtype=np.float64).reshape(-1, 3)
        n = org.shape[0]
        uv = np.empty((n, 2), dtype=np.float64)
        kd = np.empty((n, 3), dtype=np.float64)
        ka = np.empty((n, 3), dtype=np.float64)
        self._check(self.lib.mipt_hit_material(self._h, n, org.ctypes.data, dir.ctypes.data, t_max, uv.ctypes.data, kd.ctypes.data, ka.ctypes.data))
        return uv, kd, ka

    def shadow_attenuation(self, params: abi.Params, light: int, fire, keys=None):
        fire = np.ascontiguousarray(fire, dtype=np.float64).reshape(-1, 3)
        n = fire.shape[0]
        keys = np.zeros(n, dtype=np.uint64) if keys is None else np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.empty((n, 3), dtype=np.float64)
        self._check(self.lib.mipt_shadow_attenuation(self._h, C.byref(params), light, n, fire.ctypes.data, keys.ctypes.data, out.ctypes.data))
        return out

    def shadow_rays(self, params: abi.Params, P, N, keys, prims, rays_per_hit: int):
        """Ray-by-ray results of the render's own shadow kernels for the given shaded hits: (n, rays_per_hit, 3)."""
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 3)
        N = np.ascontiguousarray(N, dtype=np.float64).reshape(-1, 3)
        n = P.shape[0]
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        prims = np.ascontiguousarray(prims, dtype=np.int32)
        out = np.full((n, rays_per_hit, 3), np.nan, dtype=np.float64)
        self._check(self.lib.mipt_shadow_rays(self._h, C.byref(params), n, P.ctypes.data, N.ctypes.data, keys.ctypes.data, prims.ctypes.data, out.ctypes.data))
        return out

    def counters(self) -> dict:
        c = abi.Counters()
        self._check(self.lib.mipt_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def kernel_times(self) -> dict:
        t = abi.KernelTimes()
        self._check(self.lib.mipt_get_kernel_times(self._h, C.byref(t)))
        return t.as_dict()

    def reset_counters(self):
        self._check(self.lib.mipt_reset_counters(self._h))

    def scene_stats(self) -> dict:
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.mipt_scene_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"prims": a.value, "nodes": b.value, "device_bytes": c.value}


class RayTracer:
    """Same names and call order as the reference class (RayTracer.h:32-56)."""

    def __init__(self, device: int = 0):
        self._ctx = Context(device)
        self.scene: Optional[Scene] = None
        self.buffer: Optional[np.ndarray] = None  # (h, w, 3) uint8, row 0 = bottom
        self.buffer_f32: Optional[np.ndarray] = None
        self.buffer_width = 0
        self.buffer_height = 0
        self.params: Optional[abi.Params] = None
        self.overrides = {}

    def sceneLoaded(self) -> bool:  # noqa: N802 (reference spelling)
        return self.scene is not None

    def loadScene(self, fn) -> bool:  # noqa: N802 — RayTracer.cpp:333-415
        try:
            scene = fn if isinstance(fn, Scene) else load_json_scene(fn)
            self._ctx.upload_scene(scene)
        except (OSError, ValueError, KeyError, NotImplementedError, MiptError) as exc:
            self.last_alert = str(exc)  # the reference reports through traceUI->alert and returns false
            return False
        self.scene = scene
        return True

    def aspectRatio(self) -> float:  # noqa: N802 — RayTracer.cpp:327-330
        return self.scene.camera.aspect if self.sceneLoaded() and hasattr(self.scene.camera, "aspect") else 1.0

    def traceSetup(self, w: int, h: int):  # noqa: N802 — RayTracer.cpp:417-442
        self.buffer_width, self.buffer_height = w, h
        self.buffer = np.zeros((h, w, 3), dtype=np.uint8)
        self.buffer_f32 = np.zeros((h, w, 3), dtype=np.float32)
        self.params = abi.default_params(w, h, **self.overrides)

    def traceImage(self, w: int, h: int):  # noqa: N802 — RayTracer.cpp:462-491 (synchronous)
        self.traceSetup(w, h)
        if not self.sceneLoaded():
            return
        f, b = self._ctx.render(self.params)
        self.buffer_f32, self.buffer = f, b

    def tracePixel(self, i: int, j: int) -> np.ndarray:  # noqa: N802 — RayTracer.cpp:59-101
        if not self.sceneLoaded() or self.params is None:
            return np.zeros(3)
        color = self._ctx.trace_pixel(self.params, i, j)
        px = self.buffer[j, i]
        for k in range(3):
            v = 255.0 * color[k]
            px[k] = int(v) & 0xFF if np.isfinite(v) else 0
        return color

    def checkRender(self) -> bool:  # noqa: N802 — RayTracer.cpp:503-512
        return True

    def waitRender(self):  # noqa: N802 — RayTracer.cpp:514-526
        return None

    def getBuffer(self):  # noqa: N802 — RayTracer.cpp:319-324
        return self.buffer, self.buffer_width, self.buffer_height

    def getPixel(self, i: int, j: int):  # noqa: N802 — RayTracer.cpp:529-534
        return self.buffer[j, i].astype(np.float64) / 255.0
