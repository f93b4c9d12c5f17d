"""CPU build of the per-lane device functions (tests/host_emul) behind the same Python surface as gi_raytracer_amd.RayTracer.
Test infrastructure: lets the -m "not gpu" suite check the kernel logic against the oracle without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np

import gi_raytracer_amd as gi

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul")
_LIB = None
_dp, _ip, _up = gi._dp, gi._ip, gi._up


def lib():
    global _LIB
    if _LIB is None:
        asan = os.environ.get("GI_EMUL_ASAN") == "1"   # sanitizer build: LD_PRELOAD=$(gcc -print-file-name=libasan.so) GI_EMUL_ASAN=1 pytest ...
        subprocess.run(["make", "-C", _DIR] + (["asan"] if asan else []), check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(_DIR, "libgi_emul_asan.so" if asan else "libgi_emul.so"))
        vp = C.c_void_p
        L.emul_create.restype = vp
        L.emul_destroy.argtypes = [vp]
        L.emul_error.argtypes = [vp]
        L.emul_error.restype = C.c_char_p
        L.emul_upload_scene.argtypes = [vp, C.POINTER(gi.SceneDesc)]
        L.emul_upload_photons.argtypes = [vp, C.POINTER(gi.PhotonMapDesc)]
        L.emul_trace.argtypes = [vp, C.c_int, _dp, _ip, _ip, _dp]
        L.emul_visible.argtypes = [vp, C.c_int, _dp, _ip]
        L.emul_visible_turns.argtypes = [vp, C.c_int, _dp, _ip, C.c_int]
        L.emul_gather.argtypes = [vp, C.c_int, _dp, _dp, _ip]
        L.emul_radiance.argtypes = [vp, C.c_int, _dp, _up, C.c_uint64, _dp]
        L.emul_render.argtypes = [vp, C.POINTER(gi.RenderParams), _dp, _ip, C.POINTER(C.c_int64)]
        L.emul_emit.argtypes = [vp, C.c_int, C.c_int, C.c_uint64, _dp, C.c_int, C.POINTER(C.c_int64)]
        L.emul_halton_sample.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.emul_halton_sample.restype = C.c_float
        L.emul_halton_index.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
        L.emul_halton_index.restype = C.c_uint32
        L.emul_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.emul_rng.restype = C.c_double
        L.emul_set_wide.argtypes = [vp, C.c_int]
        L.emul_set_cull.argtypes = [vp, C.c_int]
        L.emul_leaf_order.argtypes = [vp, C.c_int, _dp, C.c_int, _ip, _ip]
        L.emul_kat.argtypes = [C.c_int, C.c_int, _dp, C.c_int, _dp]
        _LIB = L
    return _LIB


class EmulRayTracer(gi.RayTracer):
    """Same methods as gi.RayTracer, executed by the CPU build of gi_device.h."""

    def __init__(self):  # noqa: deliberately not calling the GPU constructor
        self.L = gi.lib()
        self.E = lib()
        self.h = C.c_void_p(self.E.emul_create())
        self.scene = None
        st = gi.Scene().settings
        self.photons, self.photon_depth = st.photons, st.photon_depth
        self.min_samples, self.max_samples, self.noise_thresh = st.min_samples, st.max_samples, st.noise_thresh
        self.cam_pos, self.cam_up, self.cam_forward = list(st.cam_pos), list(st.cam_up), list(st.cam_forward)
        self.sensor_diag, self.focal_dist = st.sensor_diag, st.focal_dist
        self.seed = gi.DEFAULT_SEED
        self._last_counters = None

    def __del__(self):
        try:
            self.E.emul_destroy(self.h)
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc < 0:
            raise gi.GiError(f"{what}: {self.E.emul_error(self.h).decode()}")
        return rc

    def setScene(self, scene, adopt_settings=True):
        d = scene.desc()
        self._ck(self.E.emul_upload_scene(self.h, C.byref(d)), "upload_scene")
        self.scene = scene
        if adopt_settings:
            st = scene.settings
            self.photons, self.photon_depth = st.photons, st.photon_depth
            self.min_samples, self.max_samples, self.noise_thresh = st.min_samples, st.max_samples, st.noise_thresh
            self.cam_pos, self.cam_up, self.cam_forward = list(st.cam_pos), list(st.cam_up), list(st.cam_forward)
            self.sensor_diag, self.focal_dist = st.sensor_diag, st.focal_dist
        return self

    def upload_photon_map(self):
        d = self.scene.photon_desc()
        self._ck(self.E.emul_upload_photons(self.h, C.byref(d)), "upload_photons")

    def tracePhotons(self, count=None, max_depth=5, seed=None):
        count = self.photons if count is None else count
        cap = max(count * max(self.scene.desc().n_light, 1), 1)
        out = np.zeros((cap, 9)); tries = C.c_int64()
        n = self.E.emul_emit(self.h, count, max_depth, C.c_uint64(self.seed if seed is None else seed), gi._p(out), cap, C.byref(tries))
        ph = out[:n].copy()
        self.scene.build_photon_map(ph)
        self.upload_photon_map()
        return ph, tries.value

    def run(self, w, h, f64=True, want_spp=False, want_counters=False, **kw):
        p = self.params(w, h, **kw)
        rows = self.local_rows(p)
        out = np.zeros((rows, w, 3)); spp = np.zeros((rows, w), np.int32)
        cnt = (C.c_int64 * 8)()
        self._ck(self.E.emul_render(self.h, C.byref(p), gi._p(out), gi._p(spp, _ip), cnt if want_counters else None), "render")
        self._last_counters = np.array(list(cnt), np.int64)
        res = out if f64 else out.astype(np.float32)
        return (res, spp) if want_spp else res

    def counters(self):
        return self._last_counters

    def set_wide_nodes(self, on):
        """Returns whether the wide-record walk is in use afterwards (False: the tree is not made of exact octants)."""
        rc = self.E.emul_set_wide(self.h, 1 if on else 0)
        self.photon_planes = bool(rc & 2)
        return bool(rc & 1)

    def set_content_culling(self, on):
        return bool(self.E.emul_set_cull(self.h, 1 if on else 0))

    def trace(self, rays):
        rays = gi._f64(rays).reshape(-1, 6); n = len(rays)
        hit = np.zeros(n, np.int32); ent = np.zeros(n, np.int32); res = np.zeros((n, 8))
        self.E.emul_trace(self.h, n, gi._p(rays), gi._p(hit, _ip), gi._p(ent, _ip), gi._p(res))
        return hit, ent, res

    def visible(self, q):
        q = gi._f64(q).reshape(-1, 6); vis = np.zeros(len(q), np.int32)
        self.E.emul_visible(self.h, len(q), gi._p(q), gi._p(vis, _ip))
        return vis

    def visible_turns(self, q, light_bound=False):
        """visible() as k_st_shadow walks it (turn by turn); None when the scene has no wide records."""
        q = gi._f64(q).reshape(-1, 6); vis = np.zeros(len(q), np.int32)
        if self.E.emul_visible_turns(self.h, len(q), gi._p(q), gi._p(vis, _ip), 1 if light_bound else 0) != 0:
            return None
        return vis

    def samplePhotons(self, q):
        q = gi._f64(q).reshape(-1, 6); res = np.zeros((len(q), 3)); nc = np.zeros(len(q), np.int32)
        self.E.emul_gather(self.h, len(q), gi._p(q), gi._p(res), gi._p(nc, _ip))
        return res, nc

    def radiance(self, rays, stream, seed=None):
        rays = gi._f64(rays).reshape(-1, 6); stream = np.ascontiguousarray(stream, np.uint32); out = np.zeros((len(rays), 3))
        self.E.emul_radiance(self.h, len(rays), gi._p(rays), gi._p(stream, _up), C.c_uint64(self.seed if seed is None else seed), gi._p(out))
        return out

    def kat(self, what, args):
        a = gi._f64(args)
        a = a.reshape(len(a), -1)
        out = np.zeros((len(a), 3))
        self.E.emul_kat(self.KAT[what], len(a), gi._p(a), a.shape[1], gi._p(out))
        return out

    def leaf_order(self, rays, cap=256):
        rays = gi._f64(rays).reshape(-1, 6)
        leaf = np.zeros((len(rays), cap), np.int32); n = np.zeros(len(rays), np.int32)
        self.E.emul_leaf_order(self.h, len(rays), gi._p(rays), cap, gi._p(leaf, _ip), gi._p(n, _ip))
        assert (n <= cap).all()
        return [leaf[i, :n[i]].copy() for i in range(len(rays))]

    def halton_sample(self, dim, index):
        return np.array([self.E.emul_halton_sample(self.h, int(d), int(i)) for d, i in zip(dim, index)], np.float32)

    def halton_index(self, w, h, sxy):
        return np.array([self.E.emul_halton_index(w, h, int(s), int(x), int(y)) for s, x, y in np.asarray(sxy).reshape(-1, 3)], np.uint32)
