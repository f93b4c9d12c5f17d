"""ctypes binding of the CPU oracle (oracle/libgi_oracle.so) -- test infrastructure only.

Nothing under gi_raytracer_amd/ imports this module; it is the checker used by tests/, smoke() and
bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

RNG_COUNTER, RNG_CHAIN = 0, 1
DEFAULT_SEED = 0x9E3779B97F4A7C15

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_up = C.POINTER(C.c_uint32)


def _ptr(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "libgi_oracle.so")
        src = os.path.join(ORACLE_DIR, "gi_oracle.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["make", "-C", ORACLE_DIR, "libgi_oracle.so"], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.gio_create.restype = C.c_void_p
        L.gio_destroy.argtypes = [C.c_void_p]
        L.gio_halton_index.restype = C.c_uint32
        L.gio_halton_index.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
        L.gio_halton_scale.restype = C.c_float
        L.gio_halton_scale.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
        L.gio_halton_sample.restype = C.c_float
        L.gio_halton_sample.argtypes = [C.c_uint32, C.c_uint32]
        L.gio_fast_pow.restype = C.c_double
        L.gio_fast_pow.argtypes = [C.c_double, C.c_double]
        L.gio_fast_precise_pow.restype = C.c_double
        L.gio_fast_precise_pow.argtypes = [C.c_double, C.c_double]
        L.gio_hemi_cos_n.argtypes = [c_dp, C.c_float, C.c_float, C.c_double, c_dp]
        L.gio_hemi_cos.argtypes = [C.c_float, C.c_float, C.c_double, c_dp]
        L.gio_sample_phong.argtypes = [c_dp, c_dp, C.c_double, C.c_double, C.c_double, c_dp]
        L.gio_sphere_cap.argtypes = [c_dp, C.c_float, C.c_float, C.c_double, C.c_double, c_dp]
        L.gio_unit_vec.argtypes = [C.c_double, C.c_double, c_dp]
        L.gio_refr.argtypes = [c_dp, c_dp, C.c_double, c_dp]
        L.gio_reflect.argtypes = [c_dp, c_dp, c_dp]
        L.gio_tri_box_overlap.argtypes = [c_dp, c_dp, c_dp]
        L.gio_chain_drand.restype = C.c_double
        L.gio_chain_drand.argtypes = [C.POINTER(C.c_uint64)]
        L.gio_counter_rand.restype = C.c_double
        L.gio_counter_rand.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.gio_primary_ray.restype = C.c_uint32
        L.gio_primary_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_dp]
        L.gio_set_scene.argtypes = [C.c_void_p, C.c_int, c_ip, c_dp, c_dp, c_dp, c_ip, C.c_int, c_dp, C.c_int, c_dp, c_dp]
        L.gio_set_camera.argtypes = [C.c_void_p, c_dp]
        L.gio_set_fog.argtypes = [C.c_void_p, C.c_int, c_dp, c_ip, c_dp]
        L.gio_tex_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, c_dp, c_dp]
        L.gio_set_textures.argtypes = [C.c_void_p, C.c_int, c_ip, c_dp, c_ip, C.POINTER(C.c_uint8), C.c_int64]
        L.gio_chain_discard.argtypes = [C.c_void_p, C.c_int64]
        L.gio_build_octree.argtypes = [C.c_void_p]
        L.gio_chain_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.gio_octree_counts.argtypes = [C.c_void_p, c_ip, c_ip]
        L.gio_octree_dump.argtypes = [C.c_void_p, c_dp, c_ip, c_ip, c_ip]
        L.gio_get_lights.argtypes = [C.c_void_p, c_dp]
        L.gio_ent_bbox.argtypes = [C.c_void_p, c_dp]
        L.gio_trace.argtypes = [C.c_void_p, C.c_int, c_dp, c_ip, c_ip, c_dp, c_ip]
        L.gio_leaf_order.argtypes = [C.c_void_p, c_dp, C.c_int, c_ip, c_dp]
        L.gio_visible.argtypes = [C.c_void_p, C.c_int, c_dp, c_ip, c_ip]
        L.gio_pixel8.argtypes = [C.c_int, c_dp, C.POINTER(C.c_uint8)]
        L.gio_set_photons.argtypes = [C.c_void_p, C.c_int, c_dp]
        L.gio_photon_count.argtypes = [C.c_void_p]
        L.gio_get_photons.argtypes = [C.c_void_p, c_dp]
        L.gio_emit_photons.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_int64)]
        L.gio_build_photon_map.argtypes = [C.c_void_p]
        L.gio_pmap_counts.argtypes = [C.c_void_p, c_ip, c_ip]
        L.gio_pmap_dump.argtypes = [C.c_void_p, c_dp, c_ip, c_ip, c_ip]
        L.gio_gather.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_ip]
        L.gio_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                 C.c_uint64, C.c_int, C.c_int, c_dp, C.POINTER(C.c_uint8), c_ip, C.POINTER(C.c_int64)]
        L.gio_render_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_ip, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_int, c_dp,
                                      C.POINTER(C.c_int64)]
        L.gio_radiance.argtypes = [C.c_void_p, C.c_int, c_dp, c_up, C.c_uint64, c_dp]
        _LIB = L
    return _LIB


def unique_materials(tri_mat):
    """Per-entity material rows [n][9] -> (mat_idx [n] int32, mats [m][9]) keeping first-seen order."""
    mats, idx, seen = [], np.zeros(len(tri_mat), np.int32), {}
    for i, row in enumerate(tri_mat):
        key = row.tobytes()
        if key not in seen:
            seen[key] = len(mats)
            mats.append(row)
        idx[i] = seen[key]
    return idx, np.ascontiguousarray(np.array(mats, np.float64).reshape(len(mats), -1))


class Oracle:
    """One scene in the oracle."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.gio_create())
        self.n_ent = 0
        self.n_light = 0

    def __del__(self):
        try:
            self.L.gio_destroy(self.h)
        except Exception:
            pass

    def set_scene(self, pos, nrm, uv, mat_idx, mats, lights, ambient=(0, 0, 0), kind=None):
        pos = np.ascontiguousarray(pos, np.float64)
        nrm = np.ascontiguousarray(nrm, np.float64)
        uv = np.ascontiguousarray(uv, np.float64)
        mat_idx = np.ascontiguousarray(mat_idx, np.int32)
        mats = np.ascontiguousarray(mats, np.float64)
        lights = np.ascontiguousarray(lights, np.float64).reshape(-1, 7)
        amb = np.ascontiguousarray(ambient, np.float64)
        kind_a = np.ascontiguousarray(kind, np.int32) if kind is not None else None
        self.n_ent, self.n_light = len(mat_idx), len(lights)
        r = self.L.gio_set_scene(self.h, self.n_ent, _ptr(kind_a, c_ip), _ptr(pos, c_dp), _ptr(nrm, c_dp), _ptr(uv, c_dp),
                                 _ptr(mat_idx, c_ip), len(mats), _ptr(mats, c_dp), self.n_light, _ptr(lights, c_dp), _ptr(amb, c_dp))
        assert r == 0
        return self

    def set_fog(self, params, grid_off, grid):
        params = np.ascontiguousarray(params, np.float64).reshape(-1, 12)
        grid_off = np.ascontiguousarray(grid_off, np.int32); grid = np.ascontiguousarray(grid, np.float64)
        assert self.L.gio_set_fog(self.h, len(params), _ptr(params, c_dp), _ptr(grid_off, c_ip), _ptr(grid, c_dp)) == 0
        return self

    def set_textures(self, kind, param, mat_tex, pixels):
        kind = np.ascontiguousarray(kind, np.int32); param = np.ascontiguousarray(param, np.float64)
        mat_tex = np.ascontiguousarray(mat_tex, np.int32); pixels = np.ascontiguousarray(pixels, np.uint8)
        assert self.L.gio_set_textures(self.h, len(kind), _ptr(kind, c_ip), _ptr(param, c_dp), _ptr(mat_tex, c_ip),
                                       pixels.ctypes.data_as(C.POINTER(C.c_uint8)), len(pixels)) == 0
        return self

    def tex_eval(self, tex, uv):
        uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
        out = np.zeros((len(uv), 4))
        assert self.L.gio_tex_eval(self.h, int(tex), len(uv), _ptr(uv, c_dp), _ptr(out, c_dp)) == 0
        return out

    def chain_discard(self, n):
        self.L.gio_chain_discard(self.h, int(n))
        return self

    def set_camera(self, pos, up, forward, sensor_diag=16.8, focal_dist=9.6):
        cam = np.array(list(pos) + list(up) + list(forward) + [sensor_diag, focal_dist], np.float64)
        self.L.gio_set_camera(self.h, _ptr(cam, c_dp))
        return self

    @classmethod
    def from_fixture(cls, fx):
        o = cls()
        textured = "tex_kind" in fx and (np.asarray(fx["tex_kind"]) != 0).any()
        rows = np.concatenate([fx["tri_mat"], fx["tri_tex"].astype(np.float64)], 1) if textured else fx["tri_mat"]
        mat_idx, mats = unique_materials(rows)
        o.set_scene(fx["tri_pos"], fx["tri_nrm"], fx["tri_uv"], mat_idx, np.ascontiguousarray(mats[:, :9]), fx["lights"][:, :7] if len(fx["lights"]) else np.zeros((0, 7)),
                    fx["settings"][0:3], kind=fx["ent_kind"])
        if textured:
            o.set_textures(fx["tex_kind"], fx["tex_param"], mats[:, 9:11].astype(np.int32), fx["tex_pixels"])
        s = fx["settings"]
        o.set_camera(s[8:11], s[11:14], s[14:17], s[20], s[21])
        if "fog" in fx and len(fx["fog"]):
            o.set_fog(fx["fog"], fx["fog_grid_off"], fx["fog_grid"])
        return o

    def chain_seed(self, seed):
        self.L.gio_chain_seed(self.h, C.c_uint64(seed))
        return self

    def build_octree(self):
        assert self.L.gio_build_octree(self.h) == 0
        return self

    def octree(self):
        nn, nr = C.c_int32(), C.c_int32()
        self.L.gio_octree_counts(self.h, C.byref(nn), C.byref(nr))
        bbox = np.zeros((nn.value, 6)); child = np.zeros((nn.value, 8), np.int32)
        off = np.zeros(nn.value + 1, np.int32); idx = np.zeros(max(nr.value, 1), np.int32)
        self.L.gio_octree_dump(self.h, _ptr(bbox, c_dp), _ptr(child, c_ip), _ptr(off, c_ip), _ptr(idx, c_ip))
        return bbox, child, off, idx[:nr.value]

    def lights_dir_angle(self):
        out = np.zeros((self.n_light, 4))
        self.L.gio_get_lights(self.h, _ptr(out, c_dp))
        return out

    def ent_bbox(self):
        out = np.zeros((self.n_ent, 6))
        self.L.gio_ent_bbox(self.h, _ptr(out, c_dp))
        return out

    def trace(self, rays):
        rays = np.ascontiguousarray(rays, np.float64)
        n = len(rays)
        hit = np.zeros(n, np.int32); ent = np.zeros(n, np.int32); res = np.zeros((n, 8)); nl = np.zeros(n, np.int32)
        assert self.L.gio_trace(self.h, n, _ptr(rays, c_dp), _ptr(hit, c_ip), _ptr(ent, c_ip), _ptr(res, c_dp), _ptr(nl, c_ip)) == 0
        return hit, ent, res, nl

    def leaf_order(self, ray, cap=1024):
        ray = np.ascontiguousarray(ray, np.float64)
        node = np.zeros(cap, np.int32); t0 = np.zeros(cap)
        n = self.L.gio_leaf_order(self.h, _ptr(ray, c_dp), cap, _ptr(node, c_ip), _ptr(t0, c_dp))
        return node[:n], t0[:n]

    def visible(self, q):
        q = np.ascontiguousarray(q, np.float64)
        n = len(q)
        vis = np.zeros(n, np.int32); nc = np.zeros(n, np.int32)
        assert self.L.gio_visible(self.h, n, _ptr(q, c_dp), _ptr(vis, c_ip), _ptr(nc, c_ip)) == 0
        return vis, nc

    def set_photons(self, ph):
        ph = np.ascontiguousarray(ph, np.float64).reshape(-1, 9)
        self.L.gio_set_photons(self.h, len(ph), _ptr(ph, c_dp))
        return self

    def get_photons(self):
        n = self.L.gio_photon_count(self.h)
        out = np.zeros((n, 9))
        if n:
            self.L.gio_get_photons(self.h, _ptr(out, c_dp))
        return out

    def emit_photons(self, count, max_depth=5, rng_mode=RNG_COUNTER, seed=DEFAULT_SEED):
        tries = C.c_int64()
        n = self.L.gio_emit_photons(self.h, count, max_depth, rng_mode, C.c_uint64(seed), C.byref(tries))
        assert n >= 0
        return n, tries.value

    def build_photon_map(self):
        assert self.L.gio_build_photon_map(self.h) == 0
        return self

    def pmap(self):
        nn, nr = C.c_int32(), C.c_int32()
        self.L.gio_pmap_counts(self.h, C.byref(nn), C.byref(nr))
        bbox = np.zeros((nn.value, 6)); fc = np.zeros(nn.value, np.int32)
        off = np.zeros(nn.value + 1, np.int32); idx = np.zeros(max(nr.value, 1), np.int32)
        self.L.gio_pmap_dump(self.h, _ptr(bbox, c_dp), _ptr(fc, c_ip), _ptr(off, c_ip), _ptr(idx, c_ip))
        return bbox, fc, off, idx[:nr.value]

    def gather(self, q):
        q = np.ascontiguousarray(q, np.float64)
        n = len(q)
        res = np.zeros((n, 3)); nc = np.zeros(n, np.int32)
        assert self.L.gio_gather(self.h, n, _ptr(q, c_dp), _ptr(res, c_dp), _ptr(nc, c_ip)) == 0
        return res, nc

    def render(self, w, h, min_samples, max_samples=None, noise_thresh=0.0015, rng_mode=RNG_COUNTER, seed=DEFAULT_SEED,
               chain_predraws=0, n_threads=0, y0=0, y1=None, want_u8=False, want_counters=False):
        max_samples = min_samples if max_samples is None else max_samples
        y1 = h if y1 is None else y1
        lin = np.zeros((h, w, 3)); spp = np.zeros((h, w), np.int32)
        u8 = np.zeros((h, w, 3), np.uint8) if want_u8 else None
        cnt = np.zeros(9, np.int64) if want_counters else None
        r = self.L.gio_render(self.h, w, h, y0, y1, min_samples, max_samples, noise_thresh, rng_mode, C.c_uint64(seed), chain_predraws,
                              n_threads, _ptr(lin, c_dp), _ptr(u8, C.POINTER(C.c_uint8)), _ptr(spp, c_ip),
                              _ptr(cnt, C.POINTER(C.c_int64)))
        assert r == 0
        out = {"lin": lin, "spp": spp}
        if want_u8:
            out["u8"] = u8
        if want_counters:
            out["counters"] = cnt
        return out

    def render_rows(self, w, h, rows, spp, seed=DEFAULT_SEED, n_threads=0):
        """Full-width rows `rows` of a w x h frame at fixed spp; returns (lin [h][w][3] with only those rows filled, counters[8])."""
        rows = np.ascontiguousarray(rows, np.int32)
        lin = np.zeros((h, w, 3)); cnt = np.zeros(9, np.int64)
        r = self.L.gio_render_rows(self.h, w, h, len(rows), _ptr(rows, c_ip), spp, spp, 0.0, C.c_uint64(seed), n_threads, _ptr(lin, c_dp),
                                   _ptr(cnt, C.POINTER(C.c_int64)))
        assert r == 0
        return lin, cnt

    def radiance(self, rays, stream, seed=DEFAULT_SEED):
        rays = np.ascontiguousarray(rays, np.float64)
        stream = np.ascontiguousarray(stream, np.uint32)
        out = np.zeros((len(rays), 3))
        assert self.L.gio_radiance(self.h, len(rays), _ptr(rays, c_dp), _ptr(stream, c_up), C.c_uint64(seed), _ptr(out, c_dp)) == 0
        return out

    def primary_ray(self, w, h, s, x, y):
        ray = np.zeros(6)
        idx = self.L.gio_primary_ray(self.h, w, h, s, x, y, _ptr(ray, c_dp))
        return idx, ray
