"""gi_raytracer_amd -- Python host mirror of the MI355X render hot path of GI_Raytracer.

Thin ctypes binding over the C ABI (include/gi_hip.h, gi_raytracer_amd/csrc/gi_host.h) of libgi_raytracer_hip.so.
The names mirror the reference's C++ surface for this path: `Scene` plays Octree (+ the loaders that fill it),
`RayTracer` plays RayTracer (setScene / run / trace / visible / samplePhotons / tracePhotons).  There is no CPU
implementation behind these classes: creating a RayTracer without the HIP library or without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library, in-tree.  An experimental build of the same library (tools/build_exp.sh -> exp/*.so) is loaded instead only when
# BOTH GI_EXPERIMENTAL=1 and GI_LIB_PATH are set (the tuning scripts under tools/ set them); bench.py prints the path it resolved.
LIB_PATH = os.path.join(_HERE, "libgi_raytracer_hip.so")
if os.environ.get("GI_EXPERIMENTAL") == "1" and os.environ.get("GI_LIB_PATH"):
    LIB_PATH = os.environ["GI_LIB_PATH"]
DEFAULT_SEED = 0x9E3779B97F4A7C15

GI_OK, GI_E_NO_DEVICE, GI_E_INVALID, GI_E_HIP, GI_E_STATE, GI_E_CANCELLED = 0, -1, -2, -3, -4, -5

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint32)


class GiError(RuntimeError):
    pass


class SceneDesc(C.Structure):
    _fields_ = [("n_tri", C.c_int32), ("tri_pos", _dp), ("tri_nrm", _dp), ("tri_uv", _dp), ("tri_mat", _ip),
                ("n_mat", C.c_int32), ("mats", _dp), ("n_light", C.c_int32), ("lights", _dp), ("ambient", C.c_double * 3),
                ("n_node", C.c_int32), ("node_bbox", _dp), ("node_child", _ip), ("node_ent_off", _ip), ("node_ent_idx", _ip),
                ("ent_kind", _ip), ("n_fog", C.c_int32), ("fog", _dp), ("fog_grid_off", _ip), ("fog_grid", _dp),
                ("n_tex", C.c_int32), ("tex_kind", _ip), ("tex_param", _dp), ("mat_tex", _ip), ("tex_pixels", C.POINTER(C.c_uint8)),
                ("n_tex_pixel_bytes", C.c_int64)]


class PhotonMapDesc(C.Structure):
    _fields_ = [("n_photon", C.c_int32), ("photons", _dp), ("n_node", C.c_int32), ("node_bbox", _dp), ("node_child", _ip),
                ("node_off", _ip), ("node_idx", _ip)]


class RenderParams(C.Structure):
    _fields_ = [("cam_pos", C.c_double * 3), ("cam_up", C.c_double * 3), ("cam_forward", C.c_double * 3),
                ("sensor_diag", C.c_double), ("focal_dist", C.c_double), ("width", C.c_int32), ("height", C.c_int32),
                ("stripe_h", C.c_int32), ("stripe_rank", C.c_int32), ("stripe_world", C.c_int32),
                ("min_samples", C.c_int32), ("max_samples", C.c_int32), ("noise_thresh", C.c_double), ("seed", C.c_uint64)]


class Settings(C.Structure):
    _fields_ = [("photons", C.c_int32), ("photon_depth", C.c_int32), ("min_samples", C.c_int32), ("max_samples", C.c_int32),
                ("noise_thresh", C.c_double), ("ambient", C.c_double * 3), ("cam_pos", C.c_double * 3), ("cam_up", C.c_double * 3),
                ("cam_forward", C.c_double * 3), ("sensor_diag", C.c_double), ("focal_dist", C.c_double)]


_LIB = None

# every symbol include/gi_hip.h and csrc/gi_host.h declare (tests check the library exports all of them)
ABI_SYMBOLS = [
    "gi_create", "gi_destroy", "gi_last_error", "gi_set_stream", "gi_upload_scene", "gi_upload_photons", "gi_local_rows",
    "gi_render_device", "gi_render_host", "gi_set_render_mode", "gi_set_wide_nodes", "gi_set_content_culling", "gi_set_entity_boxes", "gi_set_pool_slots", "gi_last_render_ms", "gi_last_stage_ms", "gi_last_kernel_ms", "gi_set_counters", "gi_get_counters", "gi_get_stream_counters", "gi_trace", "gi_visible",
    "gi_gather", "gi_radiance", "gi_emit_photons", "gi_halton_sample", "gi_halton_index", "gi_debug_leaf_order", "gi_debug_sort_pairs", "gi_debug_find_leaves", "gi_kat", "gi_visible_rays", "gi_build_photon_map", "gi_trace_photons", "gi_debug_photon_tables", "gi_clear_photons", "gi_group_clear_photons",
    "gi_device_count", "gi_group_create", "gi_group_destroy", "gi_group_size", "gi_group_ctx", "gi_group_last_error", "gi_group_upload_scene", "gi_group_upload_photons", "gi_group_render_host", "gi_group_render_device",
    "gih_scene_create", "gih_scene_destroy", "gih_last_error", "gih_load_scn", "gih_add_material", "gih_add_triangles",
    "gih_add_texture", "gih_add_material_tex", "gih_load_png", "gih_free",
    "gih_add_light", "gih_add_sphere", "gih_add_height_fog", "gih_set_ambient", "gih_get_settings", "gih_set_camera", "gih_build_octree", "gih_get_scene_desc",
    "gih_counts", "gih_build_photon_map", "gih_get_photon_desc", "gih_to_rgb8", "gih_entity_bbox", "gih_entity_overlaps_box", "gih_box_mesh", "gih_fog_grid", "gih_build_photon_map_in_box", "gih_add_height_fog_grid", "gih_load_obj",
]


def lib():
    """Load libgi_raytracer_hip.so (built by __graft_entry__.build() / csrc/Makefile).  Raises if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise GiError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.gi_create.argtypes = [C.POINTER(vp), C.c_int]
    L.gi_destroy.argtypes = [vp]
    L.gi_last_error.argtypes = [vp]
    L.gi_last_error.restype = C.c_char_p
    L.gi_set_stream.argtypes = [vp, vp]
    L.gi_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
    L.gi_upload_photons.argtypes = [vp, C.POINTER(PhotonMapDesc)]
    L.gi_local_rows.argtypes = [C.POINTER(RenderParams)]
    L.gi_render_device.argtypes = [vp, C.POINTER(RenderParams), vp, C.c_int, vp, vp]
    L.gi_render_host.argtypes = [vp, C.POINTER(RenderParams), vp, C.c_int, vp, vp]
    L.gi_set_render_mode.argtypes = [vp, C.c_int]
    L.gi_set_wide_nodes.argtypes = [vp, C.c_int]
    L.gi_set_content_culling.argtypes = [vp, C.c_int]
    L.gi_set_entity_boxes.argtypes = [vp, C.c_int]
    L.gi_set_pool_slots.argtypes = [vp, C.c_int64]
    L.gi_last_render_ms.argtypes = [vp, C.POINTER(C.c_float), _ip]
    L.gi_last_stage_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.gi_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.gi_set_counters.argtypes = [vp, C.c_int]
    L.gi_get_counters.argtypes = [vp, C.POINTER(C.c_int64)]
    L.gi_get_stream_counters.argtypes = [vp, C.POINTER(C.c_int64)]
    L.gi_trace.argtypes = [vp, C.c_int32, _dp, _ip, _ip, _dp]
    L.gi_visible.argtypes = [vp, C.c_int32, _dp, _ip]
    L.gi_gather.argtypes = [vp, C.c_int32, _dp, _dp, _ip]
    L.gi_radiance.argtypes = [vp, C.c_int32, _dp, _up, C.c_uint64, _dp]
    L.gi_emit_photons.argtypes = [vp, C.c_int32, C.c_int32, C.c_uint64, _dp, C.c_int32, C.POINTER(C.c_int64)]
    L.gi_halton_sample.argtypes = [vp, C.c_int32, _up, _up, C.POINTER(C.c_float)]
    L.gi_halton_index.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, _up, _up]
    L.gi_debug_leaf_order.argtypes = [vp, C.c_int32, _dp, C.c_int32, _ip, _ip]
    L.gi_debug_sort_pairs.argtypes = [vp, C.c_int32, _up, _up, C.c_int32, C.c_int32, _up, _up]
    L.gi_debug_find_leaves.argtypes = [vp, C.c_int32, _dp, _ip, _ip]
    L.gi_kat.argtypes = [vp, C.c_int32, C.c_int32, _dp, C.c_int32, _dp]
    L.gi_clear_photons.argtypes = [vp]
    L.gi_group_clear_photons.argtypes = [vp]
    L.gi_build_photon_map.argtypes = [vp, C.c_int32, _dp, _dp]
    L.gi_trace_photons.argtypes = [vp, C.c_int32, C.c_int32, C.c_uint64, _dp, C.POINTER(C.c_int64)]
    L.gi_debug_photon_tables.argtypes = [vp, _ip, _ip, _ip, vp, _ip, _dp, _dp]
    L.gi_group_create.argtypes = [C.POINTER(vp), C.c_int32, _ip]
    L.gi_group_destroy.argtypes = [vp]
    L.gi_group_size.argtypes = [vp]
    L.gi_group_ctx.argtypes = [vp, C.c_int32]
    L.gi_group_ctx.restype = vp
    L.gi_group_last_error.argtypes = [vp]
    L.gi_group_last_error.restype = C.c_char_p
    L.gi_group_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
    L.gi_group_upload_photons.argtypes = [vp, C.POINTER(PhotonMapDesc)]
    L.gi_group_render_host.argtypes = [vp, C.POINTER(RenderParams), C.c_int32, C.c_int32, C.c_int32, vp, C.c_int, vp, vp]
    L.gi_group_render_device.argtypes = [vp, C.POINTER(RenderParams), C.c_int32, vp, C.c_int, vp]
    L.gih_scene_create.restype = vp
    L.gih_scene_destroy.argtypes = [vp]
    L.gih_last_error.argtypes = [vp]
    L.gih_last_error.restype = C.c_char_p
    L.gih_load_scn.argtypes = [vp, C.c_char_p]
    L.gih_add_material.argtypes = [vp, _dp]
    L.gih_add_texture.argtypes = [vp, C.c_int32, _dp, C.POINTER(C.c_uint8), C.c_int64]
    L.gih_add_material_tex.argtypes = [vp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]
    L.gih_load_png.argtypes = [C.c_char_p, _ip, _ip, _ip, C.POINTER(C.POINTER(C.c_uint8)), C.c_char_p, C.c_int32]
    L.gih_free.argtypes = [vp]
    L.gih_add_triangles.argtypes = [vp, C.c_int32, _dp, _dp, _dp, _ip]
    L.gih_add_light.argtypes = [vp, _dp, _dp, C.c_double]
    L.gih_add_sphere.argtypes = [vp, _dp, C.c_double, C.c_int32]
    L.gih_add_height_fog.argtypes = [vp, _dp, _dp, C.c_int32, C.c_uint64]
    L.gih_set_ambient.argtypes = [vp, _dp]
    L.gih_get_settings.argtypes = [vp, C.POINTER(Settings)]
    L.gih_set_camera.argtypes = [vp, _dp, _dp]
    L.gih_build_octree.argtypes = [vp]
    L.gih_get_scene_desc.argtypes = [vp, C.POINTER(SceneDesc)]
    L.gih_counts.argtypes = [vp, _ip, _ip, _ip, _ip, _ip]
    L.gih_build_photon_map.argtypes = [vp, C.c_int32, _dp]
    L.gih_get_photon_desc.argtypes = [vp, C.POINTER(PhotonMapDesc)]
    L.gih_to_rgb8.argtypes = [vp, C.c_int32, C.c_int64, C.POINTER(C.c_uint8)]
    _LIB = L
    return L


def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def _p(a, ty=_dp):
    return a.ctypes.data_as(ty) if a is not None else None


def _np_from(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape).copy()


class Scene:
    """Host-side scene = the reference's Octree plus what its loaders write into RayTracer (include/octree.h:39-64,
    include/sceneLoader.cpp).  Build order as in the reference: push entities / lights, `rebuild()`, then render."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.gih_scene_create())

    def __del__(self):
        try:
            self.L.gih_scene_destroy(self.h)
        except Exception:
            pass

    def _err(self):
        return self.L.gih_last_error(self.h).decode()

    @classmethod
    def load(cls, scn_path):
        """loadScene(scene, raytracer, path) (include/sceneLoader.cpp:12)."""
        s = cls()
        rc = s.L.gih_load_scn(s.h, os.fsencode(scn_path))
        if rc != 0:
            raise GiError(f"load_scn({scn_path}): {s._err()}")
        return s

    def add_material(self, roughness, opacity, ior, diffuse, emissive=(0, 0, 0)):
        m = _f64([roughness, opacity, ior, *diffuse, *emissive])
        return self.L.gih_add_material(self.h, _p(m))

    def add_color_texture(self, rgb):
        """new texture(col) (include/material.h:10-30); returns the texture index."""
        return self._tex(0, [*rgb, 0, 0, 0, 0, 0], None)

    def add_checkerboard(self, a, b, tiles):
        """new checkerboard(tiles, a, b) (include/material.h:32-50)."""
        return self._tex(1, [*a, *b, tiles, 0], None)

    def add_image_texture(self, rgba, tile=(1, 1), has_alpha=True):
        """new imageTexture(file, tile) (include/material.h:52-81) from decoded pixels [h][w][4] uint8, rows top to bottom."""
        rgba = np.ascontiguousarray(rgba, np.uint8)
        h, w = rgba.shape[:2]
        return self._tex(2, [tile[0], tile[1], w, h, 1 if has_alpha else 0, 0, 0, 0], rgba.reshape(-1))

    def _tex(self, kind, params, pixels):
        par = _f64(params)
        rc = self.L.gih_add_texture(self.h, kind, _p(par), pixels.ctypes.data_as(C.POINTER(C.c_uint8)) if pixels is not None else None,
                                    len(pixels) if pixels is not None else 0)
        if rc < 0:
            raise GiError(self._err())
        return rc

    def add_material_tex(self, dif_tex, em_tex, roughness, opacity, ior=1.0):
        """new Material(tex[dif], tex[em], roughness, opacity, IOR) (include/material.h:84-100); returns the material index."""
        rc = self.L.gih_add_material_tex(self.h, int(dif_tex), int(em_tex), float(roughness), float(opacity), float(ior))
        if rc < 0:
            raise GiError(self._err())
        return rc

    def add_triangles(self, pos, nrm=None, uv=None, mat_idx=None):
        pos = _f64(pos).reshape(-1, 3, 3)
        n = len(pos)
        nrm = _f64(nrm).reshape(n, 3, 3) if nrm is not None else None
        uv = _f64(uv).reshape(n, 3, 2) if uv is not None else None
        mi = np.ascontiguousarray(mat_idx if mat_idx is not None else np.zeros(n), np.int32)
        rc = self.L.gih_add_triangles(self.h, n, _p(pos), _p(nrm), _p(uv), _p(mi, _ip))
        if rc != 0:
            raise GiError(self._err())

    def add_sphere(self, centre, radius, mat_idx):
        """o->push_back(new sphere(pos, rad, mat)) (include/sceneLoader.cpp:122-130)."""
        if self.L.gih_add_sphere(self.h, _p(_f64(centre)), float(radius), int(mat_idx)) != 0:
            raise GiError(self._err())

    def add_height_fog(self, pos, size, col, density, scatter, noise_scale, grid=None, seed=DEFAULT_SEED):
        """o->push_back(new HeightFog(...)) (include/sceneLoader.cpp:150-158).  grid=None fills the noise grid from the counter RNG
        (the reference fills it with time-seeded drand())."""
        par = _f64([*pos, *size, *col, density, scatter, noise_scale])
        g = _f64(grid) if grid is not None else None
        if self.L.gih_add_height_fog(self.h, _p(par), _p(g), 0 if g is None else len(g), C.c_uint64(seed)) != 0:
            raise GiError(self._err())

    def add_light(self, pos, col, rad):
        self.L.gih_add_light(self.h, _p(_f64(pos)), _p(_f64(col)), float(rad))

    def set_ambient(self, rgb):
        self.L.gih_set_ambient(self.h, _p(_f64(rgb)))

    def set_camera(self, pos, look_at):
        self.L.gih_set_camera(self.h, _p(_f64(pos)), _p(_f64(look_at)))

    @property
    def settings(self):
        st = Settings()
        self.L.gih_get_settings(self.h, C.byref(st))
        return st

    def rebuild(self):
        """Octree::rebuild (include/octree.cpp:53-119)."""
        rc = self.L.gih_build_octree(self.h)
        if rc != 0:
            raise GiError(f"build_octree: {self._err()}")
        return self

    def desc(self):
        d = SceneDesc()
        rc = self.L.gih_get_scene_desc(self.h, C.byref(d))
        if rc != 0:
            raise GiError("scene octree not built: call rebuild() first")
        return d

    def tables(self):
        """Numpy copies of the flattened tables (tests)."""
        d = self.desc()
        nref = _np_from(d.node_ent_off, (d.n_node + 1,), np.int32)[-1]
        return {
            "tri_pos": _np_from(d.tri_pos, (d.n_tri, 3, 3), np.float64), "tri_nrm": _np_from(d.tri_nrm, (d.n_tri, 3, 3), np.float64),
            "tri_uv": _np_from(d.tri_uv, (d.n_tri, 3, 2), np.float64), "tri_mat": _np_from(d.tri_mat, (d.n_tri,), np.int32),
            "mats": _np_from(d.mats, (d.n_mat, 9), np.float64), "lights": _np_from(d.lights, (d.n_light, 11), np.float64),
            "ambient": np.array(list(d.ambient)), "node_bbox": _np_from(d.node_bbox, (d.n_node, 6), np.float64),
            "node_child": _np_from(d.node_child, (d.n_node, 8), np.int32), "node_ent_off": _np_from(d.node_ent_off, (d.n_node + 1,), np.int32),
            "node_ent_idx": _np_from(d.node_ent_idx, (int(nref),), np.int32),
            "ent_kind": _np_from(d.ent_kind, (d.n_tri,), np.int32) if d.ent_kind else np.zeros(d.n_tri, np.int32),
            "fog": _np_from(d.fog, (d.n_fog, 12), np.float64) if d.n_fog else np.zeros((0, 12)),
            "fog_grid_off": _np_from(d.fog_grid_off, (d.n_fog + 1,), np.int32) if d.n_fog else np.zeros(1, np.int32),
            "fog_grid": _np_from(d.fog_grid, (int(_np_from(d.fog_grid_off, (d.n_fog + 1,), np.int32)[-1]),), np.float64) if d.n_fog else np.zeros(0),
            "tex_kind": _np_from(d.tex_kind, (d.n_tex,), np.int32) if d.n_tex else np.zeros(0, np.int32),
            "tex_param": _np_from(d.tex_param, (d.n_tex, 8), np.float64) if d.n_tex else np.zeros((0, 8)),
            "mat_tex": _np_from(d.mat_tex, (d.n_mat, 2), np.int32) if d.n_tex else np.zeros((0, 2), np.int32),
            "tex_pixels": _np_from(d.tex_pixels, (int(d.n_tex_pixel_bytes),), np.uint8) if d.n_tex and d.n_tex_pixel_bytes else np.zeros(0, np.uint8),
        }

    def build_photon_map(self, photons):
        """PhotonMap::push_back x n + rebuild (include/photonMap.cpp:24-47)."""
        ph = _f64(photons).reshape(-1, 9)
        rc = self.L.gih_build_photon_map(self.h, len(ph), _p(ph))
        if rc != 0:
            raise GiError(f"build_photon_map: {self._err()}")
        return self

    def photon_desc(self):
        d = PhotonMapDesc()
        self.L.gih_get_photon_desc(self.h, C.byref(d))
        return d

    def photon_tables(self):
        d = self.photon_desc()
        nref = _np_from(d.node_off, (d.n_node + 1,), np.int32)[-1] if d.n_node else 0
        return {"photons": _np_from(d.photons, (d.n_photon, 9), np.float64), "node_bbox": _np_from(d.node_bbox, (d.n_node, 6), np.float64),
                "node_child": _np_from(d.node_child, (d.n_node, 8), np.int32), "node_off": _np_from(d.node_off, (d.n_node + 1,), np.int32) if d.n_node else np.zeros(1, np.int32),
                "node_idx": _np_from(d.node_idx, (int(nref),), np.int32)}


def save_pfm(path, lin):
    """Linear radiance [h][w][3] as a little-endian PFM (rows bottom to top)."""
    a = np.ascontiguousarray(lin, np.float32)
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (a.shape[1], a.shape[0]))
        f.write(a[::-1].astype("<f4").tobytes())


def to_rgb8(lin):
    """The reference's display transform: gamma 2.2, clamp to [0, 1], truncating (int)(255 c) (include/raytracer.h:150-157, image.h:15)."""
    a = np.asarray(lin)
    a = np.ascontiguousarray(a, np.float32 if a.dtype == np.float32 else np.float64)
    out = np.zeros(a.shape, np.uint8)
    if lib().gih_to_rgb8(a.ctypes.data_as(C.c_void_p), 1 if a.dtype == np.float64 else 0, a.size, out.ctypes.data_as(C.POINTER(C.c_uint8))) != 0:
        raise GiError("to_rgb8: invalid arguments")
    return out


def save_ppm(path, lin):
    a = to_rgb8(lin)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(a.tobytes())


class RayTracerGroup:
    """Several devices driven from one process through the C ABI's gi_group_* entries (include/gi_hip.h): what the C++ RayTracer::run uses when
    more than one GPU is visible.  `devices` may repeat an ordinal (several contexts on one device)."""

    def __init__(self, devices):
        self.L = lib()
        h = C.c_void_p()
        dv = np.ascontiguousarray(devices, np.int32)
        rc = self.L.gi_group_create(C.byref(h), len(dv), _p(dv, _ip))
        if rc != GI_OK:
            raise GiError(f"gi_group_create failed ({rc}): no usable HIP device -- this path has no CPU fallback")
        self.h = h
        self.n = self.L.gi_group_size(h)

    def __del__(self):
        try:
            self.L.gi_group_destroy(self.h)
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise GiError(f"{what}: {self.L.gi_group_last_error(self.h).decode()} ({rc})")

    def setScene(self, scene, photons=None):
        d = scene.desc()
        self._check(self.L.gi_group_upload_scene(self.h, C.byref(d)), "group upload_scene")
        self.L.gi_group_clear_photons(self.h)
        if photons is not None:
            scene.build_photon_map(photons)
            pd = scene.photon_desc()
            self._check(self.L.gi_group_upload_photons(self.h, C.byref(pd)), "group upload_photons")
        return self

    def run(self, params, stripe_h=16, first_stripe=0, n_stripes=None, f64=True, frame=None):
        """gi_group_render_host: the stripes of the window into a whole-frame host buffer (rows outside the window keep what `frame` held)."""
        w, h = params.width, params.height
        total = (h + stripe_h - 1) // stripe_h
        n_stripes = total if n_stripes is None else n_stripes
        out = np.zeros((h, w, 3), np.float64 if f64 else np.float32) if frame is None else frame
        self._check(self.L.gi_group_render_host(self.h, C.byref(params), stripe_h, first_stripe, n_stripes, out.ctypes.data_as(C.c_void_p), 1 if f64 else 0, None, None), "group render_host")
        return out

    def run_device(self, params, out_ptr, stripe_h=16, f64=False):
        """gi_group_render_device: the whole frame gathered on the first context's device (raw device pointer)."""
        self._check(self.L.gi_group_render_device(self.h, C.byref(params), stripe_h, C.c_void_p(out_ptr), 1 if f64 else 0, None), "group render_device")


class RayTracer:
    """Device context = the reference's RayTracer for this path (include/raytracer.h:23-735)."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.gi_create(C.byref(h), int(device))
        if rc != GI_OK:
            raise GiError(f"gi_create failed ({rc}): no usable HIP device -- this path has no CPU fallback")
        self.h = h
        self.scene = None
        st = Scene().settings  # reference defaults (include/util.h:24-29, main.cpp:28)
        self.photons, self.photon_depth = st.photons, st.photon_depth
        self.min_samples, self.max_samples, self.noise_thresh = st.min_samples, st.max_samples, st.noise_thresh
        self.cam_pos, self.cam_up, self.cam_forward = list(st.cam_pos), list(st.cam_up), list(st.cam_forward)
        self.sensor_diag, self.focal_dist = st.sensor_diag, st.focal_dist
        self.seed = DEFAULT_SEED

    def __del__(self):
        try:
            self.L.gi_destroy(self.h)
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise GiError(f"{what}: {self.L.gi_last_error(self.h).decode()} ({rc})")
        return rc

    def set_stream(self, stream_ptr):
        self._check(self.L.gi_set_stream(self.h, C.c_void_p(stream_ptr)), "set_stream")

    def setScene(self, scene, adopt_settings=True):
        """RayTracer::setScene (include/raytracer.h:35-39) + upload of the flattened tables."""
        d = scene.desc()
        self._check(self.L.gi_upload_scene(self.h, C.byref(d)), "upload_scene")
        self._check(self.L.gi_clear_photons(self.h), "clear_photons")      # a fresh PhotonMap, as RayTracer::setScene allocates
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
        self._check(self.L.gi_upload_photons(self.h, C.byref(d)), "upload_photons")

    def clear_photons(self):
        """Drop the photon map (an empty PhotonMap: samplePhotons returns 0, include/raytracer.h:536-540)."""
        self._check(self.L.gi_clear_photons(self.h), "clear_photons")

    def tracePhotons(self, count=None, max_depth=5, seed=None):
        """RayTracer::tracePhotons(5, photons, ...) on the device, then PhotonMap::rebuild on the host and upload
        (include/raytracer.h:61-72,582-715).  Returns (photons [n][9], emission tries)."""
        count = self.photons if count is None else count
        n_light = self.scene.desc().n_light
        cap = max(count * max(n_light, 1), 1)
        out = np.zeros((cap, 9))
        tries = C.c_int64()
        n = self._check(self.L.gi_emit_photons(self.h, count, max_depth, C.c_uint64(self.seed if seed is None else seed), _p(out), cap, C.byref(tries)), "emit_photons")
        ph = out[:n].copy()
        self.scene.build_photon_map(ph)
        self.upload_photon_map()
        return ph, tries.value

    def build_photon_map_on_device(self, photons, box=None):
        """gi_build_photon_map: the photon octree and its candidate ranges built on the device from photons [n][9]."""
        ph = _f64(photons).reshape(-1, 9)
        b = _f64(box) if box is not None else None
        self._check(self.L.gi_build_photon_map(self.h, len(ph), _p(ph), _p(b)), "build_photon_map")

    def tracePhotonsOnDevice(self, count=None, max_depth=5, seed=None):
        """gi_trace_photons: emission and photon-map build without leaving the device; returns (photons stored, emission tries)."""
        count = self.photons if count is None else count
        tries = C.c_int64()
        n = self._check(self.L.gi_trace_photons(self.h, count, max_depth, C.c_uint64(self.seed if seed is None else seed), None, C.byref(tries)), "trace_photons")
        return n, tries.value

    def photon_tables_on_device(self):
        """The photon-map tables as the gather kernel sees them (gi_debug_photon_tables)."""
        nn, nr, nph = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.L.gi_debug_photon_tables(self.h, C.byref(nn), C.byref(nr), C.byref(nph), None, None, None, None), "photon_tables")
        nodes = np.zeros((nn.value, 128), np.uint8); ranges = np.zeros((nr.value, 2), np.int32)
        pos = np.zeros((nph.value, 3)); dircol = np.zeros((nph.value, 6))
        if nn.value:
            self._check(self.L.gi_debug_photon_tables(self.h, None, None, None, nodes.ctypes.data_as(C.c_void_p), _p(ranges, _ip), _p(pos), _p(dircol)), "photon_tables")
        return {"nodes": nodes, "ranges": ranges, "pos": pos, "dircol": dircol}

    def params(self, w, h, stripe_h=None, rank=0, world=1, min_samples=None, max_samples=None, noise_thresh=None, seed=None):
        p = RenderParams()
        p.cam_pos[:] = self.cam_pos; p.cam_up[:] = self.cam_up; p.cam_forward[:] = self.cam_forward
        p.sensor_diag, p.focal_dist = self.sensor_diag, self.focal_dist
        p.width, p.height = w, h
        p.stripe_h, p.stripe_rank, p.stripe_world = (h if stripe_h is None else stripe_h), rank, world
        p.min_samples = self.min_samples if min_samples is None else min_samples
        p.max_samples = self.max_samples if max_samples is None else max_samples
        p.noise_thresh = self.noise_thresh if noise_thresh is None else noise_thresh
        p.seed = self.seed if seed is None else seed
        return p

    def local_rows(self, p):
        return self.L.gi_local_rows(C.byref(p))

    def run(self, w, h, f64=True, want_spp=False, **kw):
        """RayTracer::run(w, h) pixel loop -> linear radiance [rows][w][3] on the host."""
        p = self.params(w, h, **kw)
        rows = self.local_rows(p)
        out = np.zeros((rows, w, 3), np.float64 if f64 else np.float32)
        spp = np.zeros((rows, w), np.int32) if want_spp else None
        self._check(self.L.gi_render_host(self.h, C.byref(p), out.ctypes.data_as(C.c_void_p), 1 if f64 else 0,
                                          spp.ctypes.data_as(C.c_void_p) if want_spp else None, None), "render_host")
        return (out, spp) if want_spp else out

    def run_device(self, p, out_ptr, f64=False, spp_ptr=None):
        """Render into device memory (bench / multi-GPU path); out_ptr is a raw device pointer."""
        self._check(self.L.gi_render_device(self.h, C.byref(p), C.c_void_p(out_ptr), 1 if f64 else 0, C.c_void_p(spp_ptr) if spp_ptr else None, None), "render_device")

    def set_render_mode(self, mode):
        """'wavefront' (default) or 'megakernel': two schedules of the same per-path arithmetic."""
        self._check(self.L.gi_set_render_mode(self.h, {"wavefront": 0, "megakernel": 1, "rounds": 2}[mode]), "set_render_mode")

    def set_wide_nodes(self, on):
        """Octree walk over wide records (default) or one box test per node record; returns whether the wide walk is in use."""
        rc = self.L.gi_set_wide_nodes(self.h, 1 if on else 0)
        if rc < 0:
            self._check(rc, "set_wide_nodes")
        self.photon_planes = bool(rc & 2)      # the photon octree's one-record-per-level descent is in use
        return bool(rc & 1)

    def set_content_culling(self, on):
        """Skip children of the octree walk in whose sub-tree the ray cannot hit anything (default on); returns whether it is in use."""
        return bool(self._check(self.L.gi_set_content_culling(self.h, 1 if on else 0), "set_content_culling"))

    def set_entity_boxes(self, on):
        """Run the entity tests of a leaf only on the references whose own box the ray touches (default on); returns whether it is in use."""
        return bool(self._check(self.L.gi_set_entity_boxes(self.h, 1 if on else 0), "set_entity_boxes"))

    def set_pool_slots(self, slots):
        self._check(self.L.gi_set_pool_slots(self.h, int(slots)), "set_pool_slots")

    def last_render_ms(self):
        ms, n = C.c_float(), C.c_int32()
        self._check(self.L.gi_last_render_ms(self.h, C.byref(ms), C.byref(n)), "last_render_ms")
        return ms.value, n.value

    STAGES = ("regen", "trace", "shade", "sort", "gather", "finish", "accum", "other")

    def last_stage_ms(self):
        out = (C.c_float * 8)()
        self._check(self.L.gi_last_stage_ms(self.h, out), "last_stage_ms")
        return dict(zip(self.STAGES, [float(v) for v in out]))

    KERNELS = ("regen", "trace", "shade", "sort", "gather", "finish", "accum", "other", "shadow", "reserved")

    def last_kernel_ms(self):
        """gi_last_kernel_ms: device time of the last frame per kernel family ('shade' = k_st_shade alone, 'shadow' = k_st_shadow)."""
        out = (C.c_float * 10)()
        self._check(self.L.gi_last_kernel_ms(self.h, out), "last_kernel_ms")
        return {k: float(v) for k, v in zip(self.KERNELS, out) if k != "reserved"}

    def set_counters(self, mode):
        """0 / False: off; 1 / True: the reference's visits (megakernel, per-node walk); 2 or "stream": what the streaming kernels execute."""
        m = {"stream": 2, True: 1, False: 0}.get(mode, mode)
        self._check(self.L.gi_set_counters(self.h, int(m)), "set_counters")

    STREAM_COUNTERS = ("trace_walks", "trace_records", "trace_child_boxes", "trace_content_boxes", "trace_leaves", "trace_tris", "trace_rays",
                       "shadow_walks", "shadow_records", "shadow_child_boxes", "shadow_content_boxes", "shadow_leaves", "shadow_tris", "shadow_rays",
                       "gather_queries", "gather_candidates", "shaded", "trace_entity_boxes", "shadow_entity_boxes")

    def stream_counters(self):
        """gi_get_stream_counters: the executed work of the last frame rendered with set_counters("stream"), as a dict."""
        out = (C.c_int64 * 19)()
        self._check(self.L.gi_get_stream_counters(self.h, out), "get_stream_counters")
        return dict(zip(self.STREAM_COUNTERS, [int(v) for v in out]))

    def counters(self):
        out = (C.c_int64 * 8)()
        self._check(self.L.gi_get_counters(self.h, out), "get_counters")
        return np.array(list(out), np.int64)

    def trace(self, rays):
        rays = _f64(rays).reshape(-1, 6)
        n = len(rays)
        hit = np.zeros(n, np.int32); ent = np.zeros(n, np.int32); res = np.zeros((n, 8))
        self._check(self.L.gi_trace(self.h, n, _p(rays), _p(hit, _ip), _p(ent, _ip), _p(res)), "trace")
        return hit, ent, res

    def visible(self, q):
        q = _f64(q).reshape(-1, 6)
        vis = np.zeros(len(q), np.int32)
        self._check(self.L.gi_visible(self.h, len(q), _p(q), _p(vis, _ip)), "visible")
        return vis

    def samplePhotons(self, q):
        q = _f64(q).reshape(-1, 6)
        res = np.zeros((len(q), 3)); nc = np.zeros(len(q), np.int32)
        self._check(self.L.gi_gather(self.h, len(q), _p(q), _p(res), _p(nc, _ip)), "gather")
        return res, nc

    def radiance(self, rays, stream, seed=None):
        rays = _f64(rays).reshape(-1, 6)
        stream = np.ascontiguousarray(stream, np.uint32)
        out = np.zeros((len(rays), 3))
        self._check(self.L.gi_radiance(self.h, len(rays), _p(rays), _p(stream, _up), C.c_uint64(self.seed if seed is None else seed), _p(out)), "radiance")
        return out

    KAT = {"fastPow": 0, "fastPrecisePow": 1, "hemisphereSample_cos": 2, "sample_phong": 3, "sphereCapSample_cos": 4, "randomUnitVec": 5, "refr": 6, "reflect": 7, "rng": 8,
           "sin": 16, "cos": 17, "acos": 18, "asin": 19, "atan2": 20, "pow": 21, "sqrt": 22}

    def kat(self, what, args):
        """Known answers of the scalar building blocks as the device computes them (include/gi_hip.h: gi_kat); args [n][k] -> [n][3]."""
        a = _f64(args)
        a = a.reshape(len(a), -1)
        out = np.zeros((len(a), 3))
        self._check(self.L.gi_kat(self.h, self.KAT[what], len(a), _p(a), a.shape[1], _p(out)), "kat")
        return out

    def sort_pairs(self, keys, vals, begin_bit=0, end_bit=32):
        """gi_debug_sort_pairs: the pipeline's radix sort on caller data (stable, by bits [begin_bit, end_bit) of the key)."""
        k = np.ascontiguousarray(keys, np.uint32); v = np.ascontiguousarray(vals, np.uint32)
        ko = np.zeros_like(k); vo = np.zeros_like(v)
        self._check(self.L.gi_debug_sort_pairs(self.h, len(k), _p(k, _up), _p(v, _up), begin_bit, end_bit, _p(ko, _up), _p(vo, _up)), "sort_pairs")
        return ko, vo

    def find_leaves(self, pos):
        """gi_debug_find_leaves: (fast, full) photon-map leaf of each position; fast = -2 where the quick descent declines."""
        p = _f64(pos).reshape(-1, 3)
        a = np.zeros(len(p), np.int32); b = np.zeros(len(p), np.int32)
        self._check(self.L.gi_debug_find_leaves(self.h, len(p), _p(p), _p(a, _ip), _p(b, _ip)), "find_leaves")
        return a, b

    def leaf_order(self, rays, cap=256):
        """Octree::intersectSorted as the device walk produces it: per ray the pre-order indices of the non-empty leaves in visiting order."""
        rays = _f64(rays).reshape(-1, 6)
        leaf = np.zeros((len(rays), cap), np.int32); n = np.zeros(len(rays), np.int32)
        self._check(self.L.gi_debug_leaf_order(self.h, len(rays), _p(rays), cap, _p(leaf, _ip), _p(n, _ip)), "leaf_order")
        assert (n <= cap).all(), "leaf_order: cap too small"
        return [leaf[i, :n[i]].copy() for i in range(len(rays))]

    def halton_sample(self, dim, index):
        dim = np.ascontiguousarray(dim, np.uint32); index = np.ascontiguousarray(index, np.uint32)
        out = np.zeros(len(dim), np.float32)
        self._check(self.L.gi_halton_sample(self.h, len(dim), _p(dim, _up), _p(index, _up), out.ctypes.data_as(C.POINTER(C.c_float))), "halton_sample")
        return out

    def halton_index(self, w, h, sxy):
        sxy = np.ascontiguousarray(sxy, np.uint32).reshape(-1, 3)
        out = np.zeros(len(sxy), np.uint32)
        self._check(self.L.gi_halton_index(self.h, w, h, len(sxy), _p(sxy, _up), _p(out, _up)), "halton_index")
        return out
