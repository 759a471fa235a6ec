"""ctypes binding of the CPU oracle (oracle/erasor_oracle.cpp).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (erasor_amd) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liberasor_oracle.so")


class Params(C.Structure):
    """mirror of erasor_params (include/erasor_hip.h)"""
    _fields_ = [
        ("max_range", C.c_double), ("num_rings", C.c_int32), ("num_sectors", C.c_int32),
        ("max_h", C.c_double), ("min_h", C.c_double), ("th_bin_max_h", C.c_double),
        ("scan_ratio_threshold", C.c_double), ("num_lowest_pts", C.c_int32),
        ("minimum_num_pts", C.c_int32), ("rejection_ratio", C.c_double),
        ("gf_dist_thr", C.c_double), ("gf_iter", C.c_int32), ("gf_num_lpr", C.c_int32),
        ("gf_th_seeds_height", C.c_double), ("map_voxel_size", C.c_double),
        ("version", C.c_int32), ("query_voxel_size", C.c_double),
        ("removal_interval", C.c_int32), ("voi_max_range", C.c_double),
        ("is_large_scale", C.c_int32), ("reserved0_", C.c_int32), ("submap_size", C.c_double),
        ("reserved_", C.c_int32 * 3),
    ]


class StepResult(C.Structure):
    """mirror of erasor_step_result (include/erasor_hip.h)"""
    _fields_ = [(k, C.c_uint64) for k in (
        "n_map_in", "n_voi", "n_outskirts", "n_query", "n_static_estimate", "n_complement",
        "n_map_rejected", "n_curr_rejected", "n_ground", "n_map_out", "n_static", "n_dynamic")] + [
        (k, C.c_uint32) for k in (
            "n_reverted_bins", "n_neg_sector", "n_ambiguous", "n_degenerate_plane",
            "n_voxel_overflow", "n_sort_fallback")] + [("reserved_", C.c_uint32 * 6)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved_"}


def build(force=False):
    src = os.path.join(_HERE, "erasor_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "erasor_hip.h")
    if (force or not os.path.exists(_SO)
            or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_xy2theta.restype = C.c_double
        _lib.orc_xy2theta.argtypes = [C.c_double, C.c_double]
        _lib.orc_bin_of.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def params_default():
    p = Params()
    lib().orc_params_default(C.byref(p))
    return p


def geopose2eigen(pose7):
    pose = np.ascontiguousarray(pose7, dtype=np.float64)
    T = np.zeros(16, np.float32)
    lib().orc_geopose2eigen(_p(pose), _p(T))
    return T


def invert4(T):
    T = _f32(T).reshape(16)
    Ti = np.zeros(16, np.float32)
    rc = lib().orc_invert4(_p(T), _p(Ti))
    assert rc == 0
    return Ti


def transform(cloud, T):
    cloud = _f32(cloud).reshape(-1, 4)
    out = np.empty_like(cloud)
    lib().orc_transform(_p(cloud), C.c_size_t(len(cloud)), _p(_f32(T).reshape(16)), _p(out))
    return out


def voxelize_preserving_labels(cloud, leaf):
    cloud = _f32(cloud).reshape(-1, 4)
    out = np.empty_like(cloud)
    n = C.c_size_t(0)
    rc = lib().orc_voxelize_preserving_labels(_p(cloud), C.c_size_t(len(cloud)), C.c_double(leaf), _p(out),
                                              C.c_size_t(len(cloud)), C.byref(n))
    assert rc == 0, rc
    return out[: n.value].copy()


def voxel_grid(cloud, leaf):
    """raw VoxelGrid: (centroids, sorted_pi, sorted_idx, overflow)"""
    cloud = _f32(cloud).reshape(-1, 4)
    out = np.empty_like(cloud)
    pi = np.zeros(len(cloud), np.uint32)
    idx = np.zeros(len(cloud), np.uint32)
    n = C.c_size_t(0)
    rc = lib().orc_voxel_grid(_p(cloud), C.c_size_t(len(cloud)), C.c_double(leaf), _p(out), C.c_size_t(len(cloud)),
                              C.byref(n), _p(pi), _p(idx))
    return out[: n.value].copy(), pi, idx, rc == 1


def std_sort_u32(keys, vals):
    keys = np.ascontiguousarray(keys, dtype=np.uint32).copy()
    vals = np.ascontiguousarray(vals, dtype=np.uint32).copy()
    lib().orc_std_sort_u32(_p(keys), _p(vals), C.c_size_t(len(keys)))
    return keys, vals


def std_sort_z(cloud):
    cloud = _f32(cloud).reshape(-1, 4)
    perm = np.zeros(len(cloud), np.uint32)
    lib().orc_std_sort_z(_p(cloud), C.c_size_t(len(cloud)), _p(perm))
    return perm


def mean_and_cov(cloud):
    cloud = _f32(cloud).reshape(-1, 4)
    cov = np.zeros(9, np.float32)
    mean = np.zeros(4, np.float32)
    lib().orc_mean_and_cov(_p(cloud), C.c_size_t(len(cloud)), _p(cov), _p(mean))
    return cov.reshape(3, 3), mean


def jacobi_svd3(cov):
    cov = _f32(cov).reshape(9)
    U = np.zeros(9, np.float32)
    sv = np.zeros(3, np.float32)
    lib().orc_jacobi_svd3(_p(cov), _p(U), _p(sv))
    return U.reshape(3, 3), sv


def bin_of(params, x, y, z):
    return int(lib().orc_bin_of(C.byref(params), C.c_float(x), C.c_float(y), C.c_float(z)))


def xy2theta(x, y):
    return float(lib().orc_xy2theta(x, y))


def extract_ground(params, cloud):
    cloud = _f32(cloud).reshape(-1, 4)
    mask = np.zeros(len(cloud), np.uint8)
    normals = np.zeros(params.gf_iter * 3, np.float32)
    ds = np.zeros(params.gf_iter, np.float64)
    ndeg = C.c_uint32(0)
    lib().orc_extract_ground(C.byref(params), _p(cloud), C.c_size_t(len(cloud)), _p(mask), _p(normals), _p(ds),
                             C.byref(ndeg))
    return mask.astype(bool), normals.reshape(-1, 3), ds, int(ndeg.value)


class Oracle:
    """CPU restatement of OfflineMapUpdater's hot path (non-large-scale mode)."""

    def __init__(self, params):
        self.params = params
        self.h = C.c_void_p(lib().orc_create(C.byref(params)))
        if not self.h:
            raise ValueError("orc_create failed")
        self.B = params.num_rings * params.num_sectors

    def close(self):
        if self.h:
            lib().orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_map(self, cloud):
        cloud = _f32(cloud).reshape(-1, 4)
        lib().orc_set_map(self.h, _p(cloud), C.c_size_t(len(cloud)))

    def step(self, scan, T_l2b, T_b2o, T_o2b=None):
        scan = _f32(scan).reshape(-1, 4)
        T_l2b = _f32(T_l2b).reshape(16)
        T_b2o = _f32(T_b2o).reshape(16)
        T_o2b = invert4(T_b2o) if T_o2b is None else _f32(T_o2b).reshape(16)
        res = StepResult()
        rc = lib().orc_step(self.h, _p(scan), C.c_size_t(len(scan)), _p(T_l2b), _p(T_b2o), _p(T_o2b), C.byref(res))
        if rc != 0:
            raise RuntimeError("orc_step rc=%d" % rc)
        return res

    def map_size(self):
        n = C.c_size_t(0)
        lib().orc_map_size(self.h, C.byref(n))
        return n.value

    def get_map(self):
        return self.get_cloud(7)

    def get_cloud(self, which):
        n = C.c_size_t(0)
        lib().orc_get_cloud(self.h, which, None, C.c_size_t(0), C.byref(n))
        out = np.empty((n.value, 4), np.float32)
        rc = lib().orc_get_cloud(self.h, which, _p(out), C.c_size_t(n.value), C.byref(n))
        assert rc == 0
        return out

    def get_rejected_indices(self):
        n = C.c_size_t(0)
        lib().orc_get_rejected_indices(self.h, None, C.c_size_t(0), C.byref(n))
        out = np.empty(n.value, np.uint64)
        lib().orc_get_rejected_indices(self.h, _p(out), C.c_size_t(n.value), C.byref(n))
        return out

    def get_bins(self, which):
        cnt = np.zeros(self.B, np.uint32)
        mn = np.zeros(self.B, np.float64)
        mx = np.zeros(self.B, np.float64)
        lib().orc_get_bins(self.h, which, _p(cnt), _p(mn), _p(mx))
        return cnt, mn, mx

    def get_status(self):
        st = np.zeros(self.B, np.float64)
        lib().orc_get_status(self.h, _p(st))
        return st

    def get_planes(self):
        n = C.c_size_t(0)
        lib().orc_get_planes(self.h, None, None, None, C.c_size_t(0), C.byref(n))
        nb, it = n.value, self.params.gf_iter
        bins = np.zeros(nb, np.uint32)
        normal = np.zeros((nb, it, 3), np.float32)
        d = np.zeros((nb, it), np.float64)
        if nb:
            lib().orc_get_planes(self.h, _p(bins), _p(normal), _p(d), C.c_size_t(nb), C.byref(n))
        return bins, normal, d

    def get_voi_codes(self):
        n = C.c_size_t(0)
        lib().orc_get_voi_codes(self.h, None, None, C.c_size_t(0), C.byref(n))
        code = np.zeros(n.value, np.int32)
        src = np.zeros(n.value, np.uint64)
        if n.value:
            lib().orc_get_voi_codes(self.h, _p(code), _p(src), C.c_size_t(n.value), C.byref(n))
        return code, src


class Mapgen:
    """CPU restatement of class mapgen (src/mapgen/mapgen.hpp:27-305) over the oracle's C functions.
    TEST INFRASTRUCTURE ONLY, like the rest of oracle/ (parity unpinned: the reference has no mapgen fixtures)."""

    CAR_BODY_SIZE = 2.7  # mapgen.hpp:8

    def __init__(self, leafsize, is_large_scale=False):  # setValue, mapgen.hpp:182-196
        self.leafsize = leafsize
        self.is_large_scale = is_large_scale
        self.is_initial = True
        self.cloud_curr = np.zeros((0, 4), np.float32)
        self.cloud_map = np.zeros((0, 4), np.float32)
        self.cloud_maps = []
        self.cnt_voxel = 0  # the function-local static of accumPointCloud (:248)
        self.accum_count = 0

    def accum(self, scan, T_pose, T_lidar2origin=None):  # accumPointCloud, mapgen.hpp:198-257
        scan = np.ascontiguousarray(scan, np.float32).reshape(-1, 4)
        if T_lidar2origin is None:
            T_lidar2origin = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1.73], [0, 0, 0, 1]], np.float32)  # :212-215
        # :219-228  float max_dist_square = pow(CAR_BODY_SIZE, 2); double dist_square = pow(pt.x, 2) + pow(pt.y, 2)
        max_dist_square = np.float32(self.CAR_BODY_SIZE ** 2)
        x, y = scan[:, 0].astype(np.float64), scan[:, 1].astype(np.float64)
        dist_square = x * x + y * y
        outliers = scan[~(dist_square < np.float64(max_dist_square))]
        world = transform(transform(outliers, T_lidar2origin), T_pose)  # :231-237
        self.cloud_curr = voxelize_preserving_labels(world, 0.2)  # :239
        if self.is_initial:
            self.cloud_map = self.cloud_curr.copy()
            self.is_initial = False
        else:
            self.cloud_map = np.concatenate([self.cloud_map, self.cloud_curr])
            if self.is_large_scale:
                if self.cnt_voxel % 500 == 0:  # :248-255
                    self.cloud_maps.append(voxelize_preserving_labels(self.cloud_map, self.leafsize))
                    self.cloud_map = np.zeros((0, 4), np.float32)
                self.cnt_voxel += 1
            self.accum_count += 1
        return len(self.cloud_curr)

    def naive_map(self):  # saveNaiveMap's cloud_src, :267-279
        if self.is_large_scale:
            return np.concatenate(self.cloud_maps + [self.cloud_map]) if (self.cloud_maps or len(self.cloud_map)) else self.cloud_map
        return self.cloud_map

    def save(self):  # :281-299
        return voxelize_preserving_labels(self.naive_map(), self.leafsize)
