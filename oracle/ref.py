"""ctypes binding of oracle/_ref/liberasor_ref.so — the reference's OWN sources (erasor.cpp, erasor_utils.cpp,
OfflineMapUpdater.cpp, mapgen.hpp), compiled unmodified against oracle/stubs/ by oracle/ref.mk.

TEST INFRASTRUCTURE ONLY — importable from tests/ and the cpu_baseline leg of bench.py.  The product package
(erasor_amd) never imports this.  The library is built only where /root/reference exists (this container); the
GPU box uses the prebuilt file that travels with the snapshot.

The reference keeps function-static state (OMU.cpp:206 stack_count, :344 half_size, utils.cpp:88, mapgen.hpp:248),
i.e. it assumes one updater per process.  Every RefUpdater / RefMapgen therefore loads its own private copy of the
.so, so the statics start fresh exactly as in a freshly started node.
"""
import ctypes as C
import os
import shutil
import subprocess
import tempfile

import numpy as np

from .orc import Params, _f32, _p  # same erasor_params layout

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "liberasor_ref.so")
REFERENCE = "/root/reference"


def available():
    return os.path.exists(SO)


def build(force=False):
    """compile the reference where it lies; only possible where /root/reference exists"""
    if not os.path.isdir(REFERENCE):
        return SO if os.path.exists(SO) else None
    if force and os.path.exists(SO):
        os.remove(SO)
    subprocess.check_call(["make", "-C", _HERE, "-f", "ref.mk", "-s"])
    return SO


def _load_private():
    """a fresh copy of the library = fresh function-static state"""
    if not available():
        build()
    if not available():
        raise RuntimeError("oracle/_ref/liberasor_ref.so is absent and /root/reference is not here to build it")
    d = tempfile.mkdtemp(prefix="erasor_ref_")
    path = os.path.join(d, "liberasor_ref_%d.so" % os.getpid())
    shutil.copy(SO, path)
    lib = C.CDLL(path)
    shutil.rmtree(d, ignore_errors=True)  # the mapping stays valid after unlink
    lib.ref_create.restype = C.c_void_p
    lib.ref_mapgen_create.restype = C.c_void_p
    lib.ref_mapgen_create.argtypes = [C.c_float, C.c_int]
    lib.ref_last_error.restype = C.c_char_p
    lib.ref_erasor_get_max_range.restype = C.c_double
    return lib


_shared = None


def _lib():
    """one shared copy for the stateless free functions"""
    global _shared
    if _shared is None:
        _shared = _load_private()
    return _shared


def _cloud_out(fn):
    n = C.c_size_t(0)
    rc = fn(None, C.c_size_t(0), C.byref(n))
    assert rc == 0, rc
    out = np.empty((n.value, 4), np.float32)
    rc = fn(_p(out), C.c_size_t(n.value), C.byref(n))
    assert rc == 0, rc
    return out


def geopose2eigen(pose7):
    pose = np.ascontiguousarray(pose7, dtype=np.float64)
    T = np.zeros(16, np.float32)
    _lib().ref_geopose2eigen(_p(pose), _p(T))
    return T


def eigen2geopose(T):
    T = _f32(T).reshape(16)
    pose = np.zeros(7, np.float64)
    _lib().ref_eigen2geopose(_p(T), _p(pose))
    return pose


def voxelize_preserving_labels(cloud, leaf):
    cloud = _f32(cloud).reshape(-1, 4)
    out = np.empty_like(cloud)
    n = C.c_size_t(0)
    rc = _lib().ref_voxelize_preserving_labels(_p(cloud), C.c_size_t(len(cloud)), C.c_double(leaf), _p(out),
                                               C.c_size_t(len(cloud)), C.byref(n))
    assert rc == 0, rc
    return out[: n.value].copy()


def count_stat_dyn(cloud):
    cloud = _f32(cloud).reshape(-1, 4)
    ns, nd = C.c_int(0), C.c_int(0)
    _lib().ref_count_stat_dyn(_p(cloud), C.c_size_t(len(cloud)), C.byref(ns), C.byref(nd))
    return ns.value, nd.value


def parse_dynamic_obj(cloud):
    cloud = _f32(cloud).reshape(-1, 4)
    d = np.empty_like(cloud)
    s = np.empty_like(cloud)
    nd, ns = C.c_size_t(0), C.c_size_t(0)
    _lib().ref_parse_dynamic_obj(_p(cloud), C.c_size_t(len(cloud)), _p(d), _p(s), C.byref(nd), C.byref(ns))
    return d[: nd.value].copy(), s[: ns.value].copy()


class RefUpdater:
    """the reference's erasor::OfflineMapUpdater (+ its ERASOR), driven through its ROS callback"""

    def __init__(self, params, map_cloud, lidar2body7=(0, 0, 0, 0, 0, 0, 1), honour_removal_interval=False):
        self.lib = _load_private()
        if not honour_removal_interval:  # the oracle / C ABI step on every call; the gate (OMU.cpp:206-209) is the caller's
            q = Params()
            C.memmove(C.byref(q), C.byref(params), C.sizeof(q))
            q.removal_interval = 1
            params = q
        self.params = params
        self.B = params.num_rings * params.num_sectors
        m = _f32(map_cloud).reshape(-1, 4)
        l2b = np.ascontiguousarray(lidar2body7, dtype=np.float64)
        self.h = C.c_void_p(self.lib.ref_create(C.byref(params), _p(l2b), _p(m), C.c_size_t(len(m))))
        if not self.h:
            raise RuntimeError("ref_create: %s" % self.lib.ref_last_error().decode())
        self.seq = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.ref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, scan, pose7):
        """returns 0, or raises with the exception text the reference threw"""
        scan = _f32(scan).reshape(-1, 4)
        pose = np.ascontiguousarray(pose7, dtype=np.float64)
        rc = self.lib.ref_step(self.h, _p(scan), C.c_size_t(len(scan)), _p(pose), C.c_uint32(self.seq))
        self.seq += 1
        if rc != 0:
            raise RuntimeError("ref_step rc=%d: %s" % (rc, self.lib.ref_last_error().decode()))
        return rc

    def get_cloud(self, which):
        return _cloud_out(lambda d, c, n: self.lib.ref_get_cloud(self.h, which, d, c, n))

    def get_map(self):
        return self.get_cloud(7)

    def get_bins(self, which):
        cnt = np.zeros(self.B, np.uint32)
        mn = np.zeros(self.B, np.float64)
        mx = np.zeros(self.B, np.float64)
        self.lib.ref_get_bins(self.h, which, _p(cnt), _p(mn), _p(mx))
        return cnt, mn, mx

    def get_status(self):
        st = np.zeros(self.B, np.float64)
        self.lib.ref_get_status(self.h, _p(st))
        return st

    def get_planes(self):
        """(normals[n_calls,3], last d_, last th_dist_d_): one row per estimate_plane_ call of the last step"""
        n = C.c_size_t(0)
        d, th = C.c_double(0), C.c_double(0)
        self.lib.ref_get_planes(self.h, None, C.c_size_t(0), C.byref(n), C.byref(d), C.byref(th))
        out = np.zeros((n.value, 3), np.float32)
        if n.value:
            self.lib.ref_get_planes(self.h, _p(out), C.c_size_t(n.value), C.byref(n), C.byref(d), C.byref(th))
        return out, d.value, th.value

    def polygon_likelihood(self):
        """likelihood[] of the last /SCDR/debug/polygons_marker message, in the reference's push order"""
        n = C.c_size_t(0)
        rc = self.lib.ref_get_polygon_likelihood(None, C.c_size_t(0), C.byref(n))
        assert rc == 0, rc
        out = np.zeros(n.value, np.float32)
        if n.value:
            self.lib.ref_get_polygon_likelihood(_p(out), C.c_size_t(n.value), C.byref(n))
        return out

    def polygon(self, k):
        out = np.zeros((16, 3), np.float32)
        n = self.lib.ref_get_polygon(C.c_size_t(k), _p(out), C.c_size_t(16))
        assert n >= 0, n
        return out[:n].copy()

    def get_matrices(self):
        a = np.zeros(16, np.float32)
        b = np.zeros(16, np.float32)
        self.lib.ref_get_matrices(self.h, _p(a), _p(b))
        return a, b

    def label_counts(self):
        out = np.zeros(2, np.uint64)
        self.lib.ref_get_label_counts(self.h, _p(out))
        return int(out[0]), int(out[1])

    def spans(self):
        """the reference's own two wall-clock spans of the last step, seconds ('Extracting VoI', 'ERASOR')"""
        out = np.zeros(2, np.float64)
        self.lib.ref_get_spans(_p(out))
        return float(out[0]), float(out[1])

    def erasor_run(self, map_voi, query_voi, version=3):
        """ERASOR::set_inputs + compare_vois_and_revert_ground[_w_block] on egocentric clouds; returns the rc
        (-100 = the reference threw std::out_of_range)"""
        m = _f32(map_voi).reshape(-1, 4)
        s = _f32(query_voi).reshape(-1, 4)
        return self.lib.ref_erasor_run(self.h, _p(m), C.c_size_t(len(m)), _p(s), C.c_size_t(len(s)), int(version))

    def erasor_get(self, which):
        return _cloud_out(lambda d, c, n: self.lib.ref_erasor_get(self.h, which, d, c, n))

    def is_dynamic_obj_close(self, r, theta, r_range=1, theta_range=1):
        return bool(self.lib.ref_is_dynamic_obj_close(self.h, r, theta, r_range, theta_range))

    def save_static_map(self, voxel_size):
        return _cloud_out(lambda d, c, n: self.lib.ref_save_static_map(self.h, C.c_float(voxel_size), d, c, n))


class RefMapgen:
    """the reference's class mapgen (src/mapgen/mapgen.hpp)"""

    def __init__(self, leafsize, is_large_scale=False):
        self.lib = _load_private()
        self.h = C.c_void_p(self.lib.ref_mapgen_create(C.c_float(leafsize), int(is_large_scale)))

    def accum(self, scan, pose7):
        scan = _f32(scan).reshape(-1, 4)
        pose = np.ascontiguousarray(pose7, dtype=np.float64)
        n = C.c_size_t(0)
        rc = self.lib.ref_mapgen_accum(self.h, _p(scan), C.c_size_t(len(scan)), _p(pose), C.byref(n))
        assert rc == 0, self.lib.ref_last_error()
        return n.value

    def get(self, which):
        return _cloud_out(lambda d, c, n: self.lib.ref_mapgen_get(self.h, which, d, c, n))

    def save(self):
        rc = self.lib.ref_mapgen_save(self.h)
        assert rc == 0, self.lib.ref_last_error()
        return (_cloud_out(lambda d, c, n: self.lib.ref_mapgen_saved(0, d, c, n)),
                _cloud_out(lambda d, c, n: self.lib.ref_mapgen_saved(1, d, c, n)))
