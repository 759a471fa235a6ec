"""erasor_amd — MI355X-native ERASOR hot path.

The product is `liberasor_hip.so` (hand-written HIP for gfx950, C ABI in include/erasor_hip.h) plus the
C++ shim in erasor_amd/csrc/shim (reference-compatible ERASOR / OfflineMapUpdater / erasor_utils
surface).  This Python package is plumbing only: a ctypes binding used by tests, bench.py and
__graft_entry__.py.  There is no CPU fallback: if the HIP library or a GPU is missing, calls fail loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liberasor_hip.so")
_SRC_DIR = os.path.join(_HERE, "csrc")


class Params(C.Structure):
    """erasor_params (include/erasor_hip.h) — the reference's rosparam names (erasor.h:47-61, OMU.cpp:66-83)."""
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
    """erasor_step_result (include/erasor_hip.h)"""
    _fields_ = [(k, C.c_uint64) for k in (
        "n_map_in", "n_voi", "n_outskirts", "n_query", "n_static_estimate", "n_complement",
        "n_map_rejected", "n_curr_rejected", "n_ground", "n_map_out", "n_static", "n_dynamic")] + [
        (k, C.c_uint32) for k in (
            "n_reverted_bins", "n_neg_sector", "n_ambiguous", "n_degenerate_plane",
            "n_voxel_overflow", "n_sort_fallback")] + [("reserved_", C.c_uint32 * 6)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved_"}


CLOUD_QUERY_VOI, CLOUD_MAP_VOI, CLOUD_STATIC_ESTIMATE, CLOUD_COMPLEMENT = 0, 1, 2, 3
CLOUD_MAP_REJECTED, CLOUD_CURR_REJECTED, CLOUD_GROUND_VIZ, CLOUD_MAP = 4, 5, 6, 7

E_NO_DEVICE = -2


class ErasorError(RuntimeError):
    def __init__(self, rc, msg):
        super().__init__("erasor_hip rc=%d: %s" % (rc, msg))
        self.rc = rc


def build(force=False):
    """hipcc --offload-arch=gfx950 … -shared -> erasor_amd/liberasor_hip.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(_SRC_DIR, f) for f in ("erasor_hip.hip", "kernels.hip.h", "revert_bins.hip.h", "exact_sort.hip.h", "exact_sort_core.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "erasor_hip.h"))
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _SRC_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    """Load liberasor_hip.so.  Raises if it has not been built — there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ErasorError(E_NO_DEVICE, "liberasor_hip.so missing: run erasor_amd.build() (hipcc) first; no CPU fallback exists")
        l = C.CDLL(LIB_PATH)
        l.erasor_hip_version.restype = C.c_char_p
        l.erasor_hip_last_error.restype = C.c_char_p
        l.erasor_hip_last_error.argtypes = [C.c_void_p]
        l.erasor_hip_stream.restype = C.c_void_p
        l.erasor_hip_stream.argtypes = [C.c_void_p]
        l.erasor_hip_destroy.argtypes = [C.c_void_p]
        l.erasor_hip_destroy.restype = None
        _lib = l
    return _lib


def params_default():
    p = Params()
    rc = lib().erasor_hip_params_default(C.byref(p))
    assert rc == 0
    return p


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


Mat16 = C.c_float * 16


def c_mat(T):
    """a 4x4 row-major transform as a ctypes float[16], converted once: the step / prefetch wrappers take it as it is
    (the numpy -> ctypes conversion of three matrices costs ~10 us per call, a visible share of a 0.3 ms step)"""
    return T if isinstance(T, Mat16) else Mat16(*[float(v) for v in _f32(T).reshape(16)])


def _m(T):
    return T if isinstance(T, Mat16) else _p(_f32(T).reshape(16))


def geopose2eigen(pose7):
    """erasor_utils::geoPose2eigen (erasor_utils.cpp:35-55): tf::Matrix3x3(tf::Quaternion) evaluated in double with
    tf's association, each entry narrowed to float32.  pose7 = x y z qx qy qz qw.  Returns 16 floats, row-major."""
    px, py, pz, x, y, z, w = [float(v) for v in pose7]
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    T = np.array([1.0 - (yy + zz), xy - wz, xz + wy, px,
                  xy + wz, 1.0 - (xx + zz), yz - wx, py,
                  xz - wy, yz + wx, 1.0 - (xx + yy), pz,
                  0.0, 0.0, 0.0, 1.0], np.float64)
    return T.astype(np.float32)


def invert_rigid(T):
    """T_origin2body from T_body2origin: general 4x4 inverse in float64, narrowed to float32 (the caller owns this
    choice — the reference's Eigen SSE inverse (OMU.cpp:436) is not bit-reproducible across CPUs)."""
    T = np.asarray(T, np.float32).reshape(4, 4).astype(np.float64)
    return np.linalg.inv(T).astype(np.float32).reshape(16)


def replicate_map(handles, root=0):
    """erasor_hip_replicate_map: the map of handles[root] to every other handle (one per device) -- single-process RCCL broadcast over
    xGMI, or peer copies.  Returns the transport used (1 RCCL, 2 peer copies, 0 empty map)."""
    arr = (C.c_void_p * len(handles))(*[h._h for h in handles])
    t = C.c_int(0)
    rc = lib().erasor_hip_replicate_map(arr, C.c_int(len(handles)), C.c_int(root), C.byref(t))
    if rc != 0:
        raise ErasorError(rc, (lib().erasor_hip_last_error(handles[root]._h) or b"").decode())
    return t.value


class Erasor:
    """Handle of the HIP hot path (one GPU, one stream)."""

    def __init__(self, params, device=0):
        self.params = params
        self.B = params.num_rings * params.num_sectors
        self._h = C.c_void_p()
        rc = lib().erasor_hip_create(C.byref(params), C.c_int(device), C.byref(self._h))
        self._last_res = None
        if rc != 0:
            self._h = C.c_void_p()
            raise ErasorError(rc, "erasor_hip_create failed (no GPU / invalid parameters)")

    # -- lifetime --
    def close(self):
        if getattr(self, "_h", None):
            lib().erasor_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise ErasorError(rc, (lib().erasor_hip_last_error(self._h) or b"").decode())

    # -- map --
    def set_map(self, cloud):
        cloud = _f32(cloud).reshape(-1, 4)
        self._check(lib().erasor_hip_set_map(self._h, _p(cloud), C.c_size_t(len(cloud))))

    def set_map_device(self, dptr, n):
        self._check(lib().erasor_hip_set_map_device(self._h, C.c_void_p(dptr), C.c_size_t(n)))

    def map_size(self):
        n = C.c_size_t(0)
        self._check(lib().erasor_hip_map_size(self._h, C.byref(n)))
        return n.value

    def get_map(self):
        return self.get_cloud(CLOUD_MAP)

    # -- step --
    def prefetch(self, scan, T_l2b, T_b2o=None, T_o2b=None):
        """announce the next scan (host array): its query chain starts now, beside the step in flight.  With the node's pose its VoI
        split goes ahead as well; with T_o2b too (erasor_hip_announce_origin2body) the whole front of its step runs beside the per-bin
        launch of the step before it (overlapped steps)"""
        scan = _f32(scan).reshape(-1, 4)
        # every announced buffer stays alive until a step has consumed it (up to four scans can be outstanding; a freed
        # buffer's address could be handed to the next np.ascontiguousarray)
        self._keep = (getattr(self, "_keep", []) + [scan])[-4:]
        if T_b2o is None:
            self._check(lib().erasor_hip_prefetch_scan(self._h, _p(scan), C.c_size_t(len(scan)), C.c_int(0), _m(T_l2b)))
        else:
            self._check(lib().erasor_hip_prefetch_node(self._h, _p(scan), C.c_size_t(len(scan)), C.c_int(0), _m(T_l2b), _m(T_b2o)))
            if T_o2b is not None:
                self._check(lib().erasor_hip_announce_origin2body(self._h, _m(T_o2b)))
        return scan

    def prefetch_device(self, d_ptr, n, T_l2b, T_b2o=None, T_o2b=None):
        """announce the next scan (device buffer, read in place); with its pose the next step's VoI split is launched ahead"""
        if T_b2o is None:
            self._check(lib().erasor_hip_prefetch_scan(self._h, C.c_void_p(d_ptr), C.c_size_t(n), C.c_int(1), _m(T_l2b)))
        else:
            self._check(lib().erasor_hip_prefetch_node(self._h, C.c_void_p(d_ptr), C.c_size_t(n), C.c_int(1), _m(T_l2b), _m(T_b2o)))
            if T_o2b is not None:
                self._check(lib().erasor_hip_announce_origin2body(self._h, _m(T_o2b)))

    def step(self, scan, T_l2b, T_b2o, T_o2b):
        scan = _f32(scan).reshape(-1, 4)
        res = StepResult()
        self._check(lib().erasor_hip_step(self._h, _p(scan), C.c_size_t(len(scan)), _p(_f32(T_l2b).reshape(16)),
                                          _p(_f32(T_b2o).reshape(16)), _p(_f32(T_o2b).reshape(16)), C.byref(res)))
        self._last_res = res
        return res

    def step_device(self, dptr, n, T_l2b, T_b2o, T_o2b):
        res = StepResult()
        self._check(lib().erasor_hip_step_device(self._h, C.c_void_p(dptr), C.c_size_t(n), _m(T_l2b), _m(T_b2o), _m(T_o2b), C.byref(res)))
        self._last_res = res
        return res

    # -- host records in the caller's layout, announcements by ticket (erasor_hip_step_rows / _prefetch_node_rows / _step_ticket) --
    @staticmethod
    def _rows(rows):
        """rows: a C-contiguous 2-D array of 4-byte items, one record per row: x, y, z first; returns (array, stride in bytes)"""
        rows = np.ascontiguousarray(rows)
        assert rows.ndim == 2 and rows.itemsize == 4 and rows.shape[1] >= 4
        return rows, rows.shape[1] * 4

    def step_rows(self, rows, intensity_col, T_l2b, T_b2o, T_o2b):
        """a step on host records laid out like the caller has them (pcl::PointXYZI: 8 floats per row, intensity in column 4)"""
        rows, stride = self._rows(rows)
        res = StepResult()
        self._check(lib().erasor_hip_step_rows(self._h, _p(rows), C.c_size_t(len(rows)), C.c_size_t(stride), C.c_size_t(4 * intensity_col),
                                               _m(T_l2b), _m(T_b2o), _m(T_o2b), C.byref(res)))
        self._last_res = res
        return res

    def prefetch_node_rows(self, rows, intensity_col, T_l2b, T_b2o=None, T_o2b=None):
        """announce the next node (host records); returns its ticket.  The buffer is the caller's again on return."""
        rows, stride = self._rows(rows)
        t = C.c_uint64(0)
        self._check(lib().erasor_hip_prefetch_node_rows(self._h, _p(rows), C.c_size_t(len(rows)), C.c_size_t(stride), C.c_size_t(4 * intensity_col),
                                                        _m(T_l2b), None if T_b2o is None else _m(T_b2o), C.byref(t)))
        if T_b2o is not None and T_o2b is not None:
            self._check(lib().erasor_hip_announce_origin2body(self._h, _m(T_o2b)))
        return t.value

    def step_ticket(self, ticket, T_b2o, T_o2b):
        res = StepResult()
        self._check(lib().erasor_hip_step_ticket(self._h, C.c_uint64(ticket), _m(T_b2o), _m(T_o2b), C.byref(res)))
        self._last_res = res
        return res

    def step_async(self, scan, n=None, T_l2b=None, T_b2o=None, T_o2b=None, device=False):
        """first half of a step (erasor_hip_step_async): everything is enqueued, nothing is waited for.  scan: a host array, or a
        device pointer with its point count (device=True).  step_wait() collects the results."""
        if device:
            ptr, cnt = C.c_void_p(scan), C.c_size_t(n)
        else:
            scan = _f32(scan).reshape(-1, 4)
            self._fly_keep = scan  # (must stay valid until step_wait has returned)
            ptr, cnt = _p(scan), C.c_size_t(len(scan))
        self._check(lib().erasor_hip_step_async(self._h, ptr, cnt, C.c_int(1 if device else 0), _m(T_l2b), _m(T_b2o), _m(T_o2b)))

    def step_wait(self):
        res = StepResult()
        self._check(lib().erasor_hip_step_wait(self._h, C.byref(res)))
        self._fly_keep = None
        self._last_res = res
        return res

    def run_nodes(self, scan_ptrs, n_pts, T_l2b, Tb, To, first, count, lookahead, announced, device=True):
        """erasor_hip_run_nodes: nodes [first, first + count) of a sequence in ONE native call (the offline driver's loop).
        scan_ptrs / n_pts: ctypes arrays over the whole sequence (see node_arrays); announced: a ctypes c_size_t carried
        between calls.  Returns the list of StepResult."""
        res = (StepResult * count)()
        self._check(lib().erasor_hip_run_nodes(self._h, scan_ptrs, n_pts, C.c_size_t(len(n_pts)), C.c_int(1 if device else 0), _m(T_l2b), Tb, To,
                                               C.c_size_t(first), C.c_size_t(count), C.c_int(lookahead), C.byref(announced), res))
        self._last_res = res[count - 1] if count else None
        return list(res)

    @staticmethod
    def node_arrays(ptrs, sizes, Tb_list, To_list):
        """the arguments of run_nodes from Python lists: device (or host) pointers, point counts, 4x4 matrices"""
        n = len(ptrs)
        P = (C.c_void_p * n)(*ptrs)
        N = (C.c_size_t * n)(*sizes)
        Tb = (C.c_float * (16 * n))(*[float(v) for t in Tb_list for v in np.asarray(t, np.float32).reshape(16)])
        To = (C.c_float * (16 * n))(*[float(v) for t in To_list for v in np.asarray(t, np.float32).reshape(16)])
        return P, N, Tb, To

    def last_result(self):
        """erasor_step_result of the last collected step (kept by the wrapper)"""
        return self._last_res

    def step_done(self):
        return bool(lib().erasor_hip_step_done(self._h))

    # -- read-back --
    def get_cloud(self, which):
        n = C.c_size_t(0)
        self._check(lib().erasor_hip_get_cloud(self._h, which, None, C.c_size_t(0), C.byref(n)))
        out = np.empty((n.value, 4), np.float32)
        self._check(lib().erasor_hip_get_cloud(self._h, which, _p(out), C.c_size_t(n.value), C.byref(n)))
        return out

    def get_rejected_indices(self):
        n = C.c_size_t(0)
        self._check(lib().erasor_hip_get_rejected_indices(self._h, None, C.c_size_t(0), C.byref(n)))
        out = np.empty(n.value, np.uint64)
        self._check(lib().erasor_hip_get_rejected_indices(self._h, _p(out), C.c_size_t(n.value), C.byref(n)))
        return out

    def get_bins(self, which):
        cnt = np.zeros(self.B, np.uint32)
        mn = np.zeros(self.B, np.float64)
        mx = np.zeros(self.B, np.float64)
        self._check(lib().erasor_hip_get_bins(self._h, which, _p(cnt), _p(mn), _p(mx)))
        return cnt, mn, mx

    def get_status(self):
        st = np.zeros(self.B, np.float64)
        self._check(lib().erasor_hip_get_status(self._h, _p(st)))
        return st

    def get_planes(self):
        n = C.c_size_t(0)
        self._check(lib().erasor_hip_get_planes(self._h, None, None, None, C.c_size_t(0), C.byref(n)))
        nb, it = n.value, self.params.gf_iter
        bins = np.zeros(nb, np.uint32)
        normal = np.zeros((nb, it, 3), np.float32)
        d = np.zeros((nb, it), np.float64)
        if nb:
            self._check(lib().erasor_hip_get_planes(self._h, _p(bins), _p(normal), _p(d), C.c_size_t(nb), C.byref(n)))
        return bins, normal, d

    def voxelize_preserving_labels(self, cloud, leaf):
        cloud = _f32(cloud).reshape(-1, 4)
        out = np.empty((max(len(cloud), 1), 4), np.float32)
        n = C.c_size_t(0)
        self._check(lib().erasor_hip_voxelize_preserving_labels(self._h, _p(cloud), C.c_size_t(len(cloud)), C.c_double(leaf), _p(out),
                                                                C.c_size_t(len(out)), C.byref(n)))
        return out[: n.value].copy()

    # -- mapgen (src/mapgen/mapgen.hpp) --
    def mapgen_begin(self, leafsize, is_large_scale=False):
        self._check(lib().erasor_hip_mapgen_begin(self._h, C.c_double(leafsize), C.c_int(int(is_large_scale))))

    def mapgen_accum(self, scan, T_pose, T_lidar2origin=None):
        scan = _f32(scan).reshape(-1, 4)
        tp = _f32(T_pose).reshape(16)
        tl = _f32(T_lidar2origin).reshape(16) if T_lidar2origin is not None else None
        n = C.c_size_t(0)
        self._check(lib().erasor_hip_mapgen_accum(self._h, _p(scan), C.c_size_t(len(scan)), _p(tp), _p(tl) if tl is not None else None,
                                                  C.byref(n)))
        return n.value

    def mapgen_get(self, which):
        n = C.c_size_t(0)
        self._check(lib().erasor_hip_mapgen_get(self._h, C.c_int(which), None, C.c_size_t(0), C.byref(n)))
        out = np.empty((max(n.value, 1), 4), np.float32)
        self._check(lib().erasor_hip_mapgen_get(self._h, C.c_int(which), _p(out), C.c_size_t(len(out)), C.byref(n)))
        return out[: n.value].copy()

    def mapgen_save(self):
        n = C.c_size_t(0)
        self._check(lib().erasor_hip_mapgen_get(self._h, C.c_int(2), None, C.c_size_t(0), C.byref(n)))
        out = np.empty((max(n.value, 1), 4), np.float32)
        self._check(lib().erasor_hip_mapgen_save(self._h, _p(out), C.c_size_t(len(out)), C.byref(n)))
        return out[: n.value].copy()

    def count_static_dynamic(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(lib().erasor_hip_count_static_dynamic(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- measurement --
    def profiling(self, on):
        self._check(lib().erasor_hip_profiling(self._h, C.c_int(int(on))))

    def profile_reset(self):
        self._check(lib().erasor_hip_profile_reset(self._h))

    def profile_get(self):
        n = C.c_size_t(0)
        self._check(lib().erasor_hip_profile_get(self._h, None, None, None, C.c_size_t(0), C.byref(n)))
        k = n.value
        names = (C.c_char_p * max(k, 1))()
        ms = (C.c_double * max(k, 1))()
        cnt = (C.c_uint64 * max(k, 1))()
        self._check(lib().erasor_hip_profile_get(self._h, names, ms, cnt, C.c_size_t(k), C.byref(n)))
        return {names[i].decode(): (ms[i], int(cnt[i])) for i in range(k)}

    def voi_split_bytes(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(lib().erasor_hip_voi_split_bytes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def ahead_split_counts(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(lib().erasor_hip_ahead_split_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def overlap_counts(self):
        """(steps whose front was launched beside the previous step's per-bin launch, steps that took it)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(lib().erasor_hip_overlap_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def overlap_auto(self):
        """(mode the next step will use: 1 overlapped / 0 plain, period measured plain, period measured overlapped [us; 0: not yet])"""
        m, a, b = C.c_int(0), C.c_double(0), C.c_double(0)
        self._check(lib().erasor_hip_overlap_auto(self._h, C.byref(m), C.byref(a), C.byref(b)))
        return m.value, a.value, b.value

    def chain_batch(self, n_scans, lead=3):
        """the query chains of `n_scans` announced nodes share one set of launches (erasor_hip_chain_batch); 1: every chain on its own"""
        self._check(lib().erasor_hip_chain_batch(self._h, C.c_int(n_scans), C.c_int(lead)))

    def chain_batch_counts(self):
        """(sets of shared launches made so far, chains that went into them)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(lib().erasor_hip_chain_batch_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def chain_timing(self, reset=False):
        """(average span of the main stream's chain per step, average time between a step's end and the next chunk scan, steps, average
        period chunk scan -> chunk scan) -- on the device's own clock (erasor_hip_chain_timing)."""
        a, b, p, n = C.c_double(0), C.c_double(0), C.c_double(0), C.c_uint64(0)
        self._check(lib().erasor_hip_chain_timing(self._h, C.byref(a), C.byref(b), C.byref(p), C.byref(n), C.c_int(1 if reset else 0)))
        return a.value, b.value, n.value, p.value

    def stream(self):
        return lib().erasor_hip_stream(self._h)

    def device_array(self, a):
        """a host array copied into a device buffer of the handle's device (erasor_hip_device_alloc / _upload); returns the
        device pointer.  Freed with device_free (or with the process)."""
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        self._check(lib().erasor_hip_device_alloc(self._h, C.c_size_t(a.nbytes), C.byref(p)))
        self._check(lib().erasor_hip_device_upload(self._h, p, _p(a), C.c_size_t(a.nbytes)))
        return p.value

    def device_free(self, ptr):
        self._check(lib().erasor_hip_device_free(self._h, C.c_void_p(ptr)))
