// ref_driver.cpp — C API around the reference's OWN objects, for oracle/_ref/liberasor_ref.so.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Built by oracle/ref.mk, only where /root/reference exists.
// This file contains no algorithm: it constructs the reference's erasor::OfflineMapUpdater / ERASOR / mapgen
// (compiled UNMODIFIED from /root/reference/src/... against oracle/stubs/), feeds them through the same
// doors ROS would (rosparam registry, an erasor::node message dispatched to the subscribed callback, the
// in-memory pcl::io registry) and copies their state out.  Compiled with -fno-access-control so it can read
// the private members a ROS deployment only sees through RViz topics (map_arranged_, map_voi_, ...).
//
// One updater per loaded library: the reference keeps function-static state (OMU.cpp:206 stack_count,
// :344 half_size, utils.cpp:88 voxel_filter, mapgen.hpp:248 cnt_voxel).  oracle/ref.py loads a private copy
// of the .so per object so those statics start fresh, exactly like a fresh node process.
#include <csignal>
#include <cstring>

#include "erasor/OfflineMapUpdater.h"  // /root/reference/include (via -I), unmodified
#include "mapgen.hpp"                  // /root/reference/src/mapgen (via -I), unmodified

#include "../include/erasor_hip.h"  // erasor_params / cloud ids: the same structs the oracle and the product use

namespace {

std::string g_err;
std::string g_log;
double g_span_voi = 0, g_span_erasor = 0;

struct NullBuf : std::streambuf {
    int overflow(int c) override { return c; }
};
struct Quiet {  // the reference prints through std::cout on every call
    std::streambuf *old;
    NullBuf nb;
    bool on;
    Quiet() : on(getenv("ERASOR_REF_VERBOSE") == nullptr) {
        if (on) old = std::cout.rdbuf(&nb);
    }
    ~Quiet() {
        if (on) std::cout.rdbuf(old);
    }
};

typedef pcl::PointCloud<pcl::PointXYZI> Cloud;

void to_cloud(const float *xyzi, size_t n, Cloud &c) {
    c.points.resize(n);
    for (size_t k = 0; k < n; ++k) {
        c.points[k].x = xyzi[4 * k], c.points[k].y = xyzi[4 * k + 1], c.points[k].z = xyzi[4 * k + 2];
        c.points[k].intensity = xyzi[4 * k + 3];
    }
    c.width = (uint32_t)n;
    c.height = 1;
}
int from_cloud(const Cloud &c, float *dst, size_t cap, size_t *n) {
    if (n) *n = c.size();
    if (!dst) return 0;
    if (c.size() > cap) return ERASOR_E_CAPACITY;
    for (size_t k = 0; k < c.size(); ++k) {
        dst[4 * k] = c.points[k].x, dst[4 * k + 1] = c.points[k].y, dst[4 * k + 2] = c.points[k].z;
        dst[4 * k + 3] = c.points[k].intensity;
    }
    return 0;
}
geometry_msgs::Pose to_pose(const double p[7]) {
    geometry_msgs::Pose g;
    g.position.x = p[0], g.position.y = p[1], g.position.z = p[2];
    g.orientation.x = p[3], g.orientation.y = p[4], g.orientation.z = p[5], g.orientation.w = p[6];
    return g;
}
void log_line(const std::string &s) {
    // the reference's two timing spans (OMU.cpp:253-257, 264-279)
    const char *k1 = "Extracting VoI takes ", *k2 = "ERASOR takes ";
    size_t p;
    if ((p = s.find(k1)) != std::string::npos) g_span_voi = atof(s.c_str() + p + strlen(k1));
    if ((p = s.find(k2)) != std::string::npos) g_span_erasor = atof(s.c_str() + p + strlen(k2));
    if (getenv("ERASOR_REF_VERBOSE")) fprintf(stderr, "[ref] %s\n", s.c_str());
}

struct Ref {
    erasor::OfflineMapUpdater *omu = nullptr;
    size_t svd_mark = 0;
};
bool g_created = false;
const char *kMapName = "mem://initial_map.pcd";
const char *kSaveDir = "mem://save";

}  // namespace

extern "C" {

const char *ref_last_error() { return g_err.c_str(); }

// rosparam names: erasor.h:47-61, OMU.cpp:66-83,89
void *ref_create(const erasor_params *p, const double lidar2body7[7], const float *map_xyzi, size_t n) {
    if (g_created) {
        g_err = "one updater per loaded library (function-static state in the reference)";
        return nullptr;
    }
    try {
        Quiet q;
        auto &P = ros::stub::params();
        P.clear();
        P["/erasor/max_range"] = p->max_range;
        P["/erasor/num_rings"] = (int)p->num_rings;
        P["/erasor/num_sectors"] = (int)p->num_sectors;
        P["/erasor/max_h"] = p->max_h;
        P["/erasor/min_h"] = p->min_h;
        P["/erasor/th_bin_max_h"] = p->th_bin_max_h;
        P["/erasor/scan_ratio_threshold"] = p->scan_ratio_threshold;
        P["/erasor/num_lowest_pts"] = (int)p->num_lowest_pts;
        P["/erasor/minimum_num_pts"] = (int)p->minimum_num_pts;
        P["/erasor/rejection_ratio"] = p->rejection_ratio;
        P["/erasor/gf_dist_thr"] = p->gf_dist_thr;
        P["/erasor/gf_iter"] = (int)p->gf_iter;
        P["/erasor/gf_num_lpr"] = (int)p->gf_num_lpr;
        P["/erasor/gf_th_seeds_height"] = p->gf_th_seeds_height;
        P["/erasor/map_voxel_size"] = p->map_voxel_size;
        P["/erasor/version"] = (int)p->version;
        P["/MapUpdater/query_voxel_size"] = p->query_voxel_size;
        P["/MapUpdater/map_voxel_size"] = p->map_voxel_size;
        P["/MapUpdater/removal_interval"] = (int)(p->removal_interval > 0 ? p->removal_interval : 1);
        P["/MapUpdater/data_name"] = std::string("ref");
        P["/MapUpdater/env"] = std::string("outdoor");
        P["/MapUpdater/initial_map_path"] = std::string(kMapName);
        P["/MapUpdater/save_path"] = std::string(kSaveDir);
        P["/large_scale/is_large_scale"] = (bool)(p->is_large_scale != 0);
        P["/large_scale/submap_size"] = p->submap_size;
        P["/verbose"] = false;
        if (lidar2body7) P["/tf/lidar2body"] = std::vector<double>(lidar2body7, lidar2body7 + 7);
        Cloud m;
        to_cloud(map_xyzi, n, m);
        pcl::stub::files()[kMapName].points.swap(m.points);
        ros::stub::log_sink() = log_line;
        ros::stub::capture_topics()["/SCDR/debug/polygons_marker"] = true;  // SRT status polygons (erasor.cpp:433,570)
        Ref *r = new Ref();
        r->omu = new erasor::OfflineMapUpdater();  // OMU.cpp:5-32: params, load_global_map, new ERASOR(&nh)
        pcl::stub::files().erase(kMapName);
        // OMU.cpp:78 reads /erasor/max_range for the VoI radius as well; a distinct VoI radius only exists in
        // the C ABI (erasor_params.voi_max_range), so mirror it when the caller set one.
        if (p->voi_max_range > 0) r->omu->max_range_ = p->voi_max_range;
        g_created = true;
        return r;
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

void ref_destroy(void *h) {
    Ref *r = (Ref *)h;
    if (!r) return;
    delete r->omu;
    delete r;
    ros::stub::subscribers().clear();
    g_created = false;
}

// one erasor::node message through the subscribed callback (OMU.cpp:6, 203-330)
int ref_step(void *h, const float *scan, size_t n, const double pose7[7], uint32_t seq) {
    Ref *r = (Ref *)h;
    struct sigaction old;
    sigaction(SIGINT, nullptr, &old);  // callback_node installs a SIGINT handler that exit()s (OMU.cpp:204)
    int rc = 0;
    try {
        Quiet q;
        boost::shared_ptr<erasor::node> msg(new erasor::node());
        msg->header.seq = seq;
        msg->odom = to_pose(pose7);
        msg->lidar.xyzi.assign(scan, scan + 4 * n);
        msg->lidar.width = (uint32_t)n;
        r->svd_mark = Eigen::stub::svd_normal_log().size();
        g_span_voi = g_span_erasor = 0;
        if (!ros::stub::dispatch<erasor::node>("/node/combined/optimized", msg)) {
            g_err = "no subscriber on /node/combined/optimized";
            rc = ERASOR_E_STATE;
        }
    } catch (const std::out_of_range &e) {
        g_err = std::string("std::out_of_range: ") + e.what();
        rc = -100;
    } catch (const std::invalid_argument &e) {
        g_err = std::string("std::invalid_argument: ") + e.what();
        rc = ERASOR_E_UNSUPPORTED;
    } catch (const std::exception &e) {
        g_err = e.what();
        rc = -101;
    }
    sigaction(SIGINT, &old, nullptr);
    return rc;
}

int ref_get_cloud(void *h, int which, float *dst, size_t cap, size_t *n) {
    erasor::OfflineMapUpdater *u = ((Ref *)h)->omu;
    switch (which) {
        case ERASOR_CLOUD_QUERY_VOI: return from_cloud(*u->query_voi_, dst, cap, n);
        case ERASOR_CLOUD_MAP_VOI: return from_cloud(*u->map_voi_, dst, cap, n);
        case ERASOR_CLOUD_STATIC_ESTIMATE: return from_cloud(*u->map_static_estimate_, dst, cap, n);
        case ERASOR_CLOUD_COMPLEMENT: return from_cloud(*u->map_egocentric_complement_, dst, cap, n);
        case ERASOR_CLOUD_MAP_REJECTED: return from_cloud(*u->map_rejected_, dst, cap, n);
        case ERASOR_CLOUD_CURR_REJECTED: return from_cloud(*u->query_rejected_, dst, cap, n);
        case ERASOR_CLOUD_GROUND_VIZ: return from_cloud(u->erasor_->ground_viz, dst, cap, n);
        case ERASOR_CLOUD_MAP: {  // what save_static_map starts from (OMU.cpp:179-184)
            if (!u->is_large_scale_) return from_cloud(*u->map_arranged_, dst, cap, n);
            Cloud m = *u->map_arranged_ + *u->map_arranged_complement_;
            return from_cloud(m, dst, cap, n);
        }
        case 100: return from_cloud(*u->map_outskirts_, dst, cap, n);
        case 101: return from_cloud(*u->map_arranged_, dst, cap, n);  // the submap alone in large-scale mode
        case 102: return from_cloud(*u->total_map_rejected_, dst, cap, n);
    }
    return ERASOR_E_INVALID;
}

// R-POD state: which 0 = r_pod_map, 1 = r_pod_curr, 2 = r_pod_selected (erasor.h:143-145); index ring*S+sector
int ref_get_bins(void *h, int which, uint32_t *count, double *min_h, double *max_h) {
    ERASOR *e = ((Ref *)h)->omu->erasor_.get();
    const R_POD &pod = which == 0 ? e->r_pod_map : which == 1 ? e->r_pod_curr : e->r_pod_selected;
    for (int r = 0; r < e->num_rings; ++r)
        for (int t = 0; t < e->num_sectors; ++t) {
            const Bin &b = pod[r][t];
            const size_t k = (size_t)r * e->num_sectors + t;
            count[k] = (uint32_t)b.points.size();
            min_h[k] = b.min_h;
            max_h[k] = b.max_h;
        }
    return 0;
}
int ref_get_status(void *h, double *status) {
    ERASOR *e = ((Ref *)h)->omu->erasor_.get();
    for (int r = 0; r < e->num_rings; ++r)
        for (int t = 0; t < e->num_sectors; ++t) status[(size_t)r * e->num_sectors + t] = e->r_pod_selected[r][t].status;
    return 0;
}
// least-singular vectors of every estimate_plane_ call of the last step, in call order
// (= reverted bins theta-major x gf_iter), plus the d_ / th_dist_d_ left by the last call (erasor.h:193)
int ref_get_planes(void *h, float *normals, size_t cap_rows, size_t *n_rows, double *last_d, double *last_th) {
    Ref *r = (Ref *)h;
    const auto &log = Eigen::stub::svd_normal_log();
    const size_t nr = log.size() - r->svd_mark;
    if (n_rows) *n_rows = nr;
    if (last_d) *last_d = r->omu->erasor_->d_;
    if (last_th) *last_th = r->omu->erasor_->th_dist_d_;
    if (!normals) return 0;
    if (nr > cap_rows) return ERASOR_E_CAPACITY;
    for (size_t k = 0; k < nr; ++k)
        for (int a = 0; a < 3; ++a) normals[3 * k + a] = log[r->svd_mark + k][a];
    return 0;
}
// likelihood[] of the last published /SCDR/debug/polygons_marker (erasor.cpp:345-433 v2, 496-570 v3): the SRT
// status code per pushed polygon, in the reference's push order (theta-major)
int ref_get_polygon_likelihood(float *dst, size_t cap, size_t *n) {
    auto it = ros::stub::published().find("/SCDR/debug/polygons_marker");
    if (it == ros::stub::published().end()) return ERASOR_E_STATE;
    const auto &pa = std::any_cast<const jsk_recognition_msgs::PolygonArray &>(it->second);
    if (n) *n = pa.likelihood.size();
    if (!dst) return 0;
    if (pa.likelihood.size() > cap || pa.polygons.size() != pa.likelihood.size()) return ERASOR_E_CAPACITY;
    for (size_t k = 0; k < pa.likelihood.size(); ++k) dst[k] = pa.likelihood[k];
    return 0;
}
// vertices of polygon k of that message (set_polygons, erasor.cpp:630-670): returns the vertex count
int ref_get_polygon(size_t k, float *xyz, size_t cap_pts) {
    auto it = ros::stub::published().find("/SCDR/debug/polygons_marker");
    if (it == ros::stub::published().end()) return ERASOR_E_STATE;
    const auto &pa = std::any_cast<const jsk_recognition_msgs::PolygonArray &>(it->second);
    if (k >= pa.polygons.size()) return ERASOR_E_INVALID;
    const auto &pts = pa.polygons[k].polygon.points;
    if (pts.size() > cap_pts) return ERASOR_E_CAPACITY;
    for (size_t i = 0; i < pts.size(); ++i) xyz[3 * i] = pts[i].x, xyz[3 * i + 1] = pts[i].y, xyz[3 * i + 2] = pts[i].z;
    return (int)pts.size();
}
int ref_get_matrices(void *h, float T_l2b[16], float T_b2o[16]) {
    erasor::OfflineMapUpdater *u = ((Ref *)h)->omu;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            T_l2b[i * 4 + j] = u->tf_lidar2body_(i, j);
            T_b2o[i * 4 + j] = u->tf_body2origin_(i, j);
        }
    return 0;
}
// parse_dynamic_obj results of the last step (OMU.cpp:294): [n_static, n_dynamic]
void ref_get_label_counts(void *h, uint64_t out[2]) {
    erasor::OfflineMapUpdater *u = ((Ref *)h)->omu;
    out[0] = u->static_objs_to_viz_->points.size();
    out[1] = u->dynamic_objs_to_viz_->points.size();
}
// the reference's own wall-clock spans of the last step, seconds (OMU.cpp:253-257 / 264-279)
void ref_get_spans(double out[2]) {
    out[0] = g_span_voi;
    out[1] = g_span_erasor;
}
int ref_is_dynamic_obj_close(void *h, int r_target, int theta_target, int r_range, int theta_range) {
    ERASOR *e = ((Ref *)h)->omu->erasor_.get();
    return e->is_dynamic_obj_close(e->r_pod_selected, r_target, theta_target, r_range, theta_range) ? 1 : 0;
}
// OfflineMapUpdater::save_static_map (OMU.cpp:174-196); the "file" lands in the in-memory registry
int ref_save_static_map(void *h, float voxel_size, float *dst, size_t cap, size_t *n) {
    try {
        Quiet q;
        ((Ref *)h)->omu->save_static_map(voxel_size);
    } catch (const std::exception &e) {
        g_err = e.what();
        return -101;
    }
    auto it = pcl::stub::files().find(std::string(kSaveDir) + "/ref_result.pcd");
    if (it == pcl::stub::files().end()) return ERASOR_E_STATE;
    return from_cloud(it->second, dst, cap, n);
}

// ---- class ERASOR called directly on egocentric clouds (erasor.h:109-147), as OMU.cpp:266-272 does ----
// returns -100 when the reference throws std::out_of_range (vector::at, erasor.cpp:112,136)
int ref_erasor_run(void *h, const float *map_voi, size_t nm, const float *query_voi, size_t nq, int version) {
    ERASOR *e = ((Ref *)h)->omu->erasor_.get();
    try {
        Quiet q;
        Cloud m, s;
        to_cloud(map_voi, nm, m);
        to_cloud(query_voi, nq, s);
        ((Ref *)h)->svd_mark = Eigen::stub::svd_normal_log().size();
        e->set_inputs(m, s);
        if (version == 2)
            e->compare_vois_and_revert_ground(0);
        else if (version == 3)
            e->compare_vois_and_revert_ground_w_block(0);
        return 0;
    } catch (const std::out_of_range &ex) {
        g_err = std::string("std::out_of_range: ") + ex.what();
        return -100;
    } catch (const std::exception &ex) {
        g_err = ex.what();
        return -101;
    }
}
// which: 0 static estimate (get_static_estimate arranged), 1 complement, 2 map_rejected, 3 curr_rejected (get_outliers)
int ref_erasor_get(void *h, int which, float *dst, size_t cap, size_t *n) {
    ERASOR *e = ((Ref *)h)->omu->erasor_.get();
    Cloud a, b;
    if (which < 2)
        e->get_static_estimate(a, b);
    else
        e->get_outliers(a, b);
    return from_cloud((which & 1) ? b : a, dst, cap, n);
}
double ref_erasor_get_max_range(void *h) { return ((Ref *)h)->omu->erasor_->get_max_range(); }

// ---- erasor_utils free functions (utils.cpp) ----
int ref_voxelize_preserving_labels(const float *src, size_t n, double leaf, float *dst, size_t cap, size_t *n_out) {
    Cloud::Ptr in(new Cloud());
    to_cloud(src, n, *in);
    Cloud out;
    erasor_utils::voxelize_preserving_labels(in, out, leaf);
    return from_cloud(out, dst, cap, n_out);
}
void ref_geopose2eigen(const double pose7[7], float T[16]) {
    Eigen::Matrix4f m = erasor_utils::geoPose2eigen(to_pose(pose7));
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) T[i * 4 + j] = m(i, j);
}
void ref_eigen2geopose(const float T[16], double pose7[7]) {
    Eigen::Matrix4f m;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) m(i, j) = T[i * 4 + j];
    geometry_msgs::Pose g = erasor_utils::eigen2geoPose(m);
    pose7[0] = g.position.x, pose7[1] = g.position.y, pose7[2] = g.position.z;
    pose7[3] = g.orientation.x, pose7[4] = g.orientation.y, pose7[5] = g.orientation.z, pose7[6] = g.orientation.w;
}
void ref_count_stat_dyn(const float *src, size_t n, int *num_static, int *num_dynamic) {
    Cloud in;
    to_cloud(src, n, in);
    erasor_utils::count_stat_dyn(in, *num_static, *num_dynamic);
}
int ref_parse_dynamic_obj(const float *src, size_t n, float *dyn, float *stat, size_t *n_dyn, size_t *n_stat) {
    Cloud in, d, s;
    to_cloud(src, n, in);
    erasor_utils::parse_dynamic_obj(in, d, s);
    from_cloud(d, dyn, n, n_dyn);
    from_cloud(s, stat, n, n_stat);
    return 0;
}

// ---- class mapgen (src/mapgen/mapgen.hpp) ----
void *ref_mapgen_create(float leafsize, int is_large_scale) {
    Quiet q;
    mapgen *m = new mapgen();
    m->setValue("mem://mapgen", leafsize, "00", "0", "1", 1, is_large_scale != 0);
    return m;
}
void ref_mapgen_destroy(void *h) { delete (mapgen *)h; }
int ref_mapgen_accum(void *h, const float *scan, size_t n, const double pose7[7], size_t *n_curr) {
    try {
        Quiet q;
        erasor::node msg;
        msg.odom = to_pose(pose7);
        msg.lidar.xyzi.assign(scan, scan + 4 * n);
        nav_msgs::Path path;
        ((mapgen *)h)->accumPointCloud(msg, path);
        if (n_curr) *n_curr = ((mapgen *)h)->cloud_curr.size();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -101;
    }
}
// which 0 = cloud_curr, 1 = cloud_map (getPointClouds, mapgen.hpp:264-268)
int ref_mapgen_get(void *h, int which, float *dst, size_t cap, size_t *n) {
    Cloud::Ptr a(new Cloud()), b(new Cloud());
    ((mapgen *)h)->getPointClouds(a, b);
    return from_cloud(which == 0 ? *b : *a, dst, cap, n);
}
// saveNaiveMap (mapgen.hpp:270-305): which 0 = the dense accumulation, 1 = the voxelised map
int ref_mapgen_save(void *h) {
    try {
        Quiet q;
        ((mapgen *)h)->saveNaiveMap("mem://mapgen/original.pcd", "mem://mapgen/map.pcd");
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -101;
    }
}
int ref_mapgen_saved(int which, float *dst, size_t cap, size_t *n) {
    auto it = pcl::stub::files().find(which == 0 ? "mem://mapgen/original.pcd" : "mem://mapgen/map.pcd");
    if (it == pcl::stub::files().end()) return ERASOR_E_STATE;
    return from_cloud(it->second, dst, cap, n);
}

}  // extern "C"
