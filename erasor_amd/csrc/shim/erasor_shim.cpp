// erasor_shim.cpp — see erasor_shim.h.  Host-side glue only; every cloud operation goes through the C ABI of
// liberasor_hip.so (HIP kernels).  Errors of the C ABI are re-raised as the exceptions the reference throws.
#include "erasor_shim.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <sstream>

namespace {
typedef pcl::PointCloud<pcl::PointXYZI> Cloud;

// (into a caller-owned buffer: a 2 MB vector allocated per node is an mmap and ~500 page faults each time)
static void to_xyzi_into(const Cloud &c, std::vector<float> &v) {
    v.resize(c.size() * 4);
    for (size_t i = 0; i < c.size(); ++i) {
        v[4 * i] = c.points[i].x;
        v[4 * i + 1] = c.points[i].y;
        v[4 * i + 2] = c.points[i].z;
        v[4 * i + 3] = c.points[i].intensity;
    }
}
std::vector<float> to_xyzi(const Cloud &c) {
    std::vector<float> v;
    to_xyzi_into(c, v);
    return v;
}
void from_xyzi(const std::vector<float> &v, size_t n, Cloud &c) {
    c.points.resize(n);
    for (size_t i = 0; i < n; ++i) {
        c.points[i].x = v[4 * i];
        c.points[i].y = v[4 * i + 1];
        c.points[i].z = v[4 * i + 2];
        c.points[i].intensity = v[4 * i + 3];
    }
    c.width = (unsigned)n;
    c.height = 1;
}
void check(erasor_hip_handle *h, int rc, const char *what) {
    if (rc == ERASOR_OK) return;
    std::string msg = std::string(what) + ": " + (h ? erasor_hip_last_error(h) : "") + " (rc=" + std::to_string(rc) + ")";
    if (rc == ERASOR_E_UNSUPPORTED || rc == ERASOR_E_INVALID) throw std::invalid_argument(msg);  // as OMU.cpp:125,149,274,312
    throw std::runtime_error(msg);
}
// n_known: the cloud's size, when the step's result block has said it already (one call less)
void fetch_cloud(erasor_hip_handle *h, int which, Cloud &dst, long n_known = -1) {
    size_t n = n_known >= 0 ? (size_t)n_known : 0;
    if (n_known < 0) check(h, erasor_hip_get_cloud(h, which, nullptr, 0, &n), "erasor_hip_get_cloud");
    std::vector<float> v(n * 4 + 4);
    check(h, erasor_hip_get_cloud(h, which, v.data(), n, &n), "erasor_hip_get_cloud");
    from_xyzi(v, n, dst);
}
void mat16(const Eigen::Matrix4f &T, float out[16]) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[r * 4 + c] = T(r, c);
}
bool is_dynamic_label(float intensity) {  // utils.cpp:3,64-65
    const uint32_t sem = static_cast<uint32_t>(intensity) & 0xFFFF;
    return sem >= 252 && sem <= 259;
}
}  // namespace

// ------------------------------------------------------------------------------------------------
namespace erasor_utils {

Eigen::Matrix4f geoPose2eigen(const geometry_msgs::Pose &g) {
    const double x = g.orientation.x, y = g.orientation.y, z = g.orientation.z, w = g.orientation.w;
    const double d = x * x + y * y + z * z + w * w;
    const double s = 2.0 / d;
    const double xs = x * s, ys = y * s, zs = z * s;
    const double wx = w * xs, wy = w * ys, wz = w * zs;
    const double xx = x * xs, xy = x * ys, xz = x * zs;
    const double yy = y * ys, yz = y * zs, zz = z * zs;
    Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
    T(0, 0) = (float)(1.0 - (yy + zz)); T(0, 1) = (float)(xy - wz); T(0, 2) = (float)(xz + wy);
    T(1, 0) = (float)(xy + wz); T(1, 1) = (float)(1.0 - (xx + zz)); T(1, 2) = (float)(yz - wx);
    T(2, 0) = (float)(xz - wy); T(2, 1) = (float)(yz + wx); T(2, 2) = (float)(1.0 - (xx + yy));
    T(0, 3) = (float)g.position.x; T(1, 3) = (float)g.position.y; T(2, 3) = (float)g.position.z;
    T(3, 0) = T(3, 1) = T(3, 2) = 0.f; T(3, 3) = 1.f;
    return T;
}

Eigen::Matrix4f inverse(const Eigen::Matrix4f &Tf) {
    double m[16], inv[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) m[r * 4 + c] = Tf(r, c);
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0) throw std::invalid_argument("singular transform");
    det = 1.0 / det;
    Eigen::Matrix4f R;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) R(r, c) = (float)(inv[r * 4 + c] * det);
    return R;
}

void parse_dynamic_obj(const Cloud &cloudIn, Cloud &dynamicOut, Cloud &staticOut) {
    dynamicOut.points.clear();
    staticOut.points.clear();
    for (const auto &pt : cloudIn.points) (is_dynamic_label(pt.intensity) ? dynamicOut : staticOut).points.push_back(pt);
}
void count_stat_dyn(const Cloud &cloudIn, int &num_static, int &num_dynamic) {
    int s = 0, d = 0;
    for (const auto &pt : cloudIn.points) (is_dynamic_label(pt.intensity) ? d : s)++;
    num_static = s;
    num_dynamic = d;
}

int &voxelize_device() {
    static int dev = 0;
    return dev;
}
void voxelize_preserving_labels(Cloud::Ptr src, Cloud &dst, double leaf_size) {
    static erasor_hip_handle *h = nullptr;  // process-wide, created on first use (like the reference's function-static VoxelGrid, utils.cpp:88)
    static int h_dev = -1;
    if (!h || h_dev != voxelize_device()) {
        if (h) erasor_hip_destroy(h);
        h = nullptr;
        erasor_params p;
        erasor_hip_params_default(&p);
        check(nullptr, erasor_hip_create(&p, voxelize_device(), &h), "erasor_hip_create");
        h_dev = voxelize_device();
    }
    if (!src) throw std::invalid_argument("voxelize_preserving_labels: null input cloud");
    const std::vector<float> v = to_xyzi(*src);
    std::vector<float> out(v.size() + 4);
    size_t n = 0;
    check(h, erasor_hip_voxelize_preserving_labels(h, v.data(), src->size(), leaf_size, out.data(), src->size(), &n), "voxelize_preserving_labels");
    from_xyzi(out, n, dst);
}

}  // namespace erasor_utils

// ------------------------------------------------------------------------------------------------
ERASOR::ERASOR(const erasor_params &p, int device) : P_(p), device_(device) {}
#ifdef ERASOR_SHIM_WITH_ROS
static erasor_params params_from_rosparam(ros::NodeHandle &nh) {  // erasor.h:47-61, OMU.cpp:66-83
    erasor_params p;
    erasor_hip_params_default(&p);
    int i;
    nh.param("/erasor/max_range", p.max_range, 10.0);
    nh.param("/erasor/num_rings", i, 20); p.num_rings = i;
    nh.param("/erasor/num_sectors", i, 60); p.num_sectors = i;
    nh.param("/erasor/max_h", p.max_h, 3.0);
    nh.param("/erasor/min_h", p.min_h, 0.0);
    nh.param("/erasor/th_bin_max_h", p.th_bin_max_h, 0.39);
    nh.param("/erasor/scan_ratio_threshold", p.scan_ratio_threshold, 0.22);
    nh.param("/erasor/num_lowest_pts", i, 5); p.num_lowest_pts = i;
    nh.param("/erasor/minimum_num_pts", i, 4); p.minimum_num_pts = i;
    nh.param("/erasor/rejection_ratio", p.rejection_ratio, 0.33);
    nh.param("/erasor/gf_dist_thr", p.gf_dist_thr, 0.05);
    nh.param("/erasor/gf_iter", i, 3); p.gf_iter = i;
    nh.param("/erasor/gf_num_lpr", i, 10); p.gf_num_lpr = i;
    nh.param("/erasor/gf_th_seeds_height", p.gf_th_seeds_height, 0.5);
    nh.param("/erasor/map_voxel_size", p.map_voxel_size, 0.2);
    nh.param("/erasor/version", i, 3); p.version = i;
    return p;
}
ERASOR::ERASOR(ros::NodeHandle *nodehandler) : P_(params_from_rosparam(*nodehandler)), device_(0) {}
#endif
ERASOR::~ERASOR() {
    for (auto *h : h_)
        if (h) erasor_hip_destroy(h);
}
void ERASOR::set_inputs(const Cloud &map_voi, const Cloud &query_voi) {
    debug_curr_rejected.clear();
    debug_map_rejected.clear();
    map_complement.clear();
    map_voi_ = map_voi;
    query_voi_ = query_voi;
}
void ERASOR::run(int version) {
    erasor_hip_handle *&h = h_[version == 2 ? 0 : 1];
    if (!h) {
        erasor_params p = P_;
        p.version = version;
        check(nullptr, erasor_hip_create(&p, device_, &h), "erasor_hip_create");
    }
    const std::vector<float> m = to_xyzi(map_voi_), q = to_xyzi(query_voi_);
    erasor_step_result res;
    check(h, erasor_hip_erasor_run(h, m.data(), map_voi_.size(), q.data(), query_voi_.size(), &res), "erasor_hip_erasor_run");
    fetch_cloud(h, ERASOR_CLOUD_GROUND_VIZ, ground_viz);
    fetch_cloud(h, ERASOR_CLOUD_MAP_REJECTED, debug_map_rejected);   // identity transforms: egocentric, as in the reference
    fetch_cloud(h, ERASOR_CLOUD_CURR_REJECTED, debug_curr_rejected);
    fetch_cloud(h, ERASOR_CLOUD_COMPLEMENT, map_complement);
    fetch_cloud(h, ERASOR_CLOUD_STATIC_ESTIMATE, arranged_);
    status.assign((size_t)P_.num_rings * P_.num_sectors, 0.0);
    check(h, erasor_hip_get_status(h, status.data()), "erasor_hip_get_status");
    n_ambiguous = res.n_ambiguous;
    n_neg_sector = res.n_neg_sector;
    n_degenerate_plane = res.n_degenerate_plane;
    if (res.n_ambiguous)
        fprintf(stderr, "[erasor shim] %u point(s) within 1e-11 of a sector boundary: their bin is not provably the reference's (device atan2 vs glibc)\n",
                res.n_ambiguous);
    last_run_ = version == 2 ? 0 : 1;
    if (keep_rpods) fetch_rpods();
}
void ERASOR::fetch_rpods() {
    // the public R-PODs (erasor.h:143-145)
    erasor_hip_handle *h = last_run_ >= 0 ? h_[last_run_] : nullptr;
    if (!h) return;
    const size_t B = (size_t)P_.num_rings * P_.num_sectors;
    std::vector<uint32_t> begin(B), count(B);
    R_POD *pods[3] = {&r_pod_map, &r_pod_curr, &r_pod_selected};
    for (int which = 0; which < 3; ++which) {
        size_t n = 0;
        check(h, erasor_hip_get_rpod(h, which, nullptr, 0, &n, begin.data(), count.data()), "erasor_hip_get_rpod");
        std::vector<float> v(n * 4 + 4);
        check(h, erasor_hip_get_rpod(h, which, v.data(), n, &n, begin.data(), count.data()), "erasor_hip_get_rpod");
        R_POD &pod = *pods[which];
        pod.assign((size_t)P_.num_rings, Ring((size_t)P_.num_sectors));
        for (int r = 0; r < P_.num_rings; ++r)
            for (int t = 0; t < P_.num_sectors; ++t) {
                Bin &b = pod[r][t];
                const size_t i = (size_t)r * P_.num_sectors + t;
                b.max_h = -10000000000000.0;  // clear_bin, erasor.cpp:44-52 (INF = 1e13, erasor.h:3)
                b.min_h = 10000000000000.0;
                b.x = b.y = 0;
                b.status = which == 2 ? status[i] : 0.0;
                b.is_occupied = count[i] > 0;
                b.points.points.resize(count[i]);
                for (uint32_t k = 0; k < count[i]; ++k) {  // pt2r_pod, erasor.cpp:87-98
                    pcl::PointXYZI &pt = b.points.points[k];
                    const float *q = &v[4 * ((size_t)begin[i] + k)];
                    pt.x = q[0]; pt.y = q[1]; pt.z = q[2]; pt.intensity = q[3];
                    if (pt.z >= b.max_h) { b.max_h = pt.z; b.x = pt.x; b.y = pt.y; }
                    if (pt.z <= b.min_h) b.min_h = pt.z;
                }
                b.points.width = count[i];
                b.points.height = 1;
            }
    }
}
bool ERASOR::is_dynamic_obj_close(R_POD &r_pod, int r_target, int theta_target, int r_range, int theta_range) {
    const int num_rings = P_.num_rings, num_sectors = P_.num_sectors;
    std::vector<int> theta_candidates;
    for (int j = theta_target - theta_range; j <= theta_target + theta_range; j++) {
        if (j < 0) theta_candidates.push_back(j + num_rings);            // (sic: the reference wraps by num_rings, erasor.cpp:578)
        else if (j >= num_sectors) theta_candidates.push_back(j - num_rings);
        else theta_candidates.push_back(j);
    }
    for (int r = std::max(0, r_target - r_range); r <= std::min(r_target + r_range, num_rings - 1); r++)
        for (int theta : theta_candidates) {
            if ((r == r_target) && (theta == theta_target)) continue;
            if (theta < 0 || theta >= num_sectors) continue;  // the reference reads out of bounds here (num_rings > num_sectors only)
            if (r_pod[r][theta].status == ERASOR_ST_CURR_IS_HIGHER) return true;
        }
    return false;
}
void ERASOR::compare_vois_and_revert_ground(int) { run(2); }
void ERASOR::compare_vois_and_revert_ground_w_block(int) { run(3); }
void ERASOR::get_static_estimate(Cloud &arranged, Cloud &complement) {
    arranged = arranged_;  // r_pod2pc(selected) + ground_viz (erasor.cpp:615-616)
    complement = map_complement;
}
void ERASOR::get_outliers(Cloud &map_rejected, Cloud &curr_rejected) {
    map_rejected = debug_map_rejected;
    curr_rejected = debug_curr_rejected;
}
double ERASOR::get_max_range() { return P_.max_range; }

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
mapgen::mapgen(int device) {
    erasor_params p;
    erasor_hip_params_default(&p);
    check(nullptr, erasor_hip_create(&p, device, &h_), "erasor_hip_create");
}
mapgen::~mapgen() {
    if (h_) erasor_hip_destroy(h_);
}
void mapgen::setValue(std::string pcd_save_path, float voxelsize, std::string sequence, std::string init_time_stamp,
                      std::string final_time_stamp, int frame_interval, bool is_map_large_scale) {
    save_path_ = pcd_save_path;
    leafsize_ = voxelsize;
    seq_ = sequence;
    init_stamp_ = init_time_stamp;
    final_stamp_ = final_time_stamp;
    interval_ = frame_interval;
    (void)std::stoi(final_stamp_);  // mapgen.hpp:190-191 (throws on a malformed bag name, like the reference)
    (void)std::stoi(init_stamp_);
    is_large_scale_ = is_map_large_scale;
    check(h_, erasor_hip_mapgen_begin(h_, (double)leafsize_, is_large_scale_ ? 1 : 0), "erasor_hip_mapgen_begin");
}
void mapgen::accumPointCloud(const geometry_msgs::Pose &odom, const Cloud &lidar) {
    float Tp[16];
    mat16(erasor_utils::geoPose2eigen(odom), Tp);  // mapgen.hpp:234
    const std::vector<float> s = to_xyzi(lidar);
    size_t n = 0;
    check(h_, erasor_hip_mapgen_accum(h_, s.data(), lidar.size(), Tp, nullptr, &n), "erasor_hip_mapgen_accum");
}
static void mapgen_fetch(erasor_hip_handle *h, int which, Cloud &dst) {
    size_t n = 0;
    check(h, erasor_hip_mapgen_get(h, which, nullptr, 0, &n), "erasor_hip_mapgen_get");
    std::vector<float> v(n * 4 + 4);
    check(h, erasor_hip_mapgen_get(h, which, v.data(), n, &n), "erasor_hip_mapgen_get");
    from_xyzi(v, n, dst);
}
void mapgen::getPointClouds(Cloud &map_out, Cloud &curr_out) {
    mapgen_fetch(h_, 1, map_out);
    mapgen_fetch(h_, 0, curr_out);
}
void mapgen::saveNaiveMap(const std::string &original_dir, const std::string &map_dir) {
    Cloud cloud_src, cloud_out;
    mapgen_fetch(h_, 2, cloud_src);
    if (erasor_utils::save_pcd_ascii(original_dir, cloud_src) != 0) throw std::runtime_error("cannot write " + original_dir);  // :281
    size_t n = cloud_src.size();
    std::vector<float> v(n * 4 + 4);
    check(h_, erasor_hip_mapgen_save(h_, v.data(), n, &n), "erasor_hip_mapgen_save");
    from_xyzi(v, n, cloud_out);
    if (erasor_utils::save_pcd_ascii(map_dir, cloud_out) != 0) throw std::runtime_error("cannot write " + map_dir);  // :299
}
std::string mapgen::map_file_name() const {
    return save_path_ + "/" + seq_ + "_" + init_stamp_ + "_to_" + final_stamp_ + "_w_interval" + std::to_string(interval_) + "_voxel_" +
           std::to_string(leafsize_) + ".pcd";
}

namespace erasor {

OfflineMapUpdater::OfflineMapUpdater(const Config &cfg) : cfg_(cfg) {
    if (cfg_.environment == "indoor") throw std::invalid_argument("This `indoor` mode is not perfect!");  // OMU.cpp:149
    cfg_.params.is_large_scale = cfg_.is_large_scale ? 1 : 0;  // /large_scale/is_large_scale, /large_scale/submap_size (OMU.cpp:75-76)
    if (cfg_.is_large_scale) cfg_.params.submap_size = cfg_.submap_size;
    geometry_msgs::Pose l2b;
    l2b.position.x = cfg_.lidar2body[0]; l2b.position.y = cfg_.lidar2body[1]; l2b.position.z = cfg_.lidar2body[2];
    l2b.orientation.x = cfg_.lidar2body[3]; l2b.orientation.y = cfg_.lidar2body[4]; l2b.orientation.z = cfg_.lidar2body[5];
    l2b.orientation.w = cfg_.lidar2body[6];
    tf_lidar2body_ = erasor_utils::geoPose2eigen(l2b);  // OMU.cpp:100
    // fetch_VoI's radius is /erasor/max_range read with another default (OMU.cpp:78); same value when it comes from the YAML
    int rc = erasor_hip_create(&cfg_.params, cfg_.device, &h_);
    if (rc == ERASOR_E_UNSUPPORTED) throw std::invalid_argument("Other version is not implemented!");  // OMU.cpp:274
    check(h_, rc, "erasor_hip_create");
    memset(&last, 0, sizeof(last));
    if (!cfg_.initial_map_path.empty()) {
        Cloud m;
        if (erasor_utils::load_pcd(cfg_.initial_map_path, m) == -1) throw std::invalid_argument("Maybe intiial map path is not correct!");  // OMU.cpp:125
        set_global_map(m);
    }
}
OfflineMapUpdater::~OfflineMapUpdater() {
    if (h_) erasor_hip_destroy(h_);
}
void OfflineMapUpdater::set_global_map(const Cloud &map_init) {
    const std::vector<float> v = to_xyzi(map_init);
    check(h_, erasor_hip_set_map(h_, v.data(), map_init.size()), "erasor_hip_set_map");
}
// pcl::PointXYZI records go to the library as they lie (erasor_hip_step_rows / _prefetch_node_rows): x, y, z at byte 0, intensity at
// its own offset; the pass that stages a host scan in pinned memory repacks them (rounds 2-3 rewrote 4 MB per node into XYZI rows first)
static constexpr size_t kRowStride = sizeof(pcl::PointXYZI);
static const size_t kRowIntensity = offsetof(pcl::PointXYZI, intensity);
void OfflineMapUpdater::callback_node(int seq, const geometry_msgs::Pose &odom, const Cloud &lidar) { callback_node(seq, odom, lidar, 0); }
void OfflineMapUpdater::announce_next_deferred(int seq, const Cloud &lidar, const geometry_msgs::Pose &odom) {
    def_cloud_ = &lidar;
    def_odom_ = odom;
    def_seq_ = seq;
}
// the deferred announcement, made now -- inside a callback_node (stack_count_ counts that callback already), while its step runs on the GPU.
// It is about the node behind the last one announced (with one node ahead: the next callback)
void OfflineMapUpdater::stage_deferred() {
    if (!def_cloud_) return;
    const Cloud &lidar = *def_cloud_;
    def_cloud_ = nullptr;
    const int cb_no = std::max(ahead_upto_, stack_count_) + 1;
    if (cb_no % cfg_.params.removal_interval != 0) {  // gated out (OMU.cpp:206-209)
        ahead_upto_ = cb_no;
        return;
    }
    if (n_ahead_ >= max_ahead_) return;  // as many ahead as the updater takes
    float Tl[16], Tb[16];
    mat16(tf_lidar2body_, Tl);
    mat16(erasor_utils::geoPose2eigen(def_odom_), Tb);
    uint64_t t = 0;
    check(h_, erasor_hip_prefetch_node_rows(h_, lidar.points.data(), lidar.size(), kRowStride, kRowIntensity, Tl, Tb, &t), "erasor_hip_prefetch_node_rows");
    {
        float To[16];
        mat16(erasor_utils::inverse(erasor_utils::geoPose2eigen(def_odom_)), To);
        check(h_, erasor_hip_announce_origin2body(h_, To), "erasor_hip_announce_origin2body");
    }
    ++n_ahead_;
    ahead_upto_ = cb_no;
    auto_tickets_.emplace_back(def_seq_, t);
}
void OfflineMapUpdater::callback_node(int seq, const geometry_msgs::Pose &odom, const Cloud &lidar, uint64_t ticket) {
    // the deferred announcement points at a caller local: whatever way this callback ends, it does not outlive it
    struct ClearDeferred {
        const Cloud *&p;
        ~ClearDeferred() { p = nullptr; }
    } clear_deferred{def_cloud_};
    stack_count_++;
    for (auto it = auto_tickets_.begin(); it != auto_tickets_.end(); ++it)  // announced through announce_next_deferred
        if (it->first == seq) {
            if (!ticket) ticket = it->second;
            auto_tickets_.erase(it);
            break;
        }
    if (stack_count_ % cfg_.params.removal_interval != 0) {  // OMU.cpp:206-209,327-329 "PASS!"
        if (cfg_.verbose) printf(" PASS! \n");
        stage_deferred();
    } else {
        if (cfg_.environment != "outdoor") throw std::invalid_argument("Other modes are not supported");  // OMU.cpp:312
        tf_body2origin_ = erasor_utils::geoPose2eigen(odom);  // OMU.cpp:219
        const Eigen::Matrix4f tf_origin2body = erasor_utils::inverse(tf_body2origin_);
        float Tl[16], Tb[16], To[16];
        mat16(tf_lidar2body_, Tl);
        mat16(tf_body2origin_, Tb);
        mat16(tf_origin2body, To);
        // an announced node, by ticket: the step runs on the copy that was staged when it was announced.  Otherwise the records go in as
        // they lie; if this very cloud was announced the library recognises it (pointer, size, a hash of every record) and anything
        // else is a fresh scan (the handle then drops what was announced)
        if (ticket) {
            // (in two halves: a deferred announcement of the node behind this one is staged while this step runs on the GPU)
            check(h_, erasor_hip_step_ticket_async(h_, ticket, Tb, To), "erasor_hip_step_ticket_async");
            if (n_ahead_ > 0) --n_ahead_;
            try {
                stage_deferred();
            } catch (...) {
                // the step in flight is collected whatever the announcement did: a handle with an uncollected step takes no other call
                (void)erasor_hip_step_wait(h_, &last);
                throw;
            }
            check(h_, erasor_hip_step_wait(h_, &last), "erasor_hip_step_wait");
        } else {
            // (deep announcements are consumed by ticket: a cloud that comes without one is a fresh scan, and the library drops every
            // announcement when it meets one -- erasor_hip_step_rows recognises at most the OLDEST by content; with one node ahead that is
            // the old behaviour)
            check(h_, erasor_hip_step_rows(h_, lidar.points.data(), lidar.size(), kRowStride, kRowIntensity, Tl, Tb, To, &last), "erasor_hip_step_rows");
            if (n_ahead_ > 1) {  // the step consumed the oldest or dropped all: make it "all" (a standalone call drops what is left)
                check(h_, erasor_hip_drop_announced(h_), "erasor_hip_drop_announced");
            }
            n_ahead_ = 0;  // (whatever was announced has been consumed or dropped by this step)
            ahead_upto_ = stack_count_;
            auto_tickets_.clear();
            stage_deferred();
        }
        if (last.n_ambiguous)  // (never seen on transformed clouds; said aloud because bin equality is only PROVABLE when it is zero)
            fprintf(stderr, "[erasor shim] node %d: %u point(s) within 1e-11 of a sector boundary: their bin is not provably the reference's (device atan2 vs glibc)\n",
                    seq, last.n_ambiguous);
        fetch_cloud(h_, ERASOR_CLOUD_MAP_REJECTED, map_rejected, (long)last.n_map_rejected);
        fetch_cloud(h_, ERASOR_CLOUD_CURR_REJECTED, query_rejected, (long)last.n_curr_rejected);
        ++num_processed;
        if (cfg_.verbose) {  // print_status (OMU.cpp:451-465)
            printf("ERASOR Input: %llu = %llu + %llu - %llu\n", (unsigned long long)last.n_voi, (unsigned long long)last.n_static_estimate,
                   (unsigned long long)last.n_complement, (unsigned long long)last.n_map_rejected);
            printf("[Debug] Total: %llu  dynamic %llu  static %llu\n", (unsigned long long)last.n_map_out, (unsigned long long)last.n_dynamic,
                   (unsigned long long)last.n_static);
        }
    }
}
uint64_t OfflineMapUpdater::announce_next(const Cloud &lidar) { return announce(lidar, nullptr); }
uint64_t OfflineMapUpdater::announce_next(const Cloud &lidar, const geometry_msgs::Pose &odom) { return announce(lidar, &odom); }
uint64_t OfflineMapUpdater::announce_upcoming(const Cloud &lidar, const geometry_msgs::Pose &odom) { return announce(lidar, &odom, true); }
uint64_t OfflineMapUpdater::announce(const Cloud &lidar, const geometry_msgs::Pose *odom, bool upcoming) {
    // called BEFORE callback_node(current) with the cloud of the node after it: current is callback number stack_count_ + 1
    // (round 6: up to max_ahead_ nodes, in node order: this call is about the node behind the last one announced or gated)
    if (upcoming && (n_ahead_ > 0 || ahead_upto_ > stack_count_)) return 0;  // (only in front of everything else)
    const int cb_no = upcoming ? stack_count_ + 1 : std::max(ahead_upto_, stack_count_ + 1) + 1;
    if (cb_no % cfg_.params.removal_interval != 0) {  // that node will be gated out (OMU.cpp:206-209): nothing to announce
        ahead_upto_ = cb_no;
        return 0;
    }
    if (n_ahead_ >= max_ahead_) return 0;  // as many nodes ahead as callback_node can honour (one unless set_lookahead said more)
    float Tl[16], Tb[16];
    mat16(tf_lidar2body_, Tl);
    if (odom) mat16(erasor_utils::geoPose2eigen(*odom), Tb);  // OMU.cpp:219: the whole node is known, the next fetch_VoI pass goes ahead too
    uint64_t ticket = 0;
    check(h_, erasor_hip_prefetch_node_rows(h_, lidar.points.data(), lidar.size(), kRowStride, kRowIntensity, Tl, odom ? Tb : nullptr, &ticket),
          "erasor_hip_prefetch_node_rows");
    if (odom) {  // (round 5: ... and the inverse callback_node will compute, OMU.cpp:436: that step may then overlap the one in front)
        float To[16];
        mat16(erasor_utils::inverse(erasor_utils::geoPose2eigen(*odom)), To);
        check(h_, erasor_hip_announce_origin2body(h_, To), "erasor_hip_announce_origin2body");
    }
    ++n_ahead_;
    ahead_upto_ = cb_no;
    return ticket;
}
void OfflineMapUpdater::set_lookahead(int n) { max_ahead_ = std::max(1, std::min(n, 7)); }
void OfflineMapUpdater::get_map(Cloud &dst) {
    size_t n = 0;
    check(h_, erasor_hip_map_size(h_, &n), "erasor_hip_map_size");
    std::vector<float> v(n * 4 + 4);
    check(h_, erasor_hip_get_map(h_, v.data(), n, &n), "erasor_hip_get_map");
    from_xyzi(v, n, dst);
}
void OfflineMapUpdater::save_static_map(float voxel_size) {
    Cloud src;
    get_map(src);  // *ptr_src = *map_arranged_ (OMU.cpp:183)
    const std::vector<float> v = to_xyzi(src);
    std::vector<float> out(v.size() + 4);
    size_t n = 0;
    check(h_, erasor_hip_voxelize_preserving_labels(h_, v.data(), src.size(), voxel_size, out.data(), src.size(), &n), "voxelize_preserving_labels");
    Cloud map_to_be_saved;
    from_xyzi(out, n, map_to_be_saved);
    const std::string target = cfg_.save_path + "/" + cfg_.data_name + "_result.pcd";  // OMU.cpp:191-193
    if (erasor_utils::save_pcd_ascii(target, map_to_be_saved) != 0) throw std::runtime_error("cannot write " + target);
}
}  // namespace erasor
