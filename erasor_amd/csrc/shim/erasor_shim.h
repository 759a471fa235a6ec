// erasor_shim.h — the reference's C++ surface for the hot path, backed by liberasor_hip.so.
//
// Same class / function names, argument meaning and error behaviour as
//   include/erasor/erasor.h:43-147            (class ERASOR)
//   include/erasor/OfflineMapUpdater.h:9-151   (class erasor::OfflineMapUpdater)
//   include/tools/erasor_utils.hpp:50-108      (namespace erasor_utils)
// minus ROS: parameters arrive in an erasor_params struct instead of the rosparam server, the per-node
// message is passed as (seq, odom pose, lidar cloud), debug topics become public clouds.
#ifndef ERASOR_SHIM_H
#define ERASOR_SHIM_H

#include "erasor_shim_queue.h"
#include <memory>
#include <stdexcept>
#include <deque>
#include <string>
#include <vector>

#include "../../../include/erasor_hip.h"
#include "erasor_shim_types.h"
#ifdef ERASOR_SHIM_WITH_ROS
#include <ros/ros.h>
#endif

namespace erasor_utils {
typedef pcl::PointCloud<pcl::PointXYZI> Cloud;
// utils.cpp:35-55 (tf::Matrix3x3(tf::Quaternion) in double, narrowed to float)
Eigen::Matrix4f geoPose2eigen(const geometry_msgs::Pose &geoPose);
// tf_body2origin_.inverse() (OMU.cpp:436): double-precision cofactor inverse, narrowed
Eigen::Matrix4f inverse(const Eigen::Matrix4f &T);
// utils.cpp:57-78
void parse_dynamic_obj(const Cloud &cloudIn, Cloud &dynamicOut, Cloud &staticOut);
// utils.cpp:116-138
void count_stat_dyn(const Cloud &cloudIn, int &num_static, int &num_dynamic);
// utils.cpp:6-33 (tf::Matrix3x3::getRotation in double)
geometry_msgs::Pose eigen2geoPose(const Eigen::Matrix4f &pose);
// utils.hpp:75-93 — .pcd with fields x y z [intensity] (ASCII, binary with any SIZE/TYPE/COUNT layout,
// binary_compressed); returns -1 on failure like the reference
int load_pcd(const std::string &pcd_name, Cloud &dst);
// pcl::io::savePCDFileASCII as used at OMU.cpp:193 (PCL's default: 8 significant digits), and the lossless binary form
int save_pcd_ascii(const std::string &pcd_name, const Cloud &src);
int save_pcd_binary(const std::string &pcd_name, const Cloud &src);
// utils.cpp:80-114, same signature as utils.hpp:103 (called at OMU.cpp:186,238, erasor.cpp:528, mapgen.hpp:239,252,295):
// PCL VoxelGrid + exact 1-NN label, on the GPU (erasor_hip_voxelize_preserving_labels).  The free function has no
// object to hang a device handle on: it uses one process-wide handle on device `voxelize_device()` (default 0).
void voxelize_preserving_labels(Cloud::Ptr src, Cloud &dst, double leaf_size);
int &voxelize_device();
}  // namespace erasor_utils

// struct Bin / R_POD / Ring of the reference (erasor.h:24-41), same member names
struct Bin {
    double max_h;
    double min_h;
    double x;
    double y;
    double status;
    bool is_occupied;
    pcl::PointCloud<pcl::PointXYZI> points;
};
typedef std::vector<std::vector<Bin>> R_POD;
typedef std::vector<Bin> Ring;

// class ERASOR (erasor.h:43-228).  Inputs of set_inputs are egocentric clouds, as in the reference.
class ERASOR {
public:
    explicit ERASOR(const erasor_params &p, int device = 0);
#ifdef ERASOR_SHIM_WITH_ROS
    // the reference's constructor (erasor.h:46-61): parameters from the rosparam server under their reference names
    explicit ERASOR(ros::NodeHandle *nodehandler);
#endif
    ~ERASOR();
    void set_inputs(const pcl::PointCloud<pcl::PointXYZI> &map_voi, const pcl::PointCloud<pcl::PointXYZI> &query_voi);  // erasor.cpp:57-85
    void compare_vois_and_revert_ground(int frame);           // v2, erasor.cpp:332-434
    void compare_vois_and_revert_ground_w_block(int frame);   // v3, erasor.cpp:438-571
    void get_static_estimate(pcl::PointCloud<pcl::PointXYZI> &arranged, pcl::PointCloud<pcl::PointXYZI> &complement);  // :612-626
    void get_outliers(pcl::PointCloud<pcl::PointXYZI> &map_rejected, pcl::PointCloud<pcl::PointXYZI> &curr_rejected);   // :322-327
    double get_max_range();  // :628
    // erasor.cpp:573-595: is a CURR_IS_HIGHER bin within (r_range, theta_range) of the target?  Works on the R-POD it is
    // given (the reference passes r_pod_selected) and keeps the reference's theta wrap by num_rings.
    bool is_dynamic_obj_close(R_POD &r_pod_selected, int r_target, int theta_target, int r_range, int theta_range);
    // erasor.h:143-145, filled by compare_*: r_pod[ring][sector]; Bin::points are the device's bin lists (egocentric),
    // max_h / min_h / is_occupied / status as the reference leaves them (x, y: the bin's highest point, erasor.cpp:91-94)
    R_POD r_pod_map, r_pod_curr, r_pod_selected;
    // Rebuilding the three R-PODs costs six read-backs (each with a stream synchronisation) and a per-point host loop after
    // EVERY compare_* call.  A caller that only wants get_static_estimate / get_outliers sets this to false; fetch_rpods()
    // fills them on demand for the last compare_* call.
    bool keep_rpods = true;
    void fetch_rpods();
    // guard counters of the last compare_* call (erasor_step_result): a non-zero n_ambiguous means a point sat within 1e-11 of a
    // sector boundary, where the device's atan2 may round differently from glibc's -- bin membership of that point is then not
    // provably the reference's
    unsigned n_ambiguous = 0, n_neg_sector = 0, n_degenerate_plane = 0;
    // public members of the reference (erasor.h:127,139-141)
    pcl::PointCloud<pcl::PointXYZI> ground_viz, debug_curr_rejected, debug_map_rejected, map_complement;
    // r_pod_selected[r][theta].status after compare_* (erasor.h:145), index = ring*num_sectors + sector
    std::vector<double> status;

private:
    void run(int version);
    erasor_params P_;
    int device_;
    erasor_hip_handle *h_[2] = {nullptr, nullptr};  // one handle per algorithm version
    int last_run_ = -1;                             // which of the two the last compare_* call used
    pcl::PointCloud<pcl::PointXYZI> map_voi_, query_voi_, arranged_;
};

// class mapgen (src/mapgen/mapgen.hpp:27-305): builds the naive accumulated map that OfflineMapUpdater later loads.
// Same public methods; the erasor::node message is passed as (odom pose, lidar cloud), nav_msgs::Path is dropped.
class mapgen {
public:
    explicit mapgen(int device = 0);
    ~mapgen();
    void setValue(std::string pcd_save_path, float voxelsize, std::string sequence, std::string init_time_stamp, std::string final_time_stamp,
                  int frame_interval, bool is_map_large_scale);                                                   // mapgen.hpp:182-196
    void accumPointCloud(const geometry_msgs::Pose &odom, const pcl::PointCloud<pcl::PointXYZI> &lidar);         // :198-257
    void getPointClouds(pcl::PointCloud<pcl::PointXYZI> &map_out, pcl::PointCloud<pcl::PointXYZI> &curr_out);   // :258-262
    void saveNaiveMap(const std::string &original_dir, const std::string &map_dir);                              // :264-303
    // main.cpp:34-38: <save_path>/<seq>_<from>_to_<to>_w_interval<k>_voxel_<leaf>.pcd
    std::string map_file_name() const;

private:
    erasor_hip_handle *h_ = nullptr;
    float leafsize_ = 0.05f;
    std::string seq_, init_stamp_, final_stamp_, save_path_;
    int interval_ = 1;
    bool is_large_scale_ = false;
};

namespace erasor {
// erasor::OfflineMapUpdater (OfflineMapUpdater.h:9-151): owns the (device-resident) map, one callback per node.
class OfflineMapUpdater {
public:
    struct Config {
        erasor_params params;                                  // /erasor/*, /MapUpdater/* (set_params, OMU.cpp:63-105)
        double lidar2body[7] = {0, 0, 0, 0, 0, 0, 1};          // /tf/lidar2body: x y z qx qy qz qw (OMU.cpp:89-104)
        std::string environment = "outdoor";                   // /MapUpdater/env
        std::string initial_map_path, save_path = ".", data_name = "00";
        bool is_large_scale = false;                           // /large_scale/is_large_scale (OMU.cpp:75)
        double submap_size = 200.0;                            // /large_scale/submap_size   (OMU.cpp:76)
        bool verbose = false;
        int device = 0;
    };
    explicit OfflineMapUpdater(const Config &cfg);            // OMU.cpp:5-32 (+ load_global_map when a path is given)
    ~OfflineMapUpdater();
    void set_global_map(const pcl::PointCloud<pcl::PointXYZI> &map_init);  // load_global_map's copy, OMU.cpp:133
    // one erasor::node message: header.seq, odom, lidar (msg/node.msg:1-4) — OMU.cpp:203-330
    void callback_node(int seq, const geometry_msgs::Pose &odom, const pcl::PointCloud<pcl::PointXYZI> &lidar);
    // Offline look-ahead (no counterpart in the reference, which learns about a node when its message arrives): called BEFORE
    // callback_node(k) with the cloud of node k + 1 -- tell the updater which cloud the callback after the upcoming one brings.  Its
    // voxelisation / binning then overlap the current node's map-side stages; results are unchanged.  Ignored (returns 0) for nodes
    // the removal_interval gate skips.  The cloud is copied at once: it is the caller's again when the call returns.
    // Round 4: the call returns a TICKET.  Hand it to the callback of that node (callback_node(seq, odom, lidar, ticket)) and nothing
    // has to be recognised: the step runs on the announced copy, the cloud is not read again.  Without a ticket the callback's cloud
    // is compared with the announced copy record by record (a hash of every point, ~0.1 ms): a cloud that differs anywhere is
    // processed as the new cloud it is -- rounds 2-3 sampled 257 points and could run on a stale copy.
    uint64_t announce_next(const pcl::PointCloud<pcl::PointXYZI> &lidar);
    // ... and its odometry, when the whole next node is known (erasor_hip_prefetch_node: the VoI pass of the next callback is
    // launched ahead as well)
    uint64_t announce_next(const pcl::PointCloud<pcl::PointXYZI> &lidar, const geometry_msgs::Pose &odom);
    // Round 6 -- DEEP look-ahead for drivers that know their nodes ahead (main_in_your_env.cpp:92-123 reads them from disk): after
    // set_lookahead(n), n = 1 .. 7, announce_next may be called for up to n nodes beyond the upcoming callback, IN NODE ORDER (each call is
    // about the node behind the last one announced); it returns 0 -- and announces nothing -- for a node the removal_interval gate skips
    // and when n nodes are outstanding (call again before a later callback).  With three or more nodes ahead the library lets consecutive
    // steps overlap and the announced nodes' query chains share their launches (erasor_hip_chain_batch): the offline rate of the C ABI
    // through the reference's own class.  Deep announcements are consumed BY TICKET: a callback without one drops them all.
    void set_lookahead(int n);
    // ... and the node the UPCOMING callback itself brings, while nothing is announced yet (the first node of a sequence: a callback
    // without a ticket in front of announced nodes would drop them); 0 if something is announced already or the gate skips the node
    uint64_t announce_upcoming(const pcl::PointCloud<pcl::PointXYZI> &lidar, const geometry_msgs::Pose &odom);
    int lookahead() const { return max_ahead_; }
    int outstanding() const { return n_ahead_; }  // nodes announced and not yet stepped
    // the callback of an announced node, by ticket (0: like the three-argument form)
    void callback_node(int seq, const geometry_msgs::Pose &odom, const pcl::PointCloud<pcl::PointXYZI> &lidar, uint64_t ticket);
    // The same announcement WITHOUT the copy on the caller's time: the cloud is staged by the UPCOMING callback_node while that node's
    // own step runs on the GPU (0.1 ms of host work per 125 k-point cloud that no longer precedes the step).  `lidar` must stay alive
    // and unchanged until that callback has returned; the callback of node `seq` then finds its ticket by itself.
    void announce_next_deferred(int seq, const pcl::PointCloud<pcl::PointXYZI> &lidar, const geometry_msgs::Pose &odom);
    // is node `seq` staged (its callback will not look at the cloud it is handed)?
    bool staged(int seq) const {
        for (const auto &a : auto_tickets_)
            if (a.first == seq) return true;
        return false;
    }
    void save_static_map(float voxel_size);                    // OMU.cpp:174-196
    void get_map(pcl::PointCloud<pcl::PointXYZI> &dst);        // *map_arranged_
    erasor_hip_handle *handle() { return h_; }                 // for adapters that read more of the last step (ros1_adapter.cpp)
    // last step's products (the clouds the reference publishes, OMU.cpp:316-320)
    pcl::PointCloud<pcl::PointXYZI> map_rejected, query_rejected;
    erasor_step_result last;
    size_t num_processed = 0;

private:
    Config cfg_;
    erasor_hip_handle *h_ = nullptr;
    Eigen::Matrix4f tf_lidar2body_, tf_body2origin_;
    int stack_count_ = 0;
    // nodes announced and not yet consumed; round 6: up to max_ahead_ of them (set_lookahead), in node order.  ahead_upto_: callback
    // number (stack_count_ counts callbacks, gated ones too) of the last node an announcement was made -- or skipped by the gate -- for
    int n_ahead_ = 0, max_ahead_ = 1, ahead_upto_ = 0;
    const pcl::PointCloud<pcl::PointXYZI> *def_cloud_ = nullptr;  // announce_next_deferred: staged by the upcoming callback
    geometry_msgs::Pose def_odom_;
    int def_seq_ = 0;
    // tickets of the nodes announced that way, by sequence number (their callbacks take them by themselves); round 6: several, since with
    // set_lookahead(n) the deferred announcement is about the node behind the last announced one, not about the next callback
    std::deque<std::pair<int, uint64_t>> auto_tickets_;
    void stage_deferred();
    uint64_t announce(const pcl::PointCloud<pcl::PointXYZI> &lidar, const geometry_msgs::Pose *odom, bool upcoming = false);
};
// main_in_your_env.cpp:66-70: the driver's own rosparams
struct DriverConfig {
    std::string data_dir = "/";
    double voxel_size = 0.075;
    int init_idx = 0, interval = 2;
};
// Reads a rosparam YAML file in the reference's layout (config/*.yaml).  Keys that are absent keep the values already
// in cfg (fill it with the reference's defaults first: erasor_hip_params_default + OMU.cpp:66-83).
bool load_config_yaml(const std::string &path, OfflineMapUpdater::Config &cfg, DriverConfig *drv = nullptr);
// main_in_your_env.cpp:33-59: poses_lidar2body.csv -> 4x4 float transforms (one per line after the header)
bool load_all_poses(const std::string &txt, std::vector<Eigen::Matrix4f> &poses);
}  // namespace erasor
#endif
