// ros1_adapter.cpp — the thin ROS1 node around the HIP hot path: what makes liberasor_hip.so "drop into the existing
// ROS pipeline" literally.  ZERO arithmetic here: messages in, C-ABI calls, messages out.
//
// Compiled only where ROS1 + PCL exist:  -DERASOR_SHIM_WITH_PCL -DERASOR_SHIM_WITH_ROS (catkin: roscpp pcl_ros
// pcl_conversions jsk_recognition_msgs nav_msgs + the reference's erasor/node message).  In this repository's tests
// it is compiled against the test suite's stand-in ros / pcl / message headers and driven through their in-process topic
// registry: tests/test_gpu_shim.py::test_ros1_adapter_node, tests/test_adapter_compiles.py.
//
// Same node surface as the reference's erasor::OfflineMapUpdater (OfflineMapUpdater.cpp:5-23, 63-105, 169-172, 203-330):
//   subscribes  /node/combined/optimized (erasor::node: header.seq, odom, lidar — msg/node.msg:1-4), /saveflag (std_msgs/Float32)
//   publishes   /MapUpdater/path_corrected, /MapUpdater/map_init, /MapUpdater/map_rejected, /MapUpdater/curr_rejected,
//               /MapUpdater/pc2_curr, /MapUpdater/debug/{map,pc_curr_body,map_body,map_init_arranged}, /MapUpdater/dynamic,
//               /MapUpdater/static, and ERASOR's /SCDR/debug/polygons_marker (SRT status per bin, erasor.cpp:496-570, 630-670)
//   rosparams   /erasor/*, /MapUpdater/*, /large_scale/*, /tf/lidar2body, /verbose — the reference's names and defaults
// Full-map topics (/MapUpdater/static, /dynamic, /map_init) cost a device->host copy of the whole map per node, as they
// cost the reference a serialisation of it: they are filled only while someone subscribes.
#include <ros/ros.h>
#include <pcl_conversions/pcl_conversions.h>
#include <erasor/node.h>
#include <std_msgs/Float32.h>
#include <nav_msgs/Path.h>
#include <sensor_msgs/PointCloud2.h>
#include <geometry_msgs/PolygonStamped.h>
#include <geometry_msgs/Point32.h>
#include <jsk_recognition_msgs/PolygonArray.h>

#include <cmath>
#include <memory>

#include "erasor_shim.h"

namespace erasor {

class OfflineMapUpdaterNode {
public:
    OfflineMapUpdaterNode() {
        // OMU.cpp:6-23
        sub_node_ = nh.subscribe<erasor::node>("/node/combined/optimized", 2000, &OfflineMapUpdaterNode::callback_node, this);
        sub_flag_ = nh.subscribe<std_msgs::Float32>("/saveflag", 10, &OfflineMapUpdaterNode::callback_flag, this);
        pub_path_ = nh.advertise<nav_msgs::Path>("/MapUpdater/path_corrected", 100);
        pub_map_init_ = nh.advertise<sensor_msgs::PointCloud2>("/MapUpdater/map_init", 100);
        pub_map_rejected_ = nh.advertise<sensor_msgs::PointCloud2>("/MapUpdater/map_rejected", 100);
        pub_curr_rejected_ = nh.advertise<sensor_msgs::PointCloud2>("/MapUpdater/curr_rejected", 100);
        pub_debug_pc2_curr_ = nh.advertise<sensor_msgs::PointCloud2>("/MapUpdater/pc2_curr", 100);
        pub_debug_map_ = nh.advertise<sensor_msgs::PointCloud2>("/MapUpdater/debug/map", 100);
        pub_debug_query_egocentric_ = nh.advertise<sensor_msgs::PointCloud2>("/MapUpdater/debug/pc_curr_body", 100);
        pub_debug_map_egocentric_ = nh.advertise<sensor_msgs::PointCloud2>("/MapUpdater/debug/map_body", 100);
        pub_debug_map_arranged_init_ = nh.advertise<sensor_msgs::PointCloud2>("/MapUpdater/debug/map_init_arranged", 100);
        pub_dynamic_arranged_ = nh.advertise<sensor_msgs::PointCloud2>("/MapUpdater/dynamic", 100);
        pub_static_arranged_ = nh.advertise<sensor_msgs::PointCloud2>("/MapUpdater/static", 100);
        pub_viz_bin_marker_ = nh.advertise<jsk_recognition_msgs::PolygonArray>("/SCDR/debug/polygons_marker", 100);
        set_params();
        updater_.reset(new OfflineMapUpdater(cfg_));  // loads /MapUpdater/initial_map_path (OMU.cpp:107-167)
        if (!cfg_.is_large_scale) {
            // the reference keeps map_init_ from load time on and publishes it at load and after every step (OMU.cpp:162-165,
            // 322-324); a subscriber cannot be there yet inside the constructor, so the copy is taken unconditionally
            updater_->get_map(map_init_);
            publish(map_init_, pub_map_init_);
        }
    }
    ~OfflineMapUpdaterNode() { flush_held(); }
    void save_static_map(float voxel_size) { updater_->save_static_map(voxel_size); }
    OfflineMapUpdater &updater() { return *updater_; }

private:
    // OMU.cpp:63-105 and erasor.h:47-61: every parameter under its reference name, with the reference's default
    void set_params() {
        nh = ros::NodeHandle("~");
        erasor_params &p = cfg_.params;
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
        nh.param("/MapUpdater/query_voxel_size", p.query_voxel_size, 0.05);
        nh.param("/MapUpdater/removal_interval", i, 2); p.removal_interval = i;
        nh.param<std::string>("/MapUpdater/data_name", cfg_.data_name, "00");
        nh.param<std::string>("/MapUpdater/env", cfg_.environment, "outdoor");
        nh.param<std::string>("/MapUpdater/initial_map_path", cfg_.initial_map_path, "/");
        nh.param<std::string>("/MapUpdater/save_path", cfg_.save_path, "/");
        nh.param<bool>("/large_scale/is_large_scale", cfg_.is_large_scale, false);
        nh.param("/large_scale/submap_size", cfg_.submap_size, 200.0);
        // fetch_VoI's radius is /erasor/max_range read with its own default of 60 (OMU.cpp:78)
        double voi_range;
        nh.param("/erasor/max_range", voi_range, 60.0);
        p.voi_max_range = voi_range;
        nh.param("/verbose", cfg_.verbose, true);
        nh.param<bool>("/MapUpdater/lookahead_hold", hold_, false);
        std::vector<double> lidar2body;
        if (nh.getParam("/tf/lidar2body", lidar2body) && lidar2body.size() == 7)  // OMU.cpp:89-104
            for (int k = 0; k < 7; ++k) cfg_.lidar2body[k] = lidar2body[k];
        geometry_msgs::Pose l2b;
        l2b.position.x = cfg_.lidar2body[0]; l2b.position.y = cfg_.lidar2body[1]; l2b.position.z = cfg_.lidar2body[2];
        l2b.orientation.x = cfg_.lidar2body[3]; l2b.orientation.y = cfg_.lidar2body[4]; l2b.orientation.z = cfg_.lidar2body[5];
        l2b.orientation.w = cfg_.lidar2body[6];
        tf_lidar2body_ = erasor_utils::geoPose2eigen(l2b);
    }

    void callback_flag(const std_msgs::Float32::ConstPtr &msg) {  // OMU.cpp:169-172
        flush_held();
        save_static_map(msg->data);
    }

    // /MapUpdater/lookahead_hold (default false = the reference's timing: node k is processed when it arrives).  When true the node
    // keeps ONE message back: node k is processed when node k+1 arrives, and k+1 is announced first (OfflineMapUpdater::announce_next
    // -> erasor_hip_prefetch_node), so that its voxelisation / binning and the next VoI pass overlap node k's map-side stages --
    // the look-ahead the offline driver uses, under ROS.  Results are unchanged; every publication is one message late, and the
    // last node is processed when /saveflag arrives (or at shutdown).
    void callback_node(const erasor::node::ConstPtr &msg) {
        if (!hold_) {
            process(*msg);
            return;
        }
        if (held_) {
            pcl::PointCloud<pcl::PointXYZI> next;
            pcl::fromROSMsg(msg->lidar, next);
            // announced WITHOUT a copy on this thread's time: the held node's callback stages `next` (alive until process() returns)
            // while its own step runs on the GPU, and the callback of header.seq finds its ticket by itself
            updater_->announce_next_deferred((int)msg->header.seq, next, msg->odom);
            process(*held_, held_announced_);
            held_announced_ = true;
        }
        held_ = msg;
    }
    void flush_held() {
        if (held_) process(*held_, held_announced_);
        held_.reset();
        held_announced_ = false;
    }

    void process(const erasor::node &node, bool announced = false) {
        const erasor::node *msg = &node;
        pcl::PointCloud<pcl::PointXYZI> query;
        // OMU.cpp:237 -- unless the node was announced and will be processed: its cloud is on the device already (the gate is the
        // updater's: a node the removal interval skips was not staged, and its callback does not look at the cloud either)
        if (!announced || !updater_->staged((int)msg->header.seq)) pcl::fromROSMsg(msg->lidar, query);
        const size_t before = updater_->num_processed;
        updater_->callback_node((int)msg->header.seq, msg->odom, query);  // gate, voxelise, fetch_VoI, ERASOR, write-back on the GPU
        if (updater_->num_processed == before) {
            ROS_INFO_STREAM("\033[1;32m PASS! \033[0m");  // OMU.cpp:327-329
            return;
        }
        // path (set_path, OMU.cpp:467-477): the pose goes through geoPose2eigen -> eigen2geoPose like in the reference
        geometry_msgs::PoseStamped ps;
        ps.header = msg->header;
        ps.header.frame_id = "map";
        ps.pose = erasor_utils::eigen2geoPose(erasor_utils::geoPose2eigen(msg->odom));
        path_.header = ps.header;
        path_.poses.push_back(ps);
        // the clouds the reference publishes after a step (OMU.cpp:259-260, 316-326)
        erasor_hip_handle *h = updater_->handle();
        if (pub_debug_map_egocentric_.getNumSubscribers() > 0) publish_device_cloud(h, ERASOR_CLOUD_MAP_VOI, pub_debug_map_egocentric_);
        if (pub_debug_query_egocentric_.getNumSubscribers() > 0) publish_device_cloud(h, ERASOR_CLOUD_QUERY_VOI, pub_debug_query_egocentric_);
        if (pub_debug_pc2_curr_.getNumSubscribers() > 0) {  // ptr_query_viz = body2origin(query_voi) (OMU.cpp:242)
            pcl::PointCloud<pcl::PointXYZI> q;
            fetch(h, ERASOR_CLOUD_QUERY_VOI, q);
            const Eigen::Matrix4f T = erasor_utils::geoPose2eigen(msg->odom);
            for (auto &p : q.points) {  // pcl::transformPointCloud's float formula; presentation only
                const float x = p.x, y = p.y, z = p.z;
                p.x = ((T(0, 0) * x + T(0, 1) * y) + T(0, 2) * z) + T(0, 3);
                p.y = ((T(1, 0) * x + T(1, 1) * y) + T(1, 2) * z) + T(1, 3);
                p.z = ((T(2, 0) * x + T(2, 1) * y) + T(2, 2) * z) + T(2, 3);
            }
            publish(q, pub_debug_pc2_curr_);
        }
        if (pub_static_arranged_.getNumSubscribers() > 0 || pub_dynamic_arranged_.getNumSubscribers() > 0) {
            pcl::PointCloud<pcl::PointXYZI> m, dyn, stat;
            updater_->get_map(m);
            erasor_utils::parse_dynamic_obj(m, dyn, stat);  // OMU.cpp:294
            publish(stat, pub_static_arranged_);
            publish(dyn, pub_dynamic_arranged_);
        }
        publish(updater_->map_rejected, pub_map_rejected_);
        publish(updater_->query_rejected, pub_curr_rejected_);
        if (!cfg_.is_large_scale && pub_map_init_.getNumSubscribers() > 0) publish(map_init_, pub_map_init_);
        if (pub_viz_bin_marker_.getNumSubscribers() > 0) publish_polygons(h, msg->header);
        pub_path_.publish(path_);
    }

    static void fetch(erasor_hip_handle *h, int which, pcl::PointCloud<pcl::PointXYZI> &dst) {
        size_t n = 0;
        if (erasor_hip_get_cloud(h, which, nullptr, 0, &n) != ERASOR_OK) return;
        std::vector<float> v(n * 4 + 4);
        if (erasor_hip_get_cloud(h, which, v.data(), n, &n) != ERASOR_OK) return;
        dst.points.resize(n);
        for (size_t i = 0; i < n; ++i) {
            dst.points[i].x = v[4 * i]; dst.points[i].y = v[4 * i + 1]; dst.points[i].z = v[4 * i + 2]; dst.points[i].intensity = v[4 * i + 3];
        }
        dst.width = (uint32_t)n;
        dst.height = 1;
    }
    void publish_device_cloud(erasor_hip_handle *h, int which, const ros::Publisher &pub) {
        pcl::PointCloud<pcl::PointXYZI> c;
        fetch(h, which, c);
        publish(c, pub);
    }
    void publish(const pcl::PointCloud<pcl::PointXYZI> &cloud, const ros::Publisher &publisher) {  // OMU.cpp:486-493
        pcl::toROSMsg(cloud, pc2_map_);
        pc2_map_.header.frame_id = "map";
        publisher.publish(pc2_map_);
    }

    // the SRT status of every bin as the reference's polygon array (erasor.cpp:493-498, 566-570; set_polygons :630-670)
    void publish_polygons(erasor_hip_handle *h, const std_msgs::Header &hdr) {
        const erasor_params &p = cfg_.params;
        std::vector<double> status((size_t)p.num_rings * p.num_sectors);
        if (erasor_hip_get_status(h, status.data()) != ERASOR_OK) return;
        const double ring_size = p.max_range / p.num_rings, sector_size = 2 * 3.1415926535 / p.num_sectors;  // erasor.h:4,63-64
        jsk_recognition_msgs::PolygonArray poly_list;
        poly_list.header.frame_id = "map";
        poly_list.header.stamp = hdr.stamp;
        for (int theta = 0; theta < p.num_sectors; theta++)
            for (int r = 0; r < p.num_rings; r++) {
                geometry_msgs::PolygonStamped polygons;
                polygons.header = poly_list.header;
                const int num_split = 3;
                geometry_msgs::Point32 point;
                point.z = (float)(p.max_h + 0.5);
                double r_len = r * ring_size, angle = theta * sector_size;
                auto push = [&]() {
                    point.x = (float)(r_len * cos(angle));
                    point.y = (float)(r_len * sin(angle));
                    polygons.polygon.points.push_back(point);
                };
                push();                 // RL
                r_len += ring_size;
                push();                 // RU
                for (int idx = 1; idx <= num_split; ++idx) { angle += sector_size / num_split; push(); }  // RU -> LU
                r_len -= ring_size;
                push();                 // LL
                for (int idx = 1; idx < num_split; ++idx) { angle -= sector_size / num_split; push(); }   // back along the inner arc
                poly_list.polygons.push_back(polygons);
                poly_list.likelihood.push_back((float)status[(size_t)r * p.num_sectors + theta]);
            }
        pub_viz_bin_marker_.publish(poly_list);
    }

    ros::NodeHandle nh;
    ros::Subscriber sub_node_, sub_flag_;
    ros::Publisher pub_path_, pub_map_init_, pub_static_arranged_, pub_dynamic_arranged_, pub_map_rejected_, pub_curr_rejected_;
    ros::Publisher pub_debug_map_arranged_init_, pub_debug_query_egocentric_, pub_debug_map_egocentric_, pub_debug_pc2_curr_, pub_debug_map_;
    ros::Publisher pub_viz_bin_marker_;
    OfflineMapUpdater::Config cfg_;
    std::unique_ptr<OfflineMapUpdater> updater_;
    Eigen::Matrix4f tf_lidar2body_;
    pcl::PointCloud<pcl::PointXYZI> map_init_;
    sensor_msgs::PointCloud2 pc2_map_;
    nav_msgs::Path path_;
    bool hold_ = false;
    erasor::node::ConstPtr held_;
    bool held_announced_ = false;  // the held message was announced (announce_next_deferred) when it arrived
};

}  // namespace erasor

#ifndef ERASOR_ROS1_ADAPTER_NO_MAIN
// main_kitti.cpp:4-11
int main(int argc, char **argv) {
    ros::init(argc, argv, "ERASOR_STATIC_MAP_BUILDING");
    ros::NodeHandle nh;
    erasor::OfflineMapUpdaterNode updater;
    ros::spin();
    return 0;
}
#endif
