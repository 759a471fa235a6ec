// ros1_adapter_check.cpp — drives erasor_amd/csrc/shim/ros1_adapter.cpp the way roscore would, through the in-process
// topic / parameter registries of the stand-in ROS headers (oracle/stubs).  TEST INFRASTRUCTURE.
//   ros1_adapter_check <dir> <n_nodes> <version> <removal_interval>
// <dir>/map.pcd, <dir>/scan%d.bin, <dir>/poses.txt (7 doubles per line) -> <dir>/out_<topic>_%d.bin per processed node
#define ERASOR_ROS1_ADAPTER_NO_MAIN
#include "../../erasor_amd/csrc/shim/ros1_adapter.cpp"

#include <cstdio>
#include <fstream>

static void dump(const std::string &path, const sensor_msgs::PointCloud2 &m) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(m.xyzi.data()), (std::streamsize)(m.xyzi.size() * 4));
}

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const std::string d = argv[1];
    const int n = atoi(argv[2]), version = atoi(argv[3]), interval = atoi(argv[4]);
    auto &P = ros::stub::params();
    // config/seq_05.yaml
    P["/erasor/max_range"] = 60.0; P["/erasor/num_rings"] = 15; P["/erasor/num_sectors"] = 60;
    P["/erasor/min_h"] = -1.3; P["/erasor/max_h"] = 3.2; P["/erasor/th_bin_max_h"] = 0.05;
    P["/erasor/scan_ratio_threshold"] = 0.3; P["/erasor/minimum_num_pts"] = 10;
    P["/erasor/gf_dist_thr"] = 0.15; P["/erasor/gf_iter"] = 3; P["/erasor/gf_num_lpr"] = 10; P["/erasor/gf_th_seeds_height"] = 0.5;
    P["/erasor/version"] = version;
    P["/MapUpdater/query_voxel_size"] = 0.2; P["/MapUpdater/removal_interval"] = interval;
    P["/MapUpdater/data_name"] = std::string("05"); P["/MapUpdater/initial_map_path"] = d + "/map.pcd"; P["/MapUpdater/save_path"] = d;
    P["/tf/lidar2body"] = std::vector<double>{0.0, 0.0, 1.73, 0.0, 0.0, 0.0, 1.0};
    P["/verbose"] = false;
    if (argc > 5 && atoi(argv[5])) P["/MapUpdater/lookahead_hold"] = true;  // node k processed when node k+1 arrives (announced first)
    for (const char *t : {"/MapUpdater/map_rejected", "/MapUpdater/curr_rejected", "/MapUpdater/static", "/MapUpdater/dynamic", "/MapUpdater/debug/map_body",
                          "/MapUpdater/debug/pc_curr_body", "/MapUpdater/pc2_curr", "/MapUpdater/path_corrected", "/SCDR/debug/polygons_marker"})
        ros::stub::capture_topics()[t] = true;
    int ac = 1;
    ros::init(ac, argv, "ERASOR_STATIC_MAP_BUILDING");
    erasor::OfflineMapUpdaterNode node;
    std::ifstream poses(d + "/poses.txt");
    int processed = 0;
    auto dump_published = [&](int k) {
        auto &pub = ros::stub::published();
        if (!pub.count("/MapUpdater/map_rejected")) return;  // gated out (PASS!) -- or held back (lookahead_hold)
        const std::string tag = "_" + std::to_string(processed++) + ".bin";
        dump(d + "/out_map_rejected" + tag, std::any_cast<const sensor_msgs::PointCloud2 &>(pub["/MapUpdater/map_rejected"]));
        dump(d + "/out_curr_rejected" + tag, std::any_cast<const sensor_msgs::PointCloud2 &>(pub["/MapUpdater/curr_rejected"]));
        dump(d + "/out_static" + tag, std::any_cast<const sensor_msgs::PointCloud2 &>(pub["/MapUpdater/static"]));
        dump(d + "/out_dynamic" + tag, std::any_cast<const sensor_msgs::PointCloud2 &>(pub["/MapUpdater/dynamic"]));
        dump(d + "/out_map_body" + tag, std::any_cast<const sensor_msgs::PointCloud2 &>(pub["/MapUpdater/debug/map_body"]));
        dump(d + "/out_pc_curr_body" + tag, std::any_cast<const sensor_msgs::PointCloud2 &>(pub["/MapUpdater/debug/pc_curr_body"]));
        dump(d + "/out_pc2_curr" + tag, std::any_cast<const sensor_msgs::PointCloud2 &>(pub["/MapUpdater/pc2_curr"]));
        const auto &pa = std::any_cast<const jsk_recognition_msgs::PolygonArray &>(pub["/SCDR/debug/polygons_marker"]);
        std::ofstream lf(d + "/out_likelihood" + tag, std::ios::binary);
        lf.write(reinterpret_cast<const char *>(pa.likelihood.data()), (std::streamsize)(pa.likelihood.size() * 4));
        std::ofstream pf(d + "/out_polygon0" + tag, std::ios::binary);  // vertices of the polygon of bin (ring 1, sector 2)
        const auto &poly = pa.polygons[(size_t)2 * 15 + 1].polygon.points;
        for (const auto &q : poly) {
            const float v[3] = {q.x, q.y, q.z};
            pf.write(reinterpret_cast<const char *>(v), 12);
        }
        const auto &path = std::any_cast<const nav_msgs::Path &>(pub["/MapUpdater/path_corrected"]);
        printf("node %d: path poses %zu, polygons %zu\n", k, path.poses.size(), pa.polygons.size());
    };
    for (int k = 0; k < n; ++k) {
        boost::shared_ptr<erasor::node> msg(new erasor::node());
        msg->header.seq = (uint32_t)k;
        double p[7];
        for (double &v : p) poses >> v;
        msg->odom.position.x = p[0]; msg->odom.position.y = p[1]; msg->odom.position.z = p[2];
        msg->odom.orientation.x = p[3]; msg->odom.orientation.y = p[4]; msg->odom.orientation.z = p[5]; msg->odom.orientation.w = p[6];
        std::ifstream f(d + "/scan" + std::to_string(k) + ".bin", std::ios::binary | std::ios::ate);
        const size_t nf = (size_t)f.tellg() / 4;
        f.seekg(0);
        msg->lidar.xyzi.resize(nf);
        f.read(reinterpret_cast<char *>(msg->lidar.xyzi.data()), (std::streamsize)(nf * 4));
        msg->lidar.width = (uint32_t)(nf / 4);
        ros::stub::published().clear();
        if (!ros::stub::dispatch<erasor::node>("/node/combined/optimized", msg)) return 3;
        dump_published(k);
    }
    // /saveflag -> save_static_map (OMU.cpp:169-196)
    boost::shared_ptr<std_msgs::Float32> flag(new std_msgs::Float32());
    flag->data = 0.2f;
    ros::stub::published().clear();
    if (!ros::stub::dispatch<std_msgs::Float32>("/saveflag", flag)) return 4;
    dump_published(n);  // (lookahead_hold: the node held back is processed when the flag arrives)
    printf("processed %d\n", processed);
    return 0;
}
