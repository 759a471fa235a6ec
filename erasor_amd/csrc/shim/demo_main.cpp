// erasor_offline_demo — ROS-free driver shaped like src/offline_map_updater/main_in_your_env.cpp:61-127:
//   <dir>/map.pcd, <dir>/pcds/%06d.pcd, <dir>/poses.csv (header line; idx,?,x,y,z,qx,qy,qz,qw per line, cols 2..8)
// processes every node through erasor::OfflineMapUpdater and writes <dir>/<data_name>_result.pcd and map_final.pcd.
#include <cstdio>
#include <fstream>
#include <sstream>

#include "erasor_shim.h"

static bool read_bin(const std::string &path, pcl::PointCloud<pcl::PointXYZI> &c) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const size_t n = (size_t)f.tellg() / 16;
    f.seekg(0);
    std::vector<float> v(n * 4);
    f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(n * 16));
    c.points.resize(n);
    for (size_t i = 0; i < n; ++i) { c.points[i].x = v[4 * i]; c.points[i].y = v[4 * i + 1]; c.points[i].z = v[4 * i + 2]; c.points[i].intensity = v[4 * i + 3]; }
    return true;
}
static void write_bin(const std::string &path, const pcl::PointCloud<pcl::PointXYZI> &c) {
    std::ofstream f(path, std::ios::binary);
    for (const auto &p : c.points) {
        const float v[4] = {p.x, p.y, p.z, p.intensity};
        f.write(reinterpret_cast<const char *>(v), 16);
    }
}

// --erasor-class <map_voi.bin> <query_voi.bin> <out_prefix> <version>: the ERASOR class used the way
// OfflineMapUpdater.cpp:266-284 uses it (set_inputs -> compare_* -> get_static_estimate -> get_outliers)
static int erasor_class_mode(int argc, char **argv) {
    if (argc < 6) return 2;
    erasor_params p;
    erasor_hip_params_default(&p);
    p.max_range = 60.0; p.num_rings = 15; p.num_sectors = 60; p.min_h = -1.3; p.max_h = 3.2; p.th_bin_max_h = 0.05;
    p.scan_ratio_threshold = 0.3; p.minimum_num_pts = 10; p.gf_dist_thr = 0.15; p.gf_iter = 3; p.gf_num_lpr = 10; p.gf_th_seeds_height = 0.5;
    pcl::PointCloud<pcl::PointXYZI> map_voi, query_voi, arranged, complement, map_rejected, curr_rejected;
    if (!read_bin(argv[2], map_voi) || !read_bin(argv[3], query_voi)) return 3;
    const std::string out = argv[4];
    const int version = atoi(argv[5]);
    ERASOR erasor(p);
    erasor.set_inputs(map_voi, query_voi);
    if (version == 2) erasor.compare_vois_and_revert_ground(0);
    else if (version == 3) erasor.compare_vois_and_revert_ground_w_block(0);
    else throw std::invalid_argument("Other version is not implemented!");
    erasor.get_static_estimate(arranged, complement);
    erasor.get_outliers(map_rejected, curr_rejected);
    write_bin(out + "_arranged.bin", arranged);
    write_bin(out + "_complement.bin", complement);
    write_bin(out + "_map_rejected.bin", map_rejected);
    write_bin(out + "_ground_viz.bin", erasor.ground_viz);
    {   // the public R-PODs (erasor.h:143-145): per bin count / min_h / max_h / status, theta-major point dumps, and
        // is_dynamic_obj_close for every bin (erasor.cpp:573-595)
        std::ofstream f(out + "_rpod.txt");
        pcl::PointCloud<pcl::PointXYZI> flat[3];
        R_POD *pods[3] = {&erasor.r_pod_map, &erasor.r_pod_curr, &erasor.r_pod_selected};
        for (int t = 0; t < p.num_sectors; ++t)
            for (int r = 0; r < p.num_rings; ++r) {
                for (int w = 0; w < 3; ++w)
                    if ((*pods[w])[r][t].is_occupied) flat[w] += (*pods[w])[r][t].points;  // r_pod2pc, erasor.cpp:309-320
                const Bin &m = erasor.r_pod_map[r][t], &c = erasor.r_pod_curr[r][t], &s = erasor.r_pod_selected[r][t];
                char line[256];
                snprintf(line, sizeof(line), "%d %d %zu %.17g %.17g %zu %.17g %.17g %zu %.17g %d\n", r, t, m.points.size(), m.min_h, m.max_h,
                         c.points.size(), c.min_h, c.max_h, s.points.size(), s.status,
                         erasor.is_dynamic_obj_close(erasor.r_pod_selected, r, t, 1, 1) ? 1 : 0);
                f << line;
            }
        write_bin(out + "_rpod_map.bin", flat[0]);
        write_bin(out + "_rpod_curr.bin", flat[1]);
        write_bin(out + "_rpod_selected.bin", flat[2]);
    }
    printf("ERASOR class: arranged %zu complement %zu rejected %zu max_range %.1f\n", arranged.size(), complement.size(), map_rejected.size(),
           erasor.get_max_range());
    return 0;
}

// --config <rosparam.yaml> [n_frames]: the reference's own driver, main_in_your_env.cpp:61-127, without ROS:
//   <data_dir>/poses_lidar2body.csv, <data_dir>/pcds/%06d.pcd from init_idx on, every node through
//   OfflineMapUpdater::callback_node (pose -> eigen2geoPose -> node.odom), then save_static_map(0.2).
// The initial map is /MapUpdater/initial_map_path, or <data_dir>/dense_global_map.pcd when that key is absent.
static int config_mode(int argc, char **argv) {
    if (argc < 3) return 2;
    erasor::OfflineMapUpdater::Config cfg;
    erasor_hip_params_default(&cfg.params);
    cfg.params.query_voxel_size = 0.05;  // OMU.cpp:66
    cfg.params.removal_interval = 2;     // OMU.cpp:69
    cfg.verbose = true;                  // OMU.cpp:83
    erasor::DriverConfig drv;
    if (!erasor::load_config_yaml(argv[2], cfg, &drv)) {
        fprintf(stderr, "cannot read %s\n", argv[2]);
        return 3;
    }
    const int max_frames = argc > 3 ? atoi(argv[3]) : 1 << 30;
    if (cfg.initial_map_path.empty() || cfg.initial_map_path == "/") cfg.initial_map_path = drv.data_dir + "/dense_global_map.pcd";
    if (cfg.save_path == "/" || cfg.save_path == ".") cfg.save_path = drv.data_dir;
    erasor::OfflineMapUpdater updater(cfg);
    std::vector<Eigen::Matrix4f> poses;
    if (!erasor::load_all_poses(drv.data_dir + "/poses_lidar2body.csv", poses)) {
        fprintf(stderr, "cannot read %s/poses_lidar2body.csv\n", drv.data_dir.c_str());
        return 3;
    }
    printf("Total %zu poses are loaded\n", poses.size() + 1);  // main_in_your_env.cpp:58 counts the header line too
    int done = 0;
    auto load = [&](int i, pcl::PointCloud<pcl::PointXYZI> &c) {
        char name[64];
        snprintf(name, sizeof(name), "/pcds/%06d.pcd", i);
        return erasor_utils::load_pcd(drv.data_dir + name, c) != -1;
    };
    pcl::PointCloud<pcl::PointXYZI> scan, next;
    bool have = drv.init_idx < (int)poses.size() && load(drv.init_idx, scan);
    for (int i = drv.init_idx; have && i < (int)poses.size() && done < max_frames; ++i, ++done) {
        // offline: the next node's cloud is read (and announced) before this node is processed
        const bool have_next = i + 1 < (int)poses.size() && done + 1 < max_frames && load(i + 1, next);
        if (have_next) updater.announce_next(next, erasor_utils::eigen2geoPose(poses[i + 1]));
        updater.callback_node(i, erasor_utils::eigen2geoPose(poses[i]), scan);
        scan.points.swap(next.points);
        have = have_next;
    }
    if (done == 0 && !have) return 3;
    updater.save_static_map(0.2f);  // main_in_your_env.cpp:123
    printf("Static map building complete!\n");
    return 0;
}

// --mapgen <data_dir> <n_frames> <voxelsize> <is_large_scale>: src/mapgen/main.cpp:41-64 without ROS — every node of
// <data_dir>/poses_lidar2body.csv + pcds/%06d.pcd through mapgen::accumPointCloud, then saveNaiveMap
static int mapgen_mode(int argc, char **argv) {
    if (argc < 6) return 2;
    const std::string dir = argv[2];
    const int n = atoi(argv[3]);
    mapgen gen;
    gen.setValue(dir, (float)atof(argv[4]), "05", "0", std::to_string(n - 1), 1, atoi(argv[5]) != 0);
    std::vector<Eigen::Matrix4f> poses;
    if (!erasor::load_all_poses(dir + "/poses_lidar2body.csv", poses)) return 3;
    for (int i = 0; i < n && i < (int)poses.size(); ++i) {
        char name[64];
        snprintf(name, sizeof(name), "/pcds/%06d.pcd", i);
        pcl::PointCloud<pcl::PointXYZI> scan;
        if (erasor_utils::load_pcd(dir + name, scan) == -1) return 3;
        gen.accumPointCloud(erasor_utils::eigen2geoPose(poses[i]), scan);
    }
    pcl::PointCloud<pcl::PointXYZI> m, c;
    gen.getPointClouds(m, c);
    write_bin(dir + "/mapgen_cloud_map.bin", m);
    write_bin(dir + "/mapgen_cloud_curr.bin", c);
    gen.saveNaiveMap(dir + "/05_original.pcd", gen.map_file_name());
    printf("[MAPGEN] map saved to %s\n", gen.map_file_name().c_str());
    return 0;
}

// --voxelize <in.bin> <leaf> <out.bin>: the free function erasor_utils::voxelize_preserving_labels(Ptr, Cloud&, double)
// (utils.hpp:103), as OMU.cpp:186,238 / erasor.cpp:528 / mapgen.hpp:239 call it
static int voxelize_mode(int argc, char **argv) {
    if (argc < 5) return 2;
    pcl::PointCloud<pcl::PointXYZI>::Ptr src(new pcl::PointCloud<pcl::PointXYZI>());
    pcl::PointCloud<pcl::PointXYZI> dst;
    if (!read_bin(argv[2], *src)) return 3;
    erasor_utils::voxelize_preserving_labels(src, dst, atof(argv[3]));
    write_bin(argv[4], dst);
    printf("voxelize_preserving_labels: %zu -> %zu\n", src->size(), dst.size());
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && std::string(argv[1]) == "--voxelize") {
        try {
            return voxelize_mode(argc, argv);
        } catch (const std::exception &e) {
            fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
    }
    if (argc >= 2 && std::string(argv[1]) == "--mapgen") {
        try {
            return mapgen_mode(argc, argv);
        } catch (const std::exception &e) {
            fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
    }
    if (argc >= 2 && (std::string(argv[1]) == "--erasor-class" || std::string(argv[1]) == "--config")) {
        try {
            return std::string(argv[1]) == "--config" ? config_mode(argc, argv) : erasor_class_mode(argc, argv);
        } catch (const std::exception &e) {
            fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
    }
    if (argc < 3) {
        fprintf(stderr, "usage: %s <data_dir> <n_frames> [version] [removal_interval]\n       %s --config <rosparam.yaml> [n_frames]\n", argv[0], argv[0]);
        return 2;
    }
    const std::string dir = argv[1];
    const int n = atoi(argv[2]);
    erasor::OfflineMapUpdater::Config cfg;
    erasor_hip_params_default(&cfg.params);
    // config/seq_05.yaml
    cfg.params.max_range = 60.0; cfg.params.num_rings = 15; cfg.params.num_sectors = 60; cfg.params.min_h = -1.3; cfg.params.max_h = 3.2;
    cfg.params.th_bin_max_h = 0.05; cfg.params.scan_ratio_threshold = 0.3; cfg.params.minimum_num_pts = 10; cfg.params.gf_dist_thr = 0.15;
    cfg.params.gf_iter = 3; cfg.params.gf_num_lpr = 10; cfg.params.gf_th_seeds_height = 0.5; cfg.params.query_voxel_size = 0.2;
    cfg.params.version = argc > 3 ? atoi(argv[3]) : 3;
    cfg.params.removal_interval = argc > 4 ? atoi(argv[4]) : 1;
    cfg.lidar2body[2] = 1.73;
    cfg.initial_map_path = dir + "/map.pcd";
    cfg.save_path = dir;
    cfg.data_name = "05";
    try {
        erasor::OfflineMapUpdater updater(cfg);
        std::ifstream in(dir + "/poses.csv");
        std::string line;
        std::getline(in, line);  // header
        for (int i = 0; i < n; ++i) {
            if (!std::getline(in, line)) break;
            std::vector<double> v;
            std::stringstream ss(line);
            std::string t;
            while (std::getline(ss, t, ',')) v.push_back(atof(t.c_str()));
            geometry_msgs::Pose odom;
            odom.position.x = v[2]; odom.position.y = v[3]; odom.position.z = v[4];
            odom.orientation.x = v[5]; odom.orientation.y = v[6]; odom.orientation.z = v[7]; odom.orientation.w = v[8];
            char name[64];
            snprintf(name, sizeof(name), "/pcds/%06d.pcd", i);
            pcl::PointCloud<pcl::PointXYZI> scan;
            if (erasor_utils::load_pcd(dir + name, scan) == -1) return 3;
            updater.callback_node(i, odom, scan);
            printf("frame %d: voi %llu rejected %llu reverted bins %u map %llu\n", i, (unsigned long long)updater.last.n_voi,
                   (unsigned long long)updater.last.n_map_rejected, updater.last.n_reverted_bins, (unsigned long long)updater.last.n_map_out);
        }
        pcl::PointCloud<pcl::PointXYZI> m;
        updater.get_map(m);
        erasor_utils::save_pcd_ascii(dir + "/map_final.pcd", m);
        write_bin(dir + "/map_final.bin", m);
        updater.save_static_map(0.2f);
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
