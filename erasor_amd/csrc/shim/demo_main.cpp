// erasor_offline_demo — ROS-free driver shaped like src/offline_map_updater/main_in_your_env.cpp:61-127:
//   <dir>/map.pcd, <dir>/pcds/%06d.pcd, <dir>/poses.csv (header line; idx,?,x,y,z,qx,qy,qz,qw per line, cols 2..8)
// processes every node through erasor::OfflineMapUpdater and writes <dir>/<data_name>_result.pcd and map_final.pcd.
#include <cstdlib>
#include <array>
#include <chrono>
#include <cstdio>
#include <deque>
#include <fstream>
#include <memory>
#include <sstream>
#include <thread>

#include "erasor_shim.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
static int run_config(const std::string &yaml, int max_frames, int device, bool verbose, int *nodes_done = nullptr) {
    erasor::OfflineMapUpdater::Config cfg;
    erasor_hip_params_default(&cfg.params);
    cfg.params.query_voxel_size = 0.05;  // OMU.cpp:66
    cfg.params.removal_interval = 2;     // OMU.cpp:69
    cfg.verbose = verbose;               // OMU.cpp:83
    cfg.device = device;
    erasor::DriverConfig drv;
    if (!erasor::load_config_yaml(yaml, cfg, &drv)) {
        fprintf(stderr, "cannot read %s\n", yaml.c_str());
        return 3;
    }
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
    // offline: the driver knows its nodes ahead (main_in_your_env.cpp:92-123 reads them from disk one by one).  Round 6: up to `lookahead`
    // nodes (ERASOR_DEMO_LOOKAHEAD, default 6) are read and ANNOUNCED before the node in front of them is processed -- their query chains
    // then share their launches and consecutive steps overlap (erasor_shim.h: set_lookahead); every node is stepped by the ticket of its
    // announcement.  Results do not depend on it.
    const int lookahead = std::max(1, std::min(getenv("ERASOR_DEMO_LOOKAHEAD") ? atoi(getenv("ERASOR_DEMO_LOOKAHEAD")) : 6, 7));
    updater.set_lookahead(lookahead);
    struct Ahead {
        pcl::PointCloud<pcl::PointXYZI> cloud;
        uint64_t ticket = 0;
        bool announced = false;  // (false: refused for now -- gate or capacity --, tried again before the next callback unless it is the upcoming node)
    };
    std::deque<Ahead> q;  // q[0] = the upcoming node i, q[j] = node i + j
    const int last = std::min((int)poses.size(), drv.init_idx + max_frames);  // one past the last node to process
    auto fill = [&](int i) {
        while ((int)q.size() < lookahead + 1 && i + (int)q.size() < last) {
            q.emplace_back();
            if (!load(i + (int)q.size() - 1, q.back().cloud)) {
                q.pop_back();
                return false;
            }
        }
        return true;
    };
    bool have = drv.init_idx < last && fill(drv.init_idx) && !q.empty();
    for (int i = drv.init_idx; have && i < last; ++i, ++done) {
        (void)fill(i);
        if (i == drv.init_idx && lookahead > 1) q[0].ticket = updater.announce_upcoming(q[0].cloud, erasor_utils::eigen2geoPose(poses[i]));
        // announce, in node order, every node behind the upcoming one that is not announced yet (a node the updater refuses -- gated out, or
        // as many outstanding as it takes -- stops the round: announcements are made in order)
        for (size_t j = 1; j < q.size(); ++j) {
            if (q[j].announced) continue;
            if (i > drv.init_idx && j + 1 == q.size() && updater.outstanding() < updater.lookahead()) {
                // the pipeline is full: the one new node of this round is staged INSIDE the upcoming callback, while its step runs on the
                // GPU (announce_next_deferred; its own callback finds the ticket by the sequence number)
                updater.announce_next_deferred(i + (int)j, q[j].cloud, erasor_utils::eigen2geoPose(poses[i + (int)j]));
                q[j].announced = true;
                break;
            }
            q[j].ticket = updater.announce_next(q[j].cloud, erasor_utils::eigen2geoPose(poses[i + (int)j]));
            q[j].announced = true;  // (ticket 0: gated out by removal_interval, or full: either way this node is stepped without a ticket)
            if (!q[j].ticket && updater.outstanding() >= updater.lookahead()) {
                q[j].announced = false;  // full: again before the next callback
                break;
            }
        }
        updater.callback_node(i, erasor_utils::eigen2geoPose(poses[i]), q[0].cloud, q[0].ticket);
        q.pop_front();
        have = !q.empty() || (i + 1 < last && fill(i + 1) && !q.empty());
    }
    if (nodes_done) *nodes_done = done;
    if (done == 0 && !have) return 3;
    updater.save_static_map(0.2f);  // main_in_your_env.cpp:123
    if (verbose) printf("Static map building complete!\n");
    return 0;
}
static int config_mode(int argc, char **argv) {
    if (argc < 3) return 2;
    return run_config(argv[2], argc > 3 ? atoi(argv[3]) : 1 << 30, 0, true);
}

// --queue <n_workers> <max_frames> <a.yaml> <b.yaml> ...: independent sequences (one rosparam file each) over n_workers devices
// (worker w drives device w mod the visible devices): every worker is a thread with its own updater per job, and takes the NEXT sequence
// of the list whenever it is idle (erasor::WorkQueue) -- BASELINE config 3 on a node with fewer GPUs than sequences.
static int queue_mode(int argc, char **argv) {
    if (argc < 5) return 2;
    const int n_workers = std::max(1, atoi(argv[2])), max_frames = atoi(argv[3]);
    std::vector<std::string> jobs(argv + 4, argv + argc);
    int ndev = 1;
    {   // (the C ABI answers with ERASOR_E_NO_DEVICE for a device that does not exist: probe upwards)
        erasor_params p;
        erasor_hip_params_default(&p);
        for (ndev = 0; ndev < 64; ++ndev) {
            erasor_hip_handle *h = nullptr;
            if (erasor_hip_create(&p, ndev, &h) != ERASOR_OK) break;
            erasor_hip_destroy(h);
        }
        if (ndev == 0) return 4;
    }
    erasor::WorkQueue q(jobs.size());
    std::vector<int> rc(jobs.size(), -1), nodes(jobs.size(), 0);
    std::vector<double> t_begin(jobs.size(), 0), t_end(jobs.size(), 0);
    const double t0 = now_ms();
    std::vector<std::thread> th;
    for (int w = 0; w < n_workers; ++w)
        th.emplace_back([&, w] {
            for (long j; (j = q.next(w)) >= 0;) {
                t_begin[j] = now_ms() - t0;
                try {
                    rc[j] = run_config(jobs[j], max_frames, w % ndev, false, &nodes[j]);
                } catch (const std::exception &e) {
                    fprintf(stderr, "job %ld (%s): %s\n", j, jobs[j].c_str(), e.what());
                    rc[j] = 1;
                }
                t_end[j] = now_ms() - t0;
            }
        });
    for (auto &t : th) t.join();
    int bad = 0;
    for (size_t j = 0; j < jobs.size(); ++j) {
        printf("job %zu %s: worker %d (device %d), %d nodes, %.1f .. %.1f ms, rc %d\n", j, jobs[j].c_str(), q.taken_by(j), q.taken_by(j) % ndev, nodes[j],
               t_begin[j], t_end[j], rc[j]);
        bad += rc[j] != 0;
    }
    printf("{\"mode\": \"queue\", \"workers\": %d, \"devices\": %d, \"jobs\": %zu, \"failed\": %d, \"wall_ms\": %.1f}\n", n_workers, ndev, jobs.size(), bad,
           now_ms() - t0);
    return bad ? 5 : 0;
}

// --replicas <n> <rosparam.yaml> [n_frames]: ONE process, n updaters (device r mod the visible devices), the global map loaded ONCE and
// replicated with erasor_hip_replicate_map (RCCL broadcast over xGMI / peer copies), then replica r works through nodes r, r + n, ... of the
// sequence on its own thread -- scan-parallel replicas (every replica folds ITS scans into ITS copy; a deviation from
// the reference's single sequential fold, which is why nothing is saved here: it reports what each replica removed).
static int replicas_mode(int argc, char **argv) {
    if (argc < 4) return 2;
    const int n = std::max(1, atoi(argv[2]));
    const int max_frames = argc > 4 ? atoi(argv[4]) : 1 << 30;
    erasor::OfflineMapUpdater::Config cfg;
    erasor_hip_params_default(&cfg.params);
    cfg.params.query_voxel_size = 0.05;
    cfg.params.removal_interval = 2;
    erasor::DriverConfig drv;
    if (!erasor::load_config_yaml(argv[3], cfg, &drv)) return 3;
    if (cfg.initial_map_path.empty() || cfg.initial_map_path == "/") cfg.initial_map_path = drv.data_dir + "/dense_global_map.pcd";
    const std::string map_path = cfg.initial_map_path;
    cfg.params.removal_interval = 1;  // (the interval is what spreads the nodes over the replicas here)
    int ndev = 0;
    for (; ndev < 64; ++ndev) {
        erasor_hip_handle *h = nullptr;
        if (erasor_hip_create(&cfg.params, ndev, &h) != ERASOR_OK) break;
        erasor_hip_destroy(h);
    }
    if (ndev == 0) return 4;
    std::vector<std::unique_ptr<erasor::OfflineMapUpdater>> up;
    for (int r = 0; r < n; ++r) {
        erasor::OfflineMapUpdater::Config c = cfg;
        c.device = r % ndev;
        c.initial_map_path = r == 0 ? map_path : std::string();  // only the root reads the file
        up.emplace_back(new erasor::OfflineMapUpdater(c));
    }
    std::vector<erasor_hip_handle *> hs;
    for (auto &u : up) hs.push_back(u->handle());
    int transport = 0;
    const double tb = now_ms();
    if (erasor_hip_replicate_map(hs.data(), n, 0, &transport) != ERASOR_OK) {
        fprintf(stderr, "erasor_hip_replicate_map: %s\n", erasor_hip_last_error(hs[0]));
        return 4;
    }
    const double ms_bcast = now_ms() - tb;
    std::vector<Eigen::Matrix4f> poses;
    if (!erasor::load_all_poses(drv.data_dir + "/poses_lidar2body.csv", poses)) return 3;
    const int last = std::min<int>((int)poses.size(), drv.init_idx + max_frames);
    std::vector<unsigned long long> rejected(n, 0), map_out(n, 0);
    std::vector<int> done(n, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r)
        th.emplace_back([&, r] {
            for (int i = drv.init_idx + r; i < last; i += n) {
                char name[64];
                snprintf(name, sizeof(name), "/pcds/%06d.pcd", i);
                pcl::PointCloud<pcl::PointXYZI> scan;
                if (erasor_utils::load_pcd(drv.data_dir + name, scan) == -1) break;
                up[r]->callback_node(i, erasor_utils::eigen2geoPose(poses[i]), scan);
                rejected[r] += up[r]->last.n_map_rejected;
                map_out[r] = up[r]->last.n_map_out;
                ++done[r];
            }
        });
    for (auto &t : th) t.join();
    for (int r = 0; r < n; ++r)
        printf("replica %d (device %d): %d nodes, %llu map points rejected, map %llu\n", r, r % ndev, done[r], rejected[r], map_out[r]);
    printf("{\"mode\": \"replicas\", \"replicas\": %d, \"devices\": %d, \"transport\": \"%s\", \"replicate_ms\": %.2f}\n", n, ndev,
           transport == 1 ? "rccl" : (transport == 2 ? "peer copies" : "none"), ms_bcast);
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

// --bench <dir> <n_timed> <n_warmup>: what a caller of the drop-in surface gets, timed in C++ (no Python in the loop) on the workload
// tools/export_cpp_bench.py wrote (bench.py's config 2).  Three passes over the same nodes, each on a fresh map:
//   1. erasor::OfflineMapUpdater::callback_node, host cloud in (OMU.cpp:237 fromROSMsg has just produced it), map_rejected /
//      query_rejected clouds out on the host (what OMU.cpp:316-320 publishes) -- nothing announced;
//   2. the same with the next node announced (announce_next: what an offline driver or a node holding one message back can do);
//   3. the C ABI itself with the scans resident in HBM and nodes announced two ahead (bench.py's loop, minus Python).
// Prints ONE JSON line.
static int bench_mode(int argc, char **argv) {
    if (argc < 5) return 2;
    const std::string dir = argv[2];
    const int K = atoi(argv[3]), W = atoi(argv[4]);
    erasor::OfflineMapUpdater::Config cfg;
    {
        std::ifstream f(dir + "/params.bin", std::ios::binary);
        if (!f.read(reinterpret_cast<char *>(&cfg.params), sizeof(cfg.params))) return 3;
    }
    std::vector<double> poses, l2b(7);
    {
        std::ifstream f(dir + "/poses.bin", std::ios::binary | std::ios::ate);
        if (!f) return 3;
        poses.resize((size_t)f.tellg() / 8);
        f.seekg(0);
        f.read(reinterpret_cast<char *>(poses.data()), (std::streamsize)(poses.size() * 8));
        std::ifstream g(dir + "/l2b.bin", std::ios::binary);
        if (!g.read(reinterpret_cast<char *>(l2b.data()), 56)) return 3;
    }
    const int n_nodes = (int)(poses.size() / 7);
    if (n_nodes < K + W + 2) {
        fprintf(stderr, "need %d nodes, %d exported\n", K + W + 2, n_nodes);
        return 3;
    }
    for (int k = 0; k < 7; ++k) cfg.lidar2body[k] = l2b[k];
    cfg.params.removal_interval = 1;
    cfg.verbose = false;
    pcl::PointCloud<pcl::PointXYZI> map0;
    if (!read_bin(dir + "/map.bin", map0)) return 3;
    const int NS = std::min(n_nodes, K + W + 8);  // (the deep pass announces up to six nodes ahead: as many more as were exported)
    std::vector<pcl::PointCloud<pcl::PointXYZI>> scans(NS);
    std::vector<geometry_msgs::Pose> odom(NS);
    for (int i = 0; i < NS; ++i) {
        char name[64];
        snprintf(name, sizeof(name), "/scan_%06d.bin", i);
        if (!read_bin(dir + name, scans[i])) return 3;
        const double *v = &poses[(size_t)i * 7];
        odom[i].position.x = v[0]; odom[i].position.y = v[1]; odom[i].position.z = v[2];
        odom[i].orientation.x = v[3]; odom[i].orientation.y = v[4]; odom[i].orientation.z = v[5]; odom[i].orientation.w = v[6];
    }
    double ms_cb[4] = {0, 0, 0, 0}, ms_announce = 0;
    unsigned long long rejected[4] = {0, 0, 0, 0}, map_out[4] = {0, 0, 0, 0};
    const int deep_la = std::max(1, std::min(6, NS - (K + W)));
    for (int pass = 0; pass < 4; ++pass) {
        erasor::OfflineMapUpdater updater(cfg);
        updater.set_global_map(map0);
        double t0 = 0;
        uint64_t ticket = 0, next_ticket = 0;
        std::vector<uint64_t> tk(NS, 0);
        int announced_upto = 0;  // (pass 3: nodes [1, announced_upto] are announced)
        if (pass == 3) {
            updater.set_lookahead(deep_la);
            tk[0] = updater.announce_upcoming(scans[0], odom[0]);  // (a first callback without a ticket would drop the nodes announced behind it)
        }
        for (int i = 0; i < W + K; ++i) {
            if (i == W) t0 = now_ms();
            if (pass == 3) {  // round 6: as many nodes ahead as the updater takes, in node order, each stepped by its ticket
                while (announced_upto < std::min(i + deep_la, NS - 1) && updater.outstanding() < updater.lookahead()) {
                    ++announced_upto;
                    if (i > 0 && announced_upto == i + deep_la) {  // the pipeline is full: the round's one new node is staged beside this callback's step
                        updater.announce_next_deferred(announced_upto, scans[announced_upto], odom[announced_upto]);
                        break;
                    }
                    tk[announced_upto] = updater.announce_next(scans[announced_upto], odom[announced_upto]);
                }
                updater.callback_node(i, odom[i], scans[i], tk[i]);
                if (i >= W) rejected[pass] += updater.map_rejected.size();
                continue;
            }
            if (pass == 1) {
                const double ta = now_ms();
                next_ticket = updater.announce_next(scans[i + 1], odom[i + 1]);
                if (i >= W) ms_announce += now_ms() - ta;
            } else if (pass == 2)
                updater.announce_next_deferred(i + 1, scans[i + 1], odom[i + 1]);  // (staged inside callback_node(i), beside its step)
            updater.callback_node(i, odom[i], scans[i], pass == 1 ? ticket : 0);
            ticket = next_ticket;
            if (i >= W) rejected[pass] += updater.map_rejected.size();  // (the host copies of what the node publishes)
        }
        ms_cb[pass] = (now_ms() - t0) / K;
        map_out[pass] = updater.last.n_map_out;
    }
    // pass 3: the C ABI, device-resident scans, nodes announced two ahead
    double ms_dev = 0;
    unsigned long long rejected_dev = 0, map_out_dev = 0;
    {
        erasor_hip_handle *h = nullptr;
        if (erasor_hip_create(&cfg.params, 0, &h) != ERASOR_OK) return 4;
        std::vector<float> buf(map0.size() * 4);
        for (size_t i = 0; i < map0.size(); ++i) { buf[4 * i] = map0.points[i].x; buf[4 * i + 1] = map0.points[i].y; buf[4 * i + 2] = map0.points[i].z; buf[4 * i + 3] = map0.points[i].intensity; }
        if (erasor_hip_set_map(h, buf.data(), map0.size()) != ERASOR_OK) return 4;
        std::vector<void *> d_scan(K + W + 2, nullptr);
        std::vector<size_t> n_scan(K + W + 2);
        std::vector<std::array<float, 16>> Tb(K + W + 2), To(K + W + 2);
        float Tl[16];
        geometry_msgs::Pose l2bp;
        l2bp.position.x = l2b[0]; l2bp.position.y = l2b[1]; l2bp.position.z = l2b[2];
        l2bp.orientation.x = l2b[3]; l2bp.orientation.y = l2b[4]; l2bp.orientation.z = l2b[5]; l2bp.orientation.w = l2b[6];
        const Eigen::Matrix4f TL = erasor_utils::geoPose2eigen(l2bp);
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Tl[4 * r + c] = TL(r, c);
        for (int i = 0; i < K + W + 2; ++i) {
            n_scan[i] = scans[i].size();
            buf.resize(n_scan[i] * 4);
            for (size_t j = 0; j < n_scan[i]; ++j) { buf[4 * j] = scans[i].points[j].x; buf[4 * j + 1] = scans[i].points[j].y; buf[4 * j + 2] = scans[i].points[j].z; buf[4 * j + 3] = scans[i].points[j].intensity; }
            if (erasor_hip_device_alloc(h, n_scan[i] * 16, &d_scan[i]) != ERASOR_OK || erasor_hip_device_upload(h, d_scan[i], buf.data(), n_scan[i] * 16) != ERASOR_OK) return 4;
            const Eigen::Matrix4f T = erasor_utils::geoPose2eigen(odom[i]), Ti = erasor_utils::inverse(T);
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { Tb[i][4 * r + c] = T(r, c); To[i][4 * r + c] = Ti(r, c); }
        }
        for (int j = 0; j < 2; ++j) {
            erasor_hip_prefetch_node(h, d_scan[j], n_scan[j], 1, Tl, Tb[j].data());
            erasor_hip_announce_origin2body(h, To[j].data());
        }
        double t0 = 0;
        erasor_step_result res;
        for (int i = 0; i < W + K; ++i) {
            if (i == W) t0 = now_ms();
            if (i + 2 < W + K + 2) {
                erasor_hip_prefetch_node(h, d_scan[i + 2], n_scan[i + 2], 1, Tl, Tb[i + 2].data());
                erasor_hip_announce_origin2body(h, To[i + 2].data());
            }
            if (erasor_hip_step_device(h, d_scan[i], n_scan[i], Tl, Tb[i].data(), To[i].data(), &res) != ERASOR_OK) {
                fprintf(stderr, "step %d: %s\n", i, erasor_hip_last_error(h));
                return 4;
            }
            if (i >= W) rejected_dev += res.n_map_rejected;
        }
        ms_dev = (now_ms() - t0) / K;
        map_out_dev = res.n_map_out;
        for (void *p : d_scan) erasor_hip_device_free(h, p);
        erasor_hip_destroy(h);
    }
    const bool same = rejected[0] == rejected[1] && rejected[0] == rejected[2] && rejected[0] == rejected[3] && rejected[0] == rejected_dev &&
                      map_out[0] == map_out[1] && map_out[0] == map_out[2] && map_out[0] == map_out[3] && map_out[0] == map_out_dev;
    printf("{\"bench\": \"erasor_offline_demo --bench (C++, no Python in the loop)\", \"nodes_timed\": %d, \"warmup\": %d, \"map_points\": %zu, "
           "\"scan_points\": %zu, \"ms_per_callback\": %.4f, \"ms_per_callback_next_node_announced\": %.4f, "
           "\"of_which_announce_next\": %.4f, \"ms_per_callback_next_node_announced_deferred\": %.4f, \"ms_per_callback_nodes_announced_deep\": %.4f, \"deep_lookahead\": %d, \"ms_per_step_device_resident_two_ahead\": %.4f, \"callback_note\": \"OfflineMapUpdater::callback_node: host PointXYZI cloud in "
           "(32-byte records staged as they lie; an announced node is stepped by its ticket), map_rejected / query_rejected clouds copied back to the host\", "
           "\"map_rejected_points\": %llu, \"final_map_points\": %llu, \"passes_agree\": %s}\n",
           K, W, map0.size(), scans[W].size(), ms_cb[0], ms_cb[1], ms_announce / K, ms_cb[2], ms_cb[3], deep_la, ms_dev, rejected[0], map_out[0], same ? "true" : "false");
    return same ? 0 : 5;
}

int main(int argc, char **argv) {
    // an APPLICATION's decision (the library only warns): hardware queues for several updaters' streams, before HIP starts
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    if (argc >= 2 && std::string(argv[1]) == "--bench") {
        try {
            return bench_mode(argc, argv);
        } catch (const std::exception &e) {
            fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
    }
    if (argc >= 2 && (std::string(argv[1]) == "--queue" || std::string(argv[1]) == "--replicas")) {
        try {
            return std::string(argv[1]) == "--queue" ? queue_mode(argc, argv) : replicas_mode(argc, argv);
        } catch (const std::exception &e) {
            fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
    }
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
