// erasor_io.cpp — the ROS-free file side of the reference's drivers, host C++ only:
//   * rosparam YAML files in the reference's own layout (config/*.yaml: `erasor:`, `MapUpdater:`, `large_scale:`,
//     `tf: lidar2body`, top-level data_dir / init_idx / verbose) -> erasor::OfflineMapUpdater::Config, with exactly the
//     keys and defaults of set_params() (OMU.cpp:63-105) and ERASOR's constructor (erasor.h:47-61);
//   * poses_lidar2body.csv (main_in_your_env.cpp:33-59) and the pose -> erasor::node.odom round trip
//     (Eigen::Quaternionf::toRotationMatrix in float, then erasor_utils::eigen2geoPose = tf::Matrix3x3::getRotation in
//     double, utils.cpp:6-33);
//   * .pcd files: ASCII, binary (any field layout: SIZE/TYPE/COUNT honoured) and binary_compressed (PCL's LZF, SoA);
//     writers for ASCII (pcl::io::savePCDFileASCII, OMU.cpp:193: 8 significant digits) and binary.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "erasor_shim.h"

namespace {
typedef pcl::PointCloud<pcl::PointXYZI> Cloud;

std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
    return s.substr(a, b - a);
}
// strips a trailing comment (outside quotes)
std::string strip_comment(const std::string &s) {
    bool q = false;
    char qc = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        if (!q && (s[i] == '"' || s[i] == '\'')) { q = true; qc = s[i]; }
        else if (q && s[i] == qc) q = false;
        else if (!q && s[i] == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
    }
    return s;
}
std::string unquote(const std::string &v) {
    if (v.size() >= 2 && ((v.front() == '"' && v.back() == '"') || (v.front() == '\'' && v.back() == '\''))) return v.substr(1, v.size() - 2);
    return v;
}
// the subset of YAML rosparam files of the reference use: nested maps by indentation, scalars, flow lists of scalars
bool parse_yaml(const std::string &path, std::map<std::string, std::string> &kv) {
    std::ifstream f(path);
    if (!f) return false;
    std::vector<std::pair<int, std::string>> stack;  // (indent, key)
    std::string line;
    while (std::getline(f, line)) {
        line = strip_comment(line);
        if (trim(line).empty()) continue;
        int indent = 0;
        while (indent < (int)line.size() && line[indent] == ' ') ++indent;
        const std::string body = trim(line);
        const size_t colon = body.find(':');
        if (colon == std::string::npos) continue;
        const std::string key = trim(body.substr(0, colon));
        const std::string val = trim(body.substr(colon + 1));
        while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
        std::string full;
        for (const auto &s : stack) full += "/" + s.second;
        full += "/" + key;
        if (val.empty()) stack.emplace_back(indent, key);
        else kv[full] = unquote(val);
    }
    return true;
}
bool get_d(const std::map<std::string, std::string> &kv, const char *k, double &out) {
    auto it = kv.find(k);
    if (it == kv.end()) return false;
    out = atof(it->second.c_str());
    return true;
}
bool get_i(const std::map<std::string, std::string> &kv, const char *k, int &out) {
    auto it = kv.find(k);
    if (it == kv.end()) return false;
    out = atoi(it->second.c_str());
    return true;
}
bool get_b(const std::map<std::string, std::string> &kv, const char *k, bool &out) {
    auto it = kv.find(k);
    if (it == kv.end()) return false;
    const std::string &v = it->second;
    out = (v == "true" || v == "True" || v == "TRUE" || v == "1");
    return true;
}
bool get_s(const std::map<std::string, std::string> &kv, const char *k, std::string &out) {
    auto it = kv.find(k);
    if (it == kv.end()) return false;
    out = it->second;
    return true;
}

// ---- LZF (the codec of pcl::io::savePCDFileBinaryCompressed; format by Marc Lehmann: literal runs and back references)
bool lzf_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        const unsigned ctrl = in[ip++];
        if (ctrl < 32) {  // literal run of ctrl + 1 bytes
            const size_t n = ctrl + 1;
            if (ip + n > in_len || op + n > out_len) return false;
            memcpy(out + op, in + ip, n);
            ip += n;
            op += n;
        } else {  // back reference
            size_t len = ctrl >> 5;
            if (len == 7) {
                if (ip >= in_len) return false;
                len += in[ip++];
            }
            if (ip >= in_len) return false;
            const size_t off = ((size_t)(ctrl & 0x1f) << 8) + in[ip++] + 1;
            len += 2;
            if (off > op || op + len > out_len) return false;
            for (size_t i = 0; i < len; ++i, ++op) out[op] = out[op - off];  // may overlap: byte by byte
        }
    }
    return op == out_len;
}

struct PcdField {
    std::string name;
    int size = 4, count = 1;
    char type = 'F';
    size_t offset = 0;
};
double field_value(const uint8_t *p, const PcdField &f) {
    switch (f.type) {
    case 'F':
        if (f.size == 4) { float v; memcpy(&v, p, 4); return v; }
        if (f.size == 8) { double v; memcpy(&v, p, 8); return v; }
        break;
    case 'U':
        if (f.size == 1) return *p;
        if (f.size == 2) { uint16_t v; memcpy(&v, p, 2); return v; }
        if (f.size == 4) { uint32_t v; memcpy(&v, p, 4); return v; }
        break;
    case 'I':
        if (f.size == 1) return (int8_t)*p;
        if (f.size == 2) { int16_t v; memcpy(&v, p, 2); return v; }
        if (f.size == 4) { int32_t v; memcpy(&v, p, 4); return v; }
        break;
    }
    return 0.0;
}
}  // namespace

namespace erasor_utils {

// utils.cpp:6-33: tf::Matrix3x3::getRotation (double) of the float rotation block, translation copied
geometry_msgs::Pose eigen2geoPose(const Eigen::Matrix4f &pose) {
    double m[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) m[r][c] = (double)pose(r, c);
    double t[4];
    const double trace = m[0][0] + m[1][1] + m[2][2];
    if (trace > 0.0) {
        double s = std::sqrt(trace + 1.0);
        t[3] = s * 0.5;
        s = 0.5 / s;
        t[0] = (m[2][1] - m[1][2]) * s;
        t[1] = (m[0][2] - m[2][0]) * s;
        t[2] = (m[1][0] - m[0][1]) * s;
    } else {
        const int i = m[0][0] < m[1][1] ? (m[1][1] < m[2][2] ? 2 : 1) : (m[0][0] < m[2][2] ? 2 : 0);
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        double s = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        t[i] = s * 0.5;
        s = 0.5 / s;
        t[3] = (m[k][j] - m[j][k]) * s;
        t[j] = (m[j][i] + m[i][j]) * s;
        t[k] = (m[k][i] + m[i][k]) * s;
    }
    geometry_msgs::Pose g;
    g.orientation.x = t[0];
    g.orientation.y = t[1];
    g.orientation.z = t[2];
    g.orientation.w = t[3];
    g.position.x = pose(0, 3);
    g.position.y = pose(1, 3);
    g.position.z = pose(2, 3);
    return g;
}

int load_pcd(const std::string &pcd_name, Cloud &dst) {
    std::ifstream f(pcd_name, std::ios::binary);
    if (!f) {
        fprintf(stderr, "Couldn't read file!!! \n");  // utils.hpp:80
        return -1;
    }
    std::string line, mode;
    std::vector<PcdField> fields;
    size_t npts = 0, width = 0, height = 1;
    bool have_points = false;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key;
        ss >> key;
        if (key == "FIELDS" || key == "COLUMNS") {
            std::string t;
            while (ss >> t) {
                PcdField pf;
                pf.name = t;
                fields.push_back(pf);
            }
        } else if (key == "SIZE") {
            for (auto &pf : fields) ss >> pf.size;
        } else if (key == "TYPE") {
            for (auto &pf : fields) ss >> pf.type;
        } else if (key == "COUNT") {
            for (auto &pf : fields) ss >> pf.count;
        } else if (key == "WIDTH") {
            ss >> width;
        } else if (key == "HEIGHT") {
            ss >> height;
        } else if (key == "POINTS") {
            ss >> npts;
            have_points = true;
        } else if (key == "DATA") {
            ss >> mode;
            break;
        }
    }
    if (!have_points) npts = width * height;
    if (fields.empty() || mode.empty()) return -1;
    size_t stride = 0;
    int ix = -1, iy = -1, iz = -1, ii = -1;
    for (size_t k = 0; k < fields.size(); ++k) {
        if (fields[k].size <= 0 || fields[k].count < 0) return -1;
        fields[k].offset = stride;
        stride += (size_t)fields[k].size * (size_t)fields[k].count;
        if (fields[k].name == "x") ix = (int)k;
        if (fields[k].name == "y") iy = (int)k;
        if (fields[k].name == "z") iz = (int)k;
        if (fields[k].name == "intensity") ii = (int)k;
    }
    if (ix < 0 || iy < 0 || iz < 0) return -1;
    // the header is untrusted input: check POINTS / SIZE / COUNT against what the file can hold before allocating
    {
        const std::streampos data_pos = f.tellg();
        f.seekg(0, std::ios::end);
        const std::streampos end_pos = f.tellg();
        f.seekg(data_pos);
        if (data_pos < 0 || end_pos < data_pos) return -1;
        const size_t remaining = (size_t)(end_pos - data_pos);
        if (stride == 0 || npts > SIZE_MAX / stride) return -1;
        if (mode == "binary" && stride * npts > remaining) return -1;
        if (mode == "ascii" && npts > remaining) return -1;  // at least one byte per point
        if (mode == "binary_compressed" && remaining < 8) return -1;
        if (npts > (size_t)0x3FFFFFFF) return -1;
    }
    try {
        dst.points.resize(npts);
    } catch (const std::exception &) {
        return -1;
    }
    if (mode == "ascii") {
        size_t ncol = 0;
        for (const auto &pf : fields) ncol += (size_t)pf.count;
        std::vector<double> row(ncol);
        std::vector<size_t> col0(fields.size());
        size_t c = 0;
        for (size_t k = 0; k < fields.size(); ++k) { col0[k] = c; c += (size_t)fields[k].count; }
        std::string tok;
        for (size_t i = 0; i < npts; ++i) {
            for (size_t k = 0; k < ncol; ++k) {
                // strtod, not operator>>: PCL's ASCII writer emits "nan" for invalid points, which stream extraction rejects
                if (!(f >> tok)) return -1;
                char *endp = nullptr;
                row[k] = strtod(tok.c_str(), &endp);
                if (endp == tok.c_str()) return -1;
            }
            dst.points[i].x = (float)row[col0[ix]];
            dst.points[i].y = (float)row[col0[iy]];
            dst.points[i].z = (float)row[col0[iz]];
            dst.points[i].intensity = ii >= 0 ? (float)row[col0[ii]] : 0.f;
        }
    } else if (mode == "binary") {
        std::vector<uint8_t> buf(stride * npts);
        if (!buf.empty() && !f.read(reinterpret_cast<char *>(buf.data()), (std::streamsize)buf.size())) return -1;
        for (size_t i = 0; i < npts; ++i) {
            const uint8_t *p = buf.data() + i * stride;
            dst.points[i].x = (float)field_value(p + fields[ix].offset, fields[ix]);
            dst.points[i].y = (float)field_value(p + fields[iy].offset, fields[iy]);
            dst.points[i].z = (float)field_value(p + fields[iz].offset, fields[iz]);
            dst.points[i].intensity = ii >= 0 ? (float)field_value(p + fields[ii].offset, fields[ii]) : 0.f;
        }
    } else if (mode == "binary_compressed") {
        uint32_t csize = 0, usize = 0;
        if (!f.read(reinterpret_cast<char *>(&csize), 4) || !f.read(reinterpret_cast<char *>(&usize), 4)) return -1;
        if ((size_t)usize != stride * npts) return -1;
        {
            const std::streampos here = f.tellg();
            f.seekg(0, std::ios::end);
            const std::streampos endp2 = f.tellg();
            f.seekg(here);
            if (here < 0 || endp2 < here || (size_t)(endp2 - here) < (size_t)csize) return -1;
        }
        std::vector<uint8_t> cbuf, ubuf;
        try {
            cbuf.resize(csize);
            ubuf.resize(usize);
        } catch (const std::exception &) {
            return -1;
        }
        if (csize && !f.read(reinterpret_cast<char *>(cbuf.data()), csize)) return -1;
        if (!lzf_decompress(cbuf.data(), csize, ubuf.data(), usize)) return -1;
        // structure of arrays: all values of field 0, then field 1, ...
        std::vector<size_t> base(fields.size());
        size_t o = 0;
        for (size_t k = 0; k < fields.size(); ++k) { base[k] = o; o += (size_t)fields[k].size * (size_t)fields[k].count * npts; }
        for (size_t i = 0; i < npts; ++i) {
            auto at = [&](int k) { return ubuf.data() + base[k] + i * (size_t)fields[k].size * (size_t)fields[k].count; };
            dst.points[i].x = (float)field_value(at(ix), fields[ix]);
            dst.points[i].y = (float)field_value(at(iy), fields[iy]);
            dst.points[i].z = (float)field_value(at(iz), fields[iz]);
            dst.points[i].intensity = ii >= 0 ? (float)field_value(at(ii), fields[ii]) : 0.f;
        }
    } else {
        return -1;
    }
    dst.width = (unsigned)npts;
    dst.height = 1;
    return 0;
}

int save_pcd_ascii(const std::string &pcd_name, const Cloud &src) {
    FILE *fp = fopen(pcd_name.c_str(), "w");
    if (!fp) return -1;
    fprintf(fp, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n");
    fprintf(fp, "WIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA ascii\n", src.size(), src.size());
    // PCDWriter::writeASCII's default precision is 8 significant digits (what savePCDFileASCII(file, cloud) uses)
    for (const auto &p : src.points) fprintf(fp, "%.8g %.8g %.8g %.8g\n", p.x, p.y, p.z, p.intensity);
    return fclose(fp) == 0 ? 0 : -1;
}

int save_pcd_binary(const std::string &pcd_name, const Cloud &src) {
    FILE *fp = fopen(pcd_name.c_str(), "wb");
    if (!fp) return -1;
    fprintf(fp, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n");
    fprintf(fp, "WIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n", src.size(), src.size());
    for (const auto &p : src.points) {
        const float v[4] = {p.x, p.y, p.z, p.intensity};
        if (fwrite(v, 4, 4, fp) != 4) {
            fclose(fp);
            return -1;
        }
    }
    return fclose(fp) == 0 ? 0 : -1;
}
}  // namespace erasor_utils

namespace erasor {

bool load_config_yaml(const std::string &path, OfflineMapUpdater::Config &cfg, DriverConfig *drv) {
    std::map<std::string, std::string> kv;
    if (!parse_yaml(path, kv)) return false;
    erasor_params &p = cfg.params;
    // ERASOR's constructor (erasor.h:47-61)
    get_d(kv, "/erasor/max_range", p.max_range);
    get_i(kv, "/erasor/num_rings", p.num_rings);
    get_i(kv, "/erasor/num_sectors", p.num_sectors);
    get_d(kv, "/erasor/max_h", p.max_h);
    get_d(kv, "/erasor/min_h", p.min_h);
    get_d(kv, "/erasor/th_bin_max_h", p.th_bin_max_h);
    get_d(kv, "/erasor/scan_ratio_threshold", p.scan_ratio_threshold);
    get_i(kv, "/erasor/num_lowest_pts", p.num_lowest_pts);
    get_i(kv, "/erasor/minimum_num_pts", p.minimum_num_pts);
    get_d(kv, "/erasor/rejection_ratio", p.rejection_ratio);
    get_d(kv, "/erasor/gf_dist_thr", p.gf_dist_thr);
    get_i(kv, "/erasor/gf_iter", p.gf_iter);
    get_i(kv, "/erasor/gf_num_lpr", p.gf_num_lpr);
    get_d(kv, "/erasor/gf_th_seeds_height", p.gf_th_seeds_height);
    get_d(kv, "/erasor/map_voxel_size", p.map_voxel_size);
    get_i(kv, "/erasor/version", p.version);
    // fetch_VoI's radius is the same key read a second time with ITS OWN default of 60 m (OMU.cpp:78; ERASOR's is 10 m,
    // erasor.h:47): a YAML without /erasor/max_range must not shrink the VoI to the R-POD's default
    {
        double voi = 60.0;
        get_d(kv, "/erasor/max_range", voi);
        p.voi_max_range = voi;
    }
    // set_params (OMU.cpp:63-105)
    get_d(kv, "/MapUpdater/query_voxel_size", p.query_voxel_size);
    get_i(kv, "/MapUpdater/removal_interval", p.removal_interval);
    get_s(kv, "/MapUpdater/data_name", cfg.data_name);
    get_s(kv, "/MapUpdater/env", cfg.environment);
    get_s(kv, "/MapUpdater/initial_map_path", cfg.initial_map_path);
    get_s(kv, "/MapUpdater/save_path", cfg.save_path);
    get_b(kv, "/large_scale/is_large_scale", cfg.is_large_scale);
    get_d(kv, "/large_scale/submap_size", cfg.submap_size);
    get_b(kv, "/verbose", cfg.verbose);
    std::string l2b;
    if (get_s(kv, "/tf/lidar2body", l2b)) {  // "[x, y, z, qx, qy, qz, qw]" (OMU.cpp:89-104)
        for (char &c : l2b)
            if (c == '[' || c == ']' || c == ',') c = ' ';
        std::istringstream ss(l2b);
        double v[7];
        int n = 0;
        while (n < 7 && (ss >> v[n])) ++n;
        if (n != 7) return false;
        for (int i = 0; i < 7; ++i) cfg.lidar2body[i] = v[i];
    }
    if (drv) {  // main_in_your_env.cpp:66-70
        get_s(kv, "/data_dir", drv->data_dir);
        get_d(kv, "/voxel_size", drv->voxel_size);
        get_i(kv, "/init_idx", drv->init_idx);
        get_i(kv, "/interval", drv->interval);
    }
    return true;
}

// main_in_your_env.cpp:33-59: header line skipped; columns 2..8 = x y z qx qy qz qw, parsed with stof;
// Eigen::Quaternionf(w, x, y, z).toRotationMatrix() in float (no normalisation)
bool load_all_poses(const std::string &txt, std::vector<Eigen::Matrix4f> &poses) {
    poses.clear();
    std::ifstream in(txt);
    if (!in) return false;
    std::string line;
    int count = 0;
    while (std::getline(in, line)) {
        if (count++ == 0) continue;
        std::vector<float> v;
        std::stringstream ss(line);
        std::string t;
        while (std::getline(ss, t, ',')) {
            try {
                v.push_back(std::stof(t));
            } catch (...) {
                return false;
            }
        }
        if (v.size() < 9) {
            if (trim(line).empty()) continue;
            return false;
        }
        const float x = v[5], y = v[6], z = v[7], w = v[8];
        const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
        const float twx = tx * w, twy = ty * w, twz = tz * w;
        const float txx = tx * x, txy = ty * x, txz = tz * x;
        const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
        Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
        T(0, 0) = 1.f - (tyy + tzz); T(0, 1) = txy - twz; T(0, 2) = txz + twy;
        T(1, 0) = txy + twz; T(1, 1) = 1.f - (txx + tzz); T(1, 2) = tyz - twx;
        T(2, 0) = txz - twy; T(2, 1) = tyz + twx; T(2, 2) = 1.f - (txx + tyy);
        T(0, 3) = v[2]; T(1, 3) = v[3]; T(2, 3) = v[4];
        poses.push_back(T);
    }
    return true;
}
}  // namespace erasor

// ---- C entry points for the (ctypes) tests of the host-side file code: no GPU involved ---------------------------
extern "C" {
// parses `path`; writes the values as "key=value\n" lines (fixed order) into out (capacity cap); returns bytes or -1
int erasor_shim_dump_config(const char *path, char *out, int cap) {
    erasor::OfflineMapUpdater::Config cfg;
    erasor_hip_params_default(&cfg.params);
    cfg.params.query_voxel_size = 0.05;  // OMU.cpp:66
    cfg.params.removal_interval = 2;     // OMU.cpp:69
    erasor::DriverConfig drv;
    if (!erasor::load_config_yaml(path, cfg, &drv)) return -1;
    const erasor_params &p = cfg.params;
    std::ostringstream o;
    o.precision(17);
    o << "max_range=" << p.max_range << "\nnum_rings=" << p.num_rings << "\nnum_sectors=" << p.num_sectors << "\nmin_h=" << p.min_h
      << "\nmax_h=" << p.max_h << "\nth_bin_max_h=" << p.th_bin_max_h << "\nscan_ratio_threshold=" << p.scan_ratio_threshold
      << "\nnum_lowest_pts=" << p.num_lowest_pts << "\nminimum_num_pts=" << p.minimum_num_pts << "\nrejection_ratio=" << p.rejection_ratio
      << "\ngf_dist_thr=" << p.gf_dist_thr << "\ngf_iter=" << p.gf_iter << "\ngf_num_lpr=" << p.gf_num_lpr
      << "\ngf_th_seeds_height=" << p.gf_th_seeds_height << "\nmap_voxel_size=" << p.map_voxel_size << "\nversion=" << p.version
      << "\nquery_voxel_size=" << p.query_voxel_size << "\nremoval_interval=" << p.removal_interval << "\ndata_name=" << cfg.data_name
      << "\nenv=" << cfg.environment << "\ninitial_map_path=" << cfg.initial_map_path << "\nsave_path=" << cfg.save_path
      << "\nis_large_scale=" << (cfg.is_large_scale ? 1 : 0) << "\nsubmap_size=" << cfg.submap_size << "\nverbose=" << (cfg.verbose ? 1 : 0)
      << "\nvoi_max_range=" << p.voi_max_range << "\nlidar2body=";
    for (int i = 0; i < 7; ++i) o << (i ? "," : "") << cfg.lidar2body[i];
    o << "\ndata_dir=" << drv.data_dir << "\nvoxel_size=" << drv.voxel_size << "\ninit_idx=" << drv.init_idx << "\ninterval=" << drv.interval << "\n";
    const std::string s = o.str();
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}
// ERASOR::is_dynamic_obj_close (erasor.h:132) on an R-POD that carries only statuses (status[ring*S+sector]); no device needed
int erasor_shim_is_dynamic_obj_close(const erasor_params *p, const double *status, int r_target, int theta_target, int r_range, int theta_range) {
    ERASOR e(*p);
    R_POD pod((size_t)p->num_rings, Ring((size_t)p->num_sectors));
    for (int r = 0; r < p->num_rings; ++r)
        for (int t = 0; t < p->num_sectors; ++t) {
            Bin &b = pod[r][t];
            b.max_h = b.min_h = b.x = b.y = 0;
            b.is_occupied = false;
            b.status = status[(size_t)r * p->num_sectors + t];
        }
    return e.is_dynamic_obj_close(pod, r_target, theta_target, r_range, theta_range) ? 1 : 0;
}
// loads a .pcd; copies up to cap points (x y z intensity rows); returns the point count or -1
long erasor_shim_load_pcd(const char *path, float *xyzi, long cap) {
    Cloud c;
    if (erasor_utils::load_pcd(path, c) == -1) return -1;
    const long n = (long)c.size();
    for (long i = 0; i < n && i < cap; ++i) {
        xyzi[4 * i] = c.points[i].x;
        xyzi[4 * i + 1] = c.points[i].y;
        xyzi[4 * i + 2] = c.points[i].z;
        xyzi[4 * i + 3] = c.points[i].intensity;
    }
    return n;
}
int erasor_shim_save_pcd(const char *path, const float *xyzi, long n, int binary) {
    Cloud c;
    c.points.resize((size_t)n);
    for (long i = 0; i < n; ++i) {
        c.points[i].x = xyzi[4 * i];
        c.points[i].y = xyzi[4 * i + 1];
        c.points[i].z = xyzi[4 * i + 2];
        c.points[i].intensity = xyzi[4 * i + 3];
    }
    return binary ? erasor_utils::save_pcd_binary(path, c) : erasor_utils::save_pcd_ascii(path, c);
}
// poses file -> per pose: 16 floats (row-major tf4x4) + 7 doubles of eigen2geoPose (x y z qx qy qz qw) + 16 floats of
// geoPose2eigen(eigen2geoPose(T)) (what callback_node uses, OMU.cpp:219); returns the pose count or -1
long erasor_shim_load_poses(const char *path, float *T16, double *geo7, float *T16_roundtrip, long cap) {
    std::vector<Eigen::Matrix4f> poses;
    if (!erasor::load_all_poses(path, poses)) return -1;
    for (long i = 0; i < (long)poses.size() && i < cap; ++i) {
        const geometry_msgs::Pose g = erasor_utils::eigen2geoPose(poses[i]);
        const Eigen::Matrix4f R = erasor_utils::geoPose2eigen(g);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                T16[16 * i + 4 * r + c] = poses[i](r, c);
                T16_roundtrip[16 * i + 4 * r + c] = R(r, c);
            }
        const double v[7] = {g.position.x, g.position.y, g.position.z, g.orientation.x, g.orientation.y, g.orientation.z, g.orientation.w};
        memcpy(geo7 + 7 * i, v, sizeof(v));
    }
    return (long)poses.size();
}
}
