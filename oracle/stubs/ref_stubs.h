// ref_stubs.h — stand-in ros / pcl / Eigen / tf / message headers, just wide enough to compile the
// reference's UNMODIFIED hot-path sources (erasor.cpp, erasor_utils.cpp, OfflineMapUpdater.cpp, mapgen.hpp)
// in a container that has none of those libraries (oracle/ref.mk -> oracle/_ref/liberasor_ref.so).
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing here is reference code: these are our own minimal types.
// All third-party ARITHMETIC delegates to oracle/third_party_restated.h (the only restated pieces):
//   pcl::transformPointCloud, pcl::computeMeanAndCovarianceMatrix, pcl::VoxelGrid, Eigen::JacobiSVD,
//   Eigen dense products (sequential mul/add, no FMA), Matrix4f::inverse (double cofactors, NOT Eigen's
//   SSE path), tf::Matrix3x3(tf::Quaternion).  pcl::KdTreeFLANN is an exact kd-tree with FLANN's
//   L2_Simple float distance and "lowest index wins" on exact ties (independent of the oracle's grid search).
// Plumbing that is not arithmetic (NodeHandle parameters, publishers, subscribers, PCD io) is replaced by
// in-process registries the driver (oracle/ref_driver.cpp) fills and reads.
#pragma once

#include <any>
#include <array>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "../third_party_restated.h"

// ------------------------------------------------------------------------------------------------
// boost (only shared_ptr / format-free use on this path)
// ------------------------------------------------------------------------------------------------
namespace boost {
using std::shared_ptr;
using std::make_shared;
}  // namespace boost

// ------------------------------------------------------------------------------------------------
// Eigen subset
// ------------------------------------------------------------------------------------------------
namespace Eigen {

enum DecompositionOptions { ComputeFullU = 0x04, ComputeThinU = 0x08, ComputeFullV = 0x10, ComputeThinV = 0x20 };

class Mat;

// `m.row(j) << a, b, c;`
class RowInit {
public:
    RowInit(Mat &m, int r) : m_(m), r_(r), c_(0) {}
    RowInit &operator<<(float v);
    RowInit &operator,(float v) { return (*this) << v; }

private:
    Mat &m_;
    int r_, c_;
};

// `m << a, b, c, ...;` (row-major fill)
class CommaInit {
public:
    CommaInit(Mat &m) : m_(m), k_(0) {}
    CommaInit &operator,(float v);

private:
    Mat &m_;
    int k_;
};

// dynamic float matrix, column-major like Eigen's default
class Mat {
public:
    CommaInit operator<<(float v) {
        CommaInit ci(*this);
        ci, v;
        return ci;
    }
    Mat() : r_(0), c_(0) {}
    Mat(int r, int c) : r_(r), c_(c), v_((size_t)r * c, 0.f) {}
    int rows() const { return r_; }
    int cols() const { return c_; }
    float &operator()(int i, int j) { return v_[(size_t)j * r_ + i]; }
    float operator()(int i, int j) const { return v_[(size_t)j * r_ + i]; }
    float &operator[](int i) { return v_[i]; }  // vectors
    float operator[](int i) const { return v_[i]; }
    float &operator()(int i) { return v_[i]; }
    float operator()(int i) const { return v_[i]; }
    Mat col(int j) const {
        Mat o(r_, 1);
        for (int i = 0; i < r_; ++i) o(i, 0) = (*this)(i, j);
        return o;
    }
    RowInit row(int i) { return RowInit(*this, i); }
    Mat transpose() const {
        Mat o(c_, r_);
        for (int i = 0; i < r_; ++i)
            for (int j = 0; j < c_; ++j) o(j, i) = (*this)(i, j);
        return o;
    }
    template <int N>
    Mat head() const {
        Mat o(N, 1);
        for (int i = 0; i < N; ++i) o[i] = v_[i];
        return o;
    }
    // dense product: every coefficient is a left-to-right chain of separate float mul / add
    // (what Eigen 3.3 evaluates for these small / column-at-a-time products without -mfma)
    Mat operator*(const Mat &b) const {
        if (c_ != b.r_) throw std::logic_error("Eigen stub: product size mismatch");
        Mat o(r_, b.c_);
        for (int j = 0; j < b.c_; ++j)
            for (int i = 0; i < r_; ++i) {
                float acc = (*this)(i, 0) * b(0, j);
                for (int k = 1; k < c_; ++k) acc = acc + (*this)(i, k) * b(k, j);
                o(i, j) = acc;
            }
        return o;
    }
    // general 4x4 inverse: double cofactors narrowed to float (restated, see header comment)
    Mat inverse() const {
        if (r_ != 4 || c_ != 4) throw std::logic_error("Eigen stub: inverse() only for 4x4");
        float rm[16], out[16];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) rm[i * 4 + j] = (*this)(i, j);
        if (orc::invert4(rm, out) != 0) throw std::runtime_error("Eigen stub: singular 4x4");
        Mat o(4, 4);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) o(i, j) = out[i * 4 + j];
        return o;
    }
    friend std::ostream &operator<<(std::ostream &os, const Mat &m) {
        for (int i = 0; i < m.r_; ++i) {
            for (int j = 0; j < m.c_; ++j) os << (j ? " " : "") << m(i, j);
            if (i + 1 < m.r_) os << "\n";
        }
        return os;
    }

protected:
    int r_, c_;
    std::vector<float> v_;
};

inline CommaInit &CommaInit::operator,(float v) {
    m_(k_ / m_.cols(), k_ % m_.cols()) = v;
    ++k_;
    return *this;
}
inline RowInit &RowInit::operator<<(float v) {
    m_(r_, c_++) = v;
    return *this;
}

template <int R, int C>
class Fixed : public Mat {
public:
    Fixed() : Mat(R, C) {}
    Fixed(const Mat &m) : Mat(m) {
        if (m.rows() != R || m.cols() != C) throw std::logic_error("Eigen stub: fixed-size mismatch");
    }
    Fixed &operator=(const Mat &m) {
        if (m.rows() != R || m.cols() != C) throw std::logic_error("Eigen stub: fixed-size mismatch");
        Mat::operator=(m);
        return *this;
    }
    static Fixed Identity() {
        Fixed o;
        for (int i = 0; i < (R < C ? R : C); ++i) o(i, i) = 1.f;
        return o;
    }
    static Fixed Zero() { return Fixed(); }
};

typedef Mat MatrixXf;
typedef Mat VectorXf;
typedef Fixed<4, 4> Matrix4f;
typedef Fixed<3, 3> Matrix3f;
typedef Fixed<4, 1> Vector4f;
typedef Fixed<3, 1> Vector3f;

namespace stub {
// every JacobiSVD the reference runs is logged here (least-singular vector = the plane normal of
// estimate_plane_, erasor.cpp:188-190) so the driver can expose per-iteration planes.
inline std::vector<std::array<float, 3>> &svd_normal_log() {
    static std::vector<std::array<float, 3>> v;
    return v;
}
}  // namespace stub

template <typename MatrixType>
class JacobiSVD {
public:
    JacobiSVD(const Mat &m, unsigned int /*options*/) {
        if (m.rows() != 3 || m.cols() != 3) throw std::logic_error("Eigen stub: JacobiSVD only for 3x3");
        float cov[9], U[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) cov[i * 3 + j] = m(i, j);
        orc::jacobi_svd3(cov, U, sv_);
        U_ = Mat(3, 3);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) U_(i, j) = U[i * 3 + j];
        stub::svd_normal_log().push_back({U_(0, 2), U_(1, 2), U_(2, 2)});
    }
    const Mat &matrixU() const { return U_; }

private:
    Mat U_;
    float sv_[3];
};

}  // namespace Eigen

// ------------------------------------------------------------------------------------------------
// ros subset
// ------------------------------------------------------------------------------------------------
namespace ros {

class Time {
public:
    Time() : t_(0) {}
    explicit Time(double t) : t_(t) {}
    static Time now() {
        using namespace std::chrono;
        return Time(duration<double>(steady_clock::now().time_since_epoch()).count());
    }
    double toSec() const { return t_; }

private:
    double t_;
};

namespace stub {
typedef std::variant<bool, int, double, std::string, std::vector<double>> Param;
inline std::map<std::string, Param> &params() {
    static std::map<std::string, Param> m;
    return m;
}
// published messages: last message per topic, kept only for the topics listed here — publishing is otherwise
// a no-op, like a publisher without subscribers
inline std::map<std::string, bool> &capture_topics() {
    static std::map<std::string, bool> c;
    return c;
}
inline std::map<std::string, std::any> &published() {
    static std::map<std::string, std::any> m;
    return m;
}
inline std::map<std::string, std::any> &subscribers() {
    static std::map<std::string, std::any> m;
    return m;
}
// ROS_INFO_STREAM sink (off by default)
inline std::function<void(const std::string &)> &log_sink() {
    static std::function<void(const std::string &)> f;
    return f;
}
template <class M>
bool dispatch(const std::string &topic, const boost::shared_ptr<const M> &msg) {
    auto it = subscribers().find(topic);
    if (it == subscribers().end()) return false;
    std::any_cast<std::function<void(const boost::shared_ptr<const M> &)>>(it->second)(msg);
    return true;
}
}  // namespace stub

class Publisher {
public:
    Publisher() {}
    explicit Publisher(const std::string &t) : topic_(t) {}
    template <class M>
    void publish(const M &msg) const {
        if (!stub::capture_topics().empty() && stub::capture_topics().count(topic_)) stub::published()[topic_] = msg;
    }
    const std::string &getTopic() const { return topic_; }
    uint32_t getNumSubscribers() const { return stub::capture_topics().count(topic_) ? 1u : 0u; }

private:
    std::string topic_;
};

class Subscriber {};

class NodeHandle {
public:
    NodeHandle() {}
    explicit NodeHandle(const std::string &) {}
    template <class T>
    bool param(const std::string &name, T &var, const T &def) const {
        if (getParam(name, var)) return true;
        var = def;
        return false;
    }
    template <class T>
    bool getParam(const std::string &name, T &var) const {
        auto it = stub::params().find(name);
        if (it == stub::params().end()) return false;
        if (const T *p = std::get_if<T>(&it->second)) {
            var = *p;
            return true;
        }
        if constexpr (std::is_same<T, double>::value) {  // rosparam converts int -> double
            if (const int *q = std::get_if<int>(&it->second)) {
                var = *q;
                return true;
            }
        }
        return false;
    }
    template <class M>
    Publisher advertise(const std::string &topic, uint32_t /*queue*/, bool /*latch*/ = false) {
        return Publisher(topic);
    }
    template <class M, class T>
    Subscriber subscribe(const std::string &topic, uint32_t /*queue*/, void (T::*fp)(const boost::shared_ptr<const M> &), T *obj) {
        std::function<void(const boost::shared_ptr<const M> &)> f = [fp, obj](const boost::shared_ptr<const M> &m) { (obj->*fp)(m); };
        stub::subscribers()[topic] = f;
        return Subscriber();
    }
};

inline void init(int &, char **, const std::string &) {}
inline void spin() {}
inline void spinOnce() {}
inline bool ok() { return true; }

}  // namespace ros

#define ROS_INFO_STREAM(args)                          \
    do {                                               \
        if (ros::stub::log_sink()) {                   \
            std::ostringstream ros_stub_ss__;          \
            ros_stub_ss__ << args;                     \
            ros::stub::log_sink()(ros_stub_ss__.str()); \
        }                                              \
    } while (0)
#define ROS_WARN_STREAM(args) ROS_INFO_STREAM(args)
#define ROS_ERROR_STREAM(args) ROS_INFO_STREAM(args)
#define ROS_INFO(...) \
    do {              \
    } while (0)

// ------------------------------------------------------------------------------------------------
// messages
// ------------------------------------------------------------------------------------------------
namespace std_msgs {
struct Header {
    uint32_t seq = 0;
    ros::Time stamp;
    std::string frame_id;
};
struct Float32 {
    float data = 0.f;
    typedef boost::shared_ptr<const Float32> ConstPtr;
};
struct Int32 {
    int32_t data = 0;
    typedef boost::shared_ptr<const Int32> ConstPtr;
};
}  // namespace std_msgs

namespace geometry_msgs {
struct Point {
    double x = 0, y = 0, z = 0;
};
struct Point32 {
    float x = 0, y = 0, z = 0;
};
struct Quaternion {
    double x = 0, y = 0, z = 0, w = 0;
};
struct Pose {
    Point position;
    Quaternion orientation;
};
struct PoseStamped {
    std_msgs::Header header;
    Pose pose;
};
struct Polygon {
    std::vector<Point32> points;
};
struct PolygonStamped {
    std_msgs::Header header;
    Polygon polygon;
};
}  // namespace geometry_msgs

namespace jsk_recognition_msgs {
struct PolygonArray {
    std_msgs::Header header;
    std::vector<geometry_msgs::PolygonStamped> polygons;
    std::vector<uint32_t> labels;
    std::vector<float> likelihood;
};
}  // namespace jsk_recognition_msgs

namespace nav_msgs {
struct Path {
    std_msgs::Header header;
    std::vector<geometry_msgs::PoseStamped> poses;
};
struct Odometry {
    std_msgs::Header header;
};
}  // namespace nav_msgs

namespace visualization_msgs {
struct Marker {
    std_msgs::Header header;
};
}  // namespace visualization_msgs

namespace sensor_msgs {
// payload = XYZI rows (the only point type on this path); the wire layout is out of scope
struct PointCloud2 {
    std_msgs::Header header;
    uint32_t height = 1, width = 0;
    bool is_dense = true;
    std::vector<float> xyzi;
};
struct CompressedImage {
    std_msgs::Header header;
};
struct NavSatFix {
    std_msgs::Header header;
};
}  // namespace sensor_msgs

namespace erasor {
struct node {  // msg/node.msg:1-7
    std_msgs::Header header;
    geometry_msgs::Pose odom;
    std::array<double, 36> odomCov{};
    sensor_msgs::PointCloud2 lidar;
    geometry_msgs::Pose lidarOdom;
    sensor_msgs::CompressedImage image;
    sensor_msgs::NavSatFix gps;
    typedef boost::shared_ptr<node> Ptr;
    typedef boost::shared_ptr<const node> ConstPtr;
};
}  // namespace erasor

// ------------------------------------------------------------------------------------------------
// tf subset (bullet LinearMath, double)
// ------------------------------------------------------------------------------------------------
namespace tf {
class Quaternion {
public:
    Quaternion() : x_(0), y_(0), z_(0), w_(1) {}
    Quaternion(double x, double y, double z, double w) : x_(x), y_(y), z_(z), w_(w) {}
    double getX() const { return x_; }
    double getY() const { return y_; }
    double getZ() const { return z_; }
    double getW() const { return w_; }
    double x() const { return x_; }
    double y() const { return y_; }
    double z() const { return z_; }
    double w() const { return w_; }
    void setValue(double x, double y, double z, double w) { x_ = x, y_ = y, z_ = z, w_ = w; }

private:
    double x_, y_, z_, w_;
};
class Vector3 {
public:
    double v[3] = {0, 0, 0};
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};
class Matrix3x3 {
public:
    Matrix3x3() {}
    explicit Matrix3x3(const Quaternion &q) {
        double m[9];
        orc::tf_quat_to_mat3(q.x(), q.y(), q.z(), q.w(), m);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r_[i][j] = m[i * 3 + j];
    }
    void setValue(double xx, double xy, double xz, double yx, double yy, double yz, double zx, double zy, double zz) {
        r_[0][0] = xx, r_[0][1] = xy, r_[0][2] = xz;
        r_[1][0] = yx, r_[1][1] = yy, r_[1][2] = yz;
        r_[2][0] = zx, r_[2][1] = zy, r_[2][2] = zz;
    }
    Vector3 &operator[](int i) { return r_[i]; }
    const Vector3 &operator[](int i) const { return r_[i]; }
    // Matrix3x3::getRotation (bullet): off the data path (only feeds the published path / pose echo)
    void getRotation(Quaternion &q) const {
        const double trace = r_[0][0] + r_[1][1] + r_[2][2];
        double t[4];
        if (trace > 0.0) {
            double s = std::sqrt(trace + 1.0);
            t[3] = s * 0.5;
            s = 0.5 / s;
            t[0] = (r_[2][1] - r_[1][2]) * s;
            t[1] = (r_[0][2] - r_[2][0]) * s;
            t[2] = (r_[1][0] - r_[0][1]) * s;
        } else {
            const int i = r_[0][0] < r_[1][1] ? (r_[1][1] < r_[2][2] ? 2 : 1) : (r_[0][0] < r_[2][2] ? 2 : 0);
            const int j = (i + 1) % 3, k = (i + 2) % 3;
            double s = std::sqrt(r_[i][i] - r_[j][j] - r_[k][k] + 1.0);
            t[i] = s * 0.5;
            s = 0.5 / s;
            t[3] = (r_[k][j] - r_[j][k]) * s;
            t[j] = (r_[j][i] + r_[i][j]) * s;
            t[k] = (r_[k][i] + r_[i][k]) * s;
        }
        q.setValue(t[0], t[1], t[2], t[3]);
    }

private:
    Vector3 r_[3];
};
class TransformBroadcaster {};
}  // namespace tf

// ------------------------------------------------------------------------------------------------
// pcl subset
// ------------------------------------------------------------------------------------------------
#define PCL_ERROR(...) fprintf(stderr, __VA_ARGS__)

namespace pcl {

struct alignas(16) PointXYZI {  // PCL layout: 32 bytes
    float x = 0.f, y = 0.f, z = 0.f, pad0_ = 1.f;
    float intensity = 0.f, pad1_[3] = {0.f, 0.f, 0.f};
};

struct PCLHeader {
    uint32_t seq = 0;
    uint64_t stamp = 0;
    std::string frame_id;
};

template <class PointT>
class PointCloud {
public:
    typedef boost::shared_ptr<PointCloud<PointT>> Ptr;
    typedef boost::shared_ptr<const PointCloud<PointT>> ConstPtr;
    typedef std::vector<PointT> VectorType;
    typedef typename VectorType::iterator iterator;
    typedef typename VectorType::const_iterator const_iterator;

    PCLHeader header;
    VectorType points;
    uint32_t width = 0, height = 0;
    bool is_dense = true;

    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void reserve(size_t n) { points.reserve(n); }
    void resize(size_t n) {
        points.resize(n);
        width = (uint32_t)n;
        height = 1;
    }
    void clear() {
        points.clear();
        width = 0;
        height = 0;
    }
    void push_back(const PointT &p) {
        points.push_back(p);
        width = (uint32_t)points.size();
        height = 1;
    }
    PointT &operator[](size_t i) { return points[i]; }
    const PointT &operator[](size_t i) const { return points[i]; }
    PointT &at(size_t i) { return points.at(i); }
    const PointT &at(size_t i) const { return points.at(i); }
    iterator begin() { return points.begin(); }
    iterator end() { return points.end(); }
    const_iterator begin() const { return points.begin(); }
    const_iterator end() const { return points.end(); }
    PointCloud &operator+=(const PointCloud &rhs) {  // pcl/point_cloud.h (1.8): append, width = size, height = 1
        if (rhs.header.stamp > header.stamp) header.stamp = rhs.header.stamp;
        // (self-append safe)
        const size_t nr = rhs.points.size();
        points.reserve(points.size() + nr);
        for (size_t i = 0; i < nr; ++i) points.push_back(rhs.points[i]);
        width = (uint32_t)points.size();
        height = 1;
        is_dense = is_dense && rhs.is_dense;
        return *this;
    }
    const PointCloud operator+(const PointCloud &rhs) { return (PointCloud(*this) += rhs); }
    Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};

inline orc::Pt to_pt(const PointXYZI &p) { return orc::Pt{p.x, p.y, p.z, p.intensity}; }
inline PointXYZI from_pt(const orc::Pt &p) {
    PointXYZI o;
    o.x = p.x, o.y = p.y, o.z = p.z, o.intensity = p.i;
    return o;
}
template <class PointT>
inline void to_cloud(const PointCloud<PointT> &c, orc::Cloud &o) {
    o.resize(c.size());
    for (size_t k = 0; k < c.size(); ++k) o[k] = to_pt(c.points[k]);
}

// pcl::transformPointCloud (<= 1.9 scalar formula), safe for &in == &out
template <class PointT>
void transformPointCloud(const PointCloud<PointT> &in, PointCloud<PointT> &out, const Eigen::Mat &T) {
    float t[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) t[i * 4 + j] = T(i, j);
    if (&in != &out) {
        out.header = in.header;
        out.is_dense = in.is_dense;
        out.width = in.width;
        out.height = in.height;
        out.points.reserve(in.points.size());
        out.points.assign(in.points.begin(), in.points.end());
    }
    for (size_t k = 0; k < out.points.size(); ++k) {
        const orc::Pt q = orc::transform_pt(to_pt(out.points[k]), t);
        out.points[k].x = q.x;
        out.points[k].y = q.y;
        out.points[k].z = q.z;
    }
}

template <class PointT>
unsigned int computeMeanAndCovarianceMatrix(const PointCloud<PointT> &cloud, Eigen::Matrix3f &cov, Eigen::Vector4f &centroid) {
    orc::Cloud c;
    to_cloud(cloud, c);
    float cv[9], mean[4];
    bool degenerate = false;
    const unsigned n = orc::mean_and_cov(c, cv, mean, &degenerate);
    // (an empty cloud leaves the outputs untouched in PCL = uninitialised in the reference; defined as zeros here)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) cov(i, j) = cv[i * 3 + j];
    for (int i = 0; i < 4; ++i) centroid[i] = mean[i];
    return n;
}

template <class PointT>
class VoxelGrid {
public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr &c) { input_ = c; }
    void setLeafSize(float lx, float /*ly*/, float /*lz*/) { leaf_ = lx; }
    void filter(PointCloud<PointT> &out) {
        orc::Cloud in;
        to_cloud(*input_, in);
        orc::VoxelGridOut vo;
        orc::voxel_grid(in, (double)leaf_, vo);
        if (vo.overflow) {  // "Leaf size is too small for the input dataset. Integer indices would overflow." output = *input_
            PointCloud<PointT> copy = *input_;
            out = copy;
            return;
        }
        out.header = input_->header;
        out.points.resize(vo.centroids.size());
        for (size_t k = 0; k < vo.centroids.size(); ++k) out.points[k] = from_pt(vo.centroids[k]);
        out.width = (uint32_t)out.points.size();
        out.height = 1;
        out.is_dense = true;
    }

private:
    typename PointCloud<PointT>::ConstPtr input_;
    float leaf_ = 0.f;
};

template <class PointT>
class PassThrough {
public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr &c) { input_ = c; }
    void setFilterFieldName(const std::string &f) { field_ = f; }
    void setFilterLimits(double lo, double hi) { lo_ = lo, hi_ = hi; }
    void setFilterLimitsNegative(bool n) { neg_ = n; }
    void filter(PointCloud<PointT> &out) {
        PointCloud<PointT> o;
        for (const auto &p : input_->points) {
            const float v = field_ == "x" ? p.x : field_ == "y" ? p.y : p.z;
            const bool in = !(v < lo_ || v > hi_);
            if (in != neg_) o.push_back(p);
        }
        out = o;
    }

private:
    typename PointCloud<PointT>::ConstPtr input_;
    std::string field_ = "z";
    double lo_ = -1e30, hi_ = 1e30;
    bool neg_ = false;
};

// exact kd-tree; distance = FLANN L2_Simple in float; exact ties -> lowest input index
template <class PointT>
class KdTreeFLANN {
public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr &c) {
        input_ = c;
        const size_t n = c->size();
        pts_.resize(n);
        for (size_t k = 0; k < n; ++k) pts_[k] = to_pt(c->points[k]);
        idx_.resize(n);
        for (size_t k = 0; k < n; ++k) idx_[k] = (uint32_t)k;
        nodes_.clear();
        nodes_.reserve(n / 4 + 16);
        if (n) build(0, n);
    }
    int nearestKSearch(const PointT &p, int k, std::vector<int> &k_indices, std::vector<float> &k_sqr_distances) const {
        if (k != 1) throw std::logic_error("pcl stub: only K = 1");
        if (pts_.empty()) return 0;
        const orc::Pt q = to_pt(p);
        float best = std::numeric_limits<float>::infinity();
        uint32_t best_i = 0xFFFFFFFFu;
        search(0, q, best, best_i);
        k_indices.resize(1);
        k_sqr_distances.resize(1);
        k_indices[0] = (int)best_i;
        k_sqr_distances[0] = best;
        return 1;
    }
    int radiusSearch(const PointT &p, double radius, std::vector<int> &idx, std::vector<float> &d2) const {  // dead branch OMU.cpp:401-430
        idx.clear();
        d2.clear();
        const orc::Pt q = to_pt(p);
        for (size_t k = 0; k < pts_.size(); ++k) {
            const float d = orc::l2_simple(q, pts_[k]);
            if ((double)d <= radius * radius) {
                idx.push_back((int)k);
                d2.push_back(d);
            }
        }
        return (int)idx.size();
    }

private:
    struct Node {
        uint32_t lo, hi;     // leaf: [lo,hi) into idx_
        int32_t left, right; // children (-1 = leaf)
        int axis;
        float split;
    };
    static float coord(const orc::Pt &p, int a) { return a == 0 ? p.x : a == 1 ? p.y : p.z; }
    int build(size_t lo, size_t hi) {
        const int me = (int)nodes_.size();
        nodes_.push_back(Node{(uint32_t)lo, (uint32_t)hi, -1, -1, 0, 0.f});
        if (hi - lo <= 12) return me;
        float mn[3] = {1e38f, 1e38f, 1e38f}, mx[3] = {-1e38f, -1e38f, -1e38f};
        for (size_t k = lo; k < hi; ++k)
            for (int a = 0; a < 3; ++a) {
                const float c = coord(pts_[idx_[k]], a);
                mn[a] = std::min(mn[a], c);
                mx[a] = std::max(mx[a], c);
            }
        int ax = 0;
        for (int a = 1; a < 3; ++a)
            if (mx[a] - mn[a] > mx[ax] - mn[ax]) ax = a;
        if (!(mx[ax] > mn[ax])) return me;  // all points identical: one (possibly big) leaf
        const size_t mid = lo + (hi - lo) / 2;
        std::nth_element(idx_.begin() + lo, idx_.begin() + mid, idx_.begin() + hi,
                         [&](uint32_t a, uint32_t b) { return coord(pts_[a], ax) < coord(pts_[b], ax); });
        const float split = coord(pts_[idx_[mid]], ax);
        // left: coord <= split (positions lo..mid-1 are <= split), right: coord >= split
        const int l = build(lo, mid);
        const int r = build(mid, hi);
        nodes_[me].left = l;
        nodes_[me].right = r;
        nodes_[me].axis = ax;
        nodes_[me].split = split;
        return me;
    }
    void search(int ni, const orc::Pt &q, float &best, uint32_t &best_i) const {
        const Node &nd = nodes_[ni];
        if (nd.left < 0) {
            for (uint32_t k = nd.lo; k < nd.hi; ++k) {
                const uint32_t pi = idx_[k];
                const float d = orc::l2_simple(q, pts_[pi]);
                if (d < best || (d == best && pi < best_i)) {
                    best = d;
                    best_i = pi;
                }
            }
            return;
        }
        const float qc = coord(q, nd.axis);
        const int near = qc < nd.split ? nd.left : nd.right;
        const int far = qc < nd.split ? nd.right : nd.left;
        search(near, q, best, best_i);
        // every point of the far side has |q.c - p.c| >= |q.c - split|, and float rounding is monotone, so
        // its float distance is >= s*s: prune only when that bound is strictly worse (ties must be visited)
        const float s = qc - nd.split;
        if (!(s * s > best)) search(far, q, best, best_i);
    }
    typename PointCloud<PointT>::ConstPtr input_;
    std::vector<orc::Pt> pts_;
    std::vector<uint32_t> idx_;
    std::vector<Node> nodes_;
};

// ROS <-> PCL conversion (payload copy)
template <class PointT>
void toROSMsg(const PointCloud<PointT> &cloud, sensor_msgs::PointCloud2 &msg) {
    msg.xyzi.resize(cloud.size() * 4);
    for (size_t k = 0; k < cloud.size(); ++k) {
        msg.xyzi[4 * k + 0] = cloud.points[k].x;
        msg.xyzi[4 * k + 1] = cloud.points[k].y;
        msg.xyzi[4 * k + 2] = cloud.points[k].z;
        msg.xyzi[4 * k + 3] = cloud.points[k].intensity;
    }
    msg.width = (uint32_t)cloud.size();
    msg.height = 1;
    msg.is_dense = cloud.is_dense;
    msg.header.frame_id = cloud.header.frame_id;
}
template <class PointT>
void fromROSMsg(const sensor_msgs::PointCloud2 &msg, PointCloud<PointT> &cloud) {
    const size_t n = msg.xyzi.size() / 4;
    cloud.points.resize(n);
    for (size_t k = 0; k < n; ++k) {
        cloud.points[k].x = msg.xyzi[4 * k + 0];
        cloud.points[k].y = msg.xyzi[4 * k + 1];
        cloud.points[k].z = msg.xyzi[4 * k + 2];
        cloud.points[k].intensity = msg.xyzi[4 * k + 3];
    }
    cloud.width = (uint32_t)n;
    cloud.height = 1;
    cloud.is_dense = msg.is_dense;
    cloud.header.frame_id = msg.header.frame_id;
}

namespace stub {
// in-memory "files" for pcl::io (the driver registers the initial map and reads the saved result)
inline std::map<std::string, PointCloud<PointXYZI>> &files() {
    static std::map<std::string, PointCloud<PointXYZI>> m;
    return m;
}
}  // namespace stub

namespace io {
template <class PointT>
int loadPCDFile(const std::string &name, PointCloud<PointT> &cloud) {
    auto it = stub::files().find(name);
    if (it == stub::files().end()) return -1;
    cloud = it->second;
    return 0;
}
template <class PointT>
int savePCDFileASCII(const std::string &name, const PointCloud<PointT> &cloud) {
    stub::files()[name] = cloud;
    return 0;
}
template <class PointT>
int savePCDFileBinary(const std::string &name, const PointCloud<PointT> &cloud) {
    stub::files()[name] = cloud;
    return 0;
}
}  // namespace io

}  // namespace pcl
