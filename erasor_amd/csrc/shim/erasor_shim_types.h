// erasor_shim_types.h — minimal stand-ins for the PCL / Eigen / geometry_msgs types the reference's public
// interface uses, for builds WITHOUT ROS/PCL (this image has neither).  With -DERASOR_SHIM_WITH_PCL the real
// headers are used instead and the shim classes compile against pcl::PointCloud<pcl::PointXYZI> unchanged.
#ifndef ERASOR_SHIM_TYPES_H
#define ERASOR_SHIM_TYPES_H

#ifdef ERASOR_SHIM_WITH_PCL
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <Eigen/Dense>
#include <geometry_msgs/Pose.h>
#else
#include <cstddef>
#include <memory>
#include <vector>

namespace pcl {
struct PointXYZI {  // same field names as pcl::PointXYZI; 16 B here (PCL pads to 32 B)
    float x = 0, y = 0, z = 0, intensity = 0;
};
template <class T>
struct PointCloud {
    typedef std::shared_ptr<PointCloud<T>> Ptr;             // boost::shared_ptr in PCL 1.8
    typedef std::shared_ptr<const PointCloud<T>> ConstPtr;
    std::vector<T> points;
    unsigned width = 0, height = 1;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void clear() { points.clear(); width = 0; }
    void reserve(size_t n) { points.reserve(n); }
    void push_back(const T &p) { points.push_back(p); width = (unsigned)points.size(); }
    T &operator[](size_t i) { return points[i]; }
    const T &operator[](size_t i) const { return points[i]; }
    PointCloud &operator+=(const PointCloud &o) {
        points.insert(points.end(), o.points.begin(), o.points.end());
        width = (unsigned)points.size();
        return *this;
    }
    PointCloud operator+(const PointCloud &o) const { PointCloud r = *this; r += o; return r; }
};
}  // namespace pcl

namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
}  // namespace geometry_msgs

namespace Eigen {
struct Matrix4f {  // row-major 4x4, (r,c) accessor like Eigen
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float &operator()(int r, int c) { return m[r * 4 + c]; }
    float operator()(int r, int c) const { return m[r * 4 + c]; }
    static Matrix4f Identity() { return Matrix4f(); }
};
}  // namespace Eigen
#endif
#endif
