// third_party_restated.h — the ONLY restated third-party arithmetic of the oracle.
//
// *** TEST INFRASTRUCTURE ONLY (see erasor_oracle.cpp). ***
// These are restatements, from their published algorithms, of the pieces of PCL 1.8.1, Eigen 3.3.4 and tf
// that the reference calls on the hot path and that are NOT under /root/reference (un-vendored, unpinned
// dependencies: package.xml:15,26, CMakeLists.txt:36; versions = Ubuntu 18.04 / ROS Melodic, README.md:41-43).
// Two users:
//   * oracle/erasor_oracle.cpp        — the CPU restatement of the reference's own logic;
//   * oracle/stubs/ (ref_stubs.h)     — stand-in ros/pcl/Eigen/tf headers against which the reference's
//                                       UNMODIFIED sources are compiled into oracle/_ref (oracle/ref.mk).
// So "oracle == _ref" pins every line of reference-owned logic; what stays unpinned is exactly this file.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace orc {

struct Pt {
    float x, y, z, i;
};
typedef std::vector<Pt> Cloud;


// tf::Matrix3x3::setRotation(tf::Quaternion) (bullet LinearMath), all in double — utils.cpp:37-38
static inline void tf_quat_to_mat3(double x, double y, double z, double w, double m[9]) {
    const double d = x * x + y * y + z * z + w * w;  // q.length2()
    const double s = 2.0 / d;
    const double xs = x * s, ys = y * s, zs = z * s;
    const double wx = w * xs, wy = w * ys, wz = w * zs;
    const double xx = x * xs, xy = x * ys, xz = x * zs;
    const double yy = y * ys, yz = y * zs, zz = z * zs;
    m[0] = 1.0 - (yy + zz); m[1] = xy - wz;         m[2] = xz + wy;
    m[3] = xy + wz;         m[4] = 1.0 - (xx + zz); m[5] = yz - wx;
    m[6] = xz - wy;         m[7] = yz + wx;         m[8] = 1.0 - (xx + yy);
}


// general 4x4 inverse in double (cofactors), narrowed to float.  NOT a restatement of Eigen's SSE
// inverse (OMU.cpp:436): both oracle and device receive the same 16 floats from the caller.
static int invert4(const float Tf[16], float out[16]) {
    double m[16], inv[16];
    for (int k = 0; k < 16; ++k) m[k] = Tf[k];
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
    if (det == 0.0) return -1;
    det = 1.0 / det;
    for (int k = 0; k < 16; ++k) out[k] = (float)(inv[k] * det);
    return 0;
}

// pcl::transformPointCloud (PCL <= 1.9, dense cloud), float32, left-associated, no FMA:
//   x' = ((T00*x + T01*y) + T02*z) + T03   (OMU.cpp:240,436,447)
static inline Pt transform_pt(const Pt &p, const float T[16]) {
    Pt o;
    o.x = ((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3];
    o.y = ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7];
    o.z = ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11];
    o.i = p.i;
    return o;
}
static void transform_cloud(const Cloud &in, Cloud &out, const float T[16]) {
    Cloud tmp(in.size());
    for (size_t k = 0; k < in.size(); ++k) tmp[k] = transform_pt(in[k], T);
    out.swap(tmp);
}

// ---------------------------------------------------------------------------------------------
// PCL 1.8 VoxelGrid<PointXYZI> (downsample_all_data, min_points_per_voxel 0) — utils.cpp:88-91
// ---------------------------------------------------------------------------------------------
struct IdxPair {  // pcl::cloud_point_index_idx
    unsigned int idx;
    unsigned int pi;
    bool operator<(const IdxPair &o) const { return idx < o.idx; }
};

struct VoxelGridOut {
    bool overflow = false;          // dx*dy*dz > INT_MAX: output = input
    int min_b[3] = {0, 0, 0};
    int div_b[3] = {0, 0, 0};
    float inv_leaf = 0.f;
    std::vector<IdxPair> sorted;    // after std::sort
    std::vector<uint32_t> run_begin;  // per output voxel: [run_begin[v], run_begin[v+1]) into sorted
    Cloud centroids;                // intensity = averaged (before label reassignment)
};

static void voxel_grid(const Cloud &in, double leaf_d, VoxelGridOut &vo) {
    vo = VoxelGridOut();
    const size_t n = in.size();
    if (n == 0) {
        vo.run_begin.push_back(0);
        return;
    }
    const float leaf = (float)leaf_d;      // setLeafSize(float,float,float)
    const float inv = 1.0f / leaf;         // inverse_leaf_size_ = Array4f::Ones() / leaf_size_
    vo.inv_leaf = inv;
    float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float mx[3] = {-mn[0], -mn[1], -mn[2]};
    for (size_t k = 0; k < n; ++k) {  // getMinMax3D, dense
        const float c[3] = {in[k].x, in[k].y, in[k].z};
        for (int a = 0; a < 3; ++a) {
            mn[a] = (c[a] < mn[a]) ? c[a] : mn[a];
            mx[a] = (c[a] > mx[a]) ? c[a] : mx[a];
        }
    }
    int64_t d[3];
    for (int a = 0; a < 3; ++a) d[a] = static_cast<int64_t>((mx[a] - mn[a]) * inv) + 1;
    // PCL: (dx*dy*dz) > INT_MAX in int64.  Evaluated with saturation here: where PCL's own int64 product would wrap
    // (extents beyond ~2e6 voxels per axis: undefined behaviour in the reference) this reports the overflow it stands for.
    bool too_many = false;
    {
        int64_t prod = 1;
        for (int a = 0; a < 3; ++a) {
            if (d[a] <= 0 || d[a] > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) too_many = true;
            if (!too_many) {
                prod *= d[a];
                if (prod > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) too_many = true;
            }
        }
    }
    if (too_many) {
        vo.overflow = true;  // "Leaf size is too small ... Integer indices would overflow." -> output = *input_
        return;
    }
    int max_b[3];
    for (int a = 0; a < 3; ++a) {
        vo.min_b[a] = static_cast<int>(std::floor(mn[a] * inv));
        max_b[a] = static_cast<int>(std::floor(mx[a] * inv));
        vo.div_b[a] = max_b[a] - vo.min_b[a] + 1;
    }
    const int mul[3] = {1, vo.div_b[0], vo.div_b[0] * vo.div_b[1]};
    vo.sorted.resize(n);
    for (size_t k = 0; k < n; ++k) {
        const int ijk0 = static_cast<int>(std::floor(in[k].x * inv) - static_cast<float>(vo.min_b[0]));
        const int ijk1 = static_cast<int>(std::floor(in[k].y * inv) - static_cast<float>(vo.min_b[1]));
        const int ijk2 = static_cast<int>(std::floor(in[k].z * inv) - static_cast<float>(vo.min_b[2]));
        const int idx = ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2];
        vo.sorted[k].idx = static_cast<unsigned int>(idx);
        vo.sorted[k].pi = static_cast<unsigned int>(k);
    }
    std::sort(vo.sorted.begin(), vo.sorted.end(), std::less<IdxPair>());  // unstable; tie order = libstdc++ introsort
    size_t index = 0;
    while (index < n) {
        size_t i = index + 1;
        while (i < n && vo.sorted[i].idx == vo.sorted[index].idx) ++i;
        vo.run_begin.push_back((uint32_t)index);
        // CentroidPoint<PointXYZI>: float running sums in sorted order, each divided by (float)n
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        for (size_t li = index; li < i; ++li) {
            const Pt &p = in[vo.sorted[li].pi];
            sx += p.x;
            sy += p.y;
            sz += p.z;
            si += p.i;
        }
        const float cnt = static_cast<float>(i - index);
        Pt c;
        c.x = sx / cnt;
        c.y = sy / cnt;
        c.z = sz / cnt;
        c.i = si / cnt;
        vo.centroids.push_back(c);
        index = i;
    }
    vo.run_begin.push_back((uint32_t)n);
}

// exact 1-NN (float32 squared L2 as FLANN L2_Simple: ((0+dx*dx)+dy*dy)+dz*dz; lowest index on ties)
// over `in`, accelerated by the voxel runs of `vo` (cells of one leaf).  The result does not
// depend on the acceleration structure (every point inside the final search ball is examined).
static inline float l2_simple(const Pt &a, const Pt &b) {
    float r = 0.f;
    float d0 = a.x - b.x;
    r += d0 * d0;
    float d1 = a.y - b.y;
    r += d1 * d1;
    float d2 = a.z - b.z;
    r += d2 * d2;
    return r;
}

static uint32_t nn_index_grid(const Cloud &in, const VoxelGridOut &vo, const std::vector<uint32_t> &ukeys,
                              size_t v /*voxel whose centroid is queried*/) {
    const Pt &c = vo.centroids[v];
    const uint32_t key = ukeys[v];
    const int dx = vo.div_b[0], dy = vo.div_b[1], dz = vo.div_b[2];
    const int ci = (int)(key % (uint32_t)dx), cj = (int)((key / (uint32_t)dx) % (uint32_t)dy), ck = (int)(key / ((uint32_t)dx * (uint32_t)dy));
    const double L = 1.0 / (double)vo.inv_leaf;
    float best = std::numeric_limits<float>::infinity();
    uint32_t best_i = 0xFFFFFFFFu;
    const int maxrho = std::max(dx, std::max(dy, dz));
    for (int rho = 1;; ++rho) {
        // visit the shell of cells at Chebyshev distance exactly rho-? : simply (re)visit the whole
        // (2rho+1)^3 block minus the (2rho-1)^3 block already visited (rho==1: whole block).
        for (int kk = ck - rho; kk <= ck + rho; ++kk) {
            if (kk < 0 || kk >= dz) continue;
            for (int jj = cj - rho; jj <= cj + rho; ++jj) {
                if (jj < 0 || jj >= dy) continue;
                for (int ii = ci - rho; ii <= ci + rho; ++ii) {
                    if (ii < 0 || ii >= dx) continue;
                    if (rho > 1 && std::abs(ii - ci) < rho && std::abs(jj - cj) < rho && std::abs(kk - ck) < rho) continue;
                    const uint32_t q = (uint32_t)ii + (uint32_t)jj * (uint32_t)dx + (uint32_t)kk * (uint32_t)dx * (uint32_t)dy;
                    auto it = std::lower_bound(ukeys.begin(), ukeys.end(), q);
                    if (it == ukeys.end() || *it != q) continue;
                    const size_t u = (size_t)(it - ukeys.begin());
                    for (uint32_t li = vo.run_begin[u]; li < vo.run_begin[u + 1]; ++li) {
                        const uint32_t pi = vo.sorted[li].pi;
                        const float dd = l2_simple(c, in[pi]);
                        if (dd < best || (dd == best && pi < best_i)) {
                            best = dd;
                            best_i = pi;
                        }
                    }
                }
            }
        }
        if (rho >= maxrho) break;
        // conservative distance from c to the outside of the visited block
        const double cc[3] = {c.x, c.y, c.z};
        const int cidx[3] = {ci, cj, ck};
        double g = std::numeric_limits<double>::infinity();
        for (int a = 0; a < 3; ++a) {
            const double lo = (double)(vo.min_b[a] + cidx[a] - rho) * L;
            const double hi = (double)(vo.min_b[a] + cidx[a] + rho + 1) * L;
            const double margin = 1e-3 * L + 1e-6 * std::fabs(cc[a]);
            g = std::min(g, std::min(cc[a] - lo, hi - cc[a]) - margin);
        }
        if (best_i != 0xFFFFFFFFu && g > 0.0 && (double)best <= g * g) break;
    }
    return best_i;
}

// ---------------------------------------------------------------------------------------------
// pcl::computeMeanAndCovarianceMatrix (PCL 1.7-1.9, float) — erasor.cpp:186
// ---------------------------------------------------------------------------------------------
static unsigned mean_and_cov(const Cloud &c, float cov[9], float mean[4], bool *degenerate) {
    float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const size_t n = c.size();
    for (size_t k = 0; k < n; ++k) {
        const Pt &p = c[k];
        a[0] += p.x * p.x;
        a[1] += p.x * p.y;
        a[2] += p.x * p.z;
        a[3] += p.y * p.y;
        a[4] += p.y * p.z;
        a[5] += p.z * p.z;
        a[6] += p.x;
        a[7] += p.y;
        a[8] += p.z;
    }
    if (n == 0) {  // reference: outputs untouched (uninitialised).  Defined here as zeros.
        for (int k = 0; k < 9; ++k) cov[k] = 0.f;
        mean[0] = mean[1] = mean[2] = 0.f;
        mean[3] = 1.f;
        if (degenerate) *degenerate = true;
        return 0;
    }
    const float fn = static_cast<float>(n);
    for (int k = 0; k < 9; ++k) a[k] /= fn;
    mean[0] = a[6];
    mean[1] = a[7];
    mean[2] = a[8];
    mean[3] = 1.f;
    cov[0] = a[0] - a[6] * a[6];
    cov[1] = a[1] - a[6] * a[7];
    cov[2] = a[2] - a[6] * a[8];
    cov[4] = a[3] - a[7] * a[7];
    cov[5] = a[4] - a[7] * a[8];
    cov[8] = a[5] - a[8] * a[8];
    cov[3] = cov[1];
    cov[6] = cov[2];
    cov[7] = cov[5];
    return (unsigned)n;
}

// ---------------------------------------------------------------------------------------------
// Eigen 3.3 JacobiSVD<MatrixXf>(3x3, ComputeFullU) — erasor.cpp:188-190.  W row-major 3x3.
// ---------------------------------------------------------------------------------------------
struct Rot {
    float c, s;
};
static inline Rot make_jacobi(float x, float y, float z) {  // JacobiRotation::makeJacobi(x,y,z)
    Rot r;
    const float deno = 2.0f * std::fabs(y);
    if (deno < std::numeric_limits<float>::min()) {
        r.c = 1.f;
        r.s = 0.f;
    } else {
        const float tau = (x - z) / deno;
        const float w = std::sqrt(tau * tau + 1.0f);
        float t;
        if (tau > 0.f)
            t = 1.0f / (tau + w);
        else
            t = 1.0f / (tau - w);
        const float sign_t = t > 0.f ? 1.0f : -1.0f;
        const float n = 1.0f / std::sqrt(t * t + 1.0f);
        r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
        r.c = n;
    }
    return r;
}
// apply_rotation_in_the_plane(x, y, j): x' = c*x + s*y ; y' = -s*x + c*y
static inline void rot_apply(float &x, float &y, float c, float s) {
    const float xi = x, yi = y;
    x = c * xi + s * yi;
    y = -s * xi + c * yi;
}
static void jacobi_svd3(const float cov[9], float U[9], float sv[3]) {
    const float precision = 2.0f * std::numeric_limits<float>::epsilon();
    const float considerAsZero = std::numeric_limits<float>::min();
    float scale = 0.f;
    for (int k = 0; k < 9; ++k) scale = std::max(scale, std::fabs(cov[k]));  // cwiseAbs().maxCoeff()
    if (scale == 0.f) scale = 1.f;
    float W[9];
    for (int k = 0; k < 9; ++k) W[k] = cov[k] / scale;
    for (int k = 0; k < 9; ++k) U[k] = (k % 4 == 0) ? 1.f : 0.f;
    float maxDiag = std::max(std::fabs(W[0]), std::max(std::fabs(W[4]), std::fabs(W[8])));
    bool finished = false;
    int guard = 0;
    while (!finished && guard++ < 1000) {
        finished = true;
        for (int p = 1; p < 3; ++p) {
            for (int q = 0; q < p; ++q) {
                const float threshold = std::max(considerAsZero, precision * maxDiag);
                if (std::fabs(W[p * 3 + q]) > threshold || std::fabs(W[q * 3 + p]) > threshold) {
                    finished = false;
                    // real_2x2_jacobi_svd(W, p, q, &j_left, &j_right)
                    float m00 = W[p * 3 + p], m01 = W[p * 3 + q], m10 = W[q * 3 + p], m11 = W[q * 3 + q];
                    Rot rot1;
                    const float t = m00 + m11;
                    const float d = m10 - m01;
                    if (std::fabs(d) < std::numeric_limits<float>::min()) {
                        rot1.s = 0.f;
                        rot1.c = 1.f;
                    } else {
                        const float u = t / d;
                        const float tmp = std::sqrt(1.0f + u * u);
                        rot1.s = 1.0f / tmp;
                        rot1.c = u / tmp;
                    }
                    // m.applyOnTheLeft(0,1,rot1): rows 0,1 of m
                    if (!(rot1.c == 1.f && rot1.s == 0.f)) {
                        rot_apply(m00, m10, rot1.c, rot1.s);
                        rot_apply(m01, m11, rot1.c, rot1.s);
                    }
                    const Rot jr = make_jacobi(m00, m01, m11);
                    // j_left = rot1 * j_right.transpose();  transpose = (c, -s)
                    const float oc = jr.c, os = -jr.s;
                    Rot jl;
                    jl.c = rot1.c * oc - rot1.s * os;
                    jl.s = rot1.c * os + rot1.s * oc;
                    // W.applyOnTheLeft(p,q,j_left): rows p,q
                    if (!(jl.c == 1.f && jl.s == 0.f)) {
                        for (int col = 0; col < 3; ++col) rot_apply(W[p * 3 + col], W[q * 3 + col], jl.c, jl.s);
                        // U.applyOnTheRight(p,q,j_left.transpose()) -> rotation_in_the_plane(col p, col q, j_left)
                        for (int row = 0; row < 3; ++row) rot_apply(U[row * 3 + p], U[row * 3 + q], jl.c, jl.s);
                    }
                    // W.applyOnTheRight(p,q,j_right) -> rotation_in_the_plane(col p, col q, j_right.transpose() = (c,-s))
                    if (!(jr.c == 1.f && -jr.s == 0.f)) {
                        for (int row = 0; row < 3; ++row) rot_apply(W[row * 3 + p], W[row * 3 + q], jr.c, -jr.s);
                    }
                    maxDiag = std::max(maxDiag, std::max(std::fabs(W[p * 3 + p]), std::fabs(W[q * 3 + q])));
                }
            }
        }
    }
    for (int i = 0; i < 3; ++i) {
        const float a = W[i * 3 + i];
        sv[i] = std::fabs(a);
        if (a < 0.f)
            for (int row = 0; row < 3; ++row) U[row * 3 + i] = -U[row * 3 + i];
    }
    for (int i = 0; i < 3; ++i) sv[i] *= scale;
    for (int i = 0; i < 3; ++i) {  // selection sort, descending; maxCoeff = first maximum
        int pos = 0;
        float mx = sv[i];
        for (int k = i + 1; k < 3; ++k)
            if (sv[k] > mx) {
                mx = sv[k];
                pos = k - i;
            }
        if (mx == 0.f) break;
        if (pos) {
            pos += i;
            std::swap(sv[i], sv[pos]);
            for (int row = 0; row < 3; ++row) std::swap(U[row * 3 + i], U[row * 3 + pos]);
        }
    }
}

}  // namespace orc
