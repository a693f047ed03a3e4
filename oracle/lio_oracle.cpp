// =============================================================================
// oracle/lio_oracle.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY)
//
// A plain C++17 restatement of the reference's FastLIO scan-matching hot path.
// It is the *checker* for the HIP path and the `cpu_baseline` of bench.py.
// Nothing under lidar-slam-detection_amd/ (the product) may include, link or
// call this file; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg do.
//
// PARITY STATUS: the reference ships no tests, fixtures or golden vectors for
// this path and its own build (cmake, PCL, Boost,
// OpenCV) cannot run here.  Its sources for this path DO compile from where they
// lie once those dependencies are shimmed (oracle/ref_shims): oracle/Makefile
// builds, into oracle/_ref, the pieces (iVox, esti_plane, MTK / IKFoM filter:
// tests/test_oracle_vs_ref.py, tests/test_ikfom_vs_ref.py, fixtures under
// tests/golden/) and the whole translation units laserMapping.cpp +
// IMU_Processing.hpp + preprocess.cpp driven end to end
// (oracle/ref_fastlio.cpp, tests/test_fastlio_vs_ref.py), and this restatement
// is checked against them.  pcl::VoxelGrid itself is third-party source that is
// not in the tree; the stage is restated from PCL 1.9.1 voxel_grid.hpp semantics
// and PINNED to the PCL-derived filter the tree does hold -- ndt_omp's
// pclomp::VoxelGridCovariance::applyFilter, which carries pcl::VoxelGrid's box,
// overflow guard, keys and per-voxel f32 sums statement for statement
// (oracle/ref_voxelgrid_cov.cpp, tests/test_voxelgrid_vs_ref.py: bit-exact).  What that
// cannot pin: the order of the addends inside a voxel after pcl::VoxelGrid's
// std::sort of its index vector (input order is fixed here).
//
// Reference lines followed (paths relative to /root/reference/slam/mapping/fastlio):
//   voxel downsample ........ PCL 1.9.1 VoxelGrid::applyFilter, called at
//                             src/laserMapping.cpp:1206-1207
//   iVox map ................ include/ivox3d/ivox3d.h:139-171,179-210,231-261
//                             include/ivox3d/ivox3d_node.hpp:87-127
//                             include/ivox3d/eigen_types.h:73-76
//   esti_plane .............. include/common_lib.h:236-268
//   measurement model ....... src/laserMapping.cpp:813-982 (h_share_model_geometric)
//                             src/laserMapping.cpp:984-1023 (h_share_model)
//   map update .............. src/laserMapping.cpp:523-576 (map_incremental)
//   per-scan driver ......... src/laserMapping.cpp:1126-1345 (fastlio_main)
//   constants ............... src/laserMapping.cpp:1025-1124 (fastlio_init)
//   iterated ESKF ........... include/IKFoM_toolkit/esekfom/esekfom.hpp:1619-1931
//   manifold math ........... include/IKFoM_toolkit/mtk/src/mtkmath.hpp:142-174,235-288
//                             include/IKFoM_toolkit/mtk/types/SOn.hpp:233-245,284-297
//                             include/IKFoM_toolkit/mtk/types/S2.hpp:136-167,179-197,259-280
//   state layout ............ include/use-ikfom.hpp:12-21
//
// Floating-point contract (shared with the HIP kernels so that per-point
// results can be compared bit-for-bit): all per-point f32 arithmetic is written
// as explicit, sequential IEEE operations and this file is compiled with
// -ffp-contract=off (no FMA fusion); sqrt and division are correctly rounded.
// Neighbour SETS are the reference's, exact ties at the fifth-nearest distance
// included (IVox::closest); the five are then put in a canonical total order
// (d2, x, y, z): the reference only guarantees "element 0 is the nearest, the
// rest unordered" (ivox3d.h:160-165), of which the canonical order is one
// valid instance.
// =============================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <limits>
#include <list>
#include <unordered_map>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct P4 {
    float x, y, z, w;  // w = intensity
};

// ----------------------------------------------------------------------------
// PCL 1.9.1 VoxelGrid<PointT>::applyFilter (third-party, source not in tree).
// leaf -> inverse leaf in f32; bbox over finite points; int32 overflow guard
// returns the input unchanged; idx = (floor(p*inv) - min_b) . (1, dx, dx*dy);
// sort by idx; per-voxel centroid of all fields in f32 running sums, divided by
// (float)count; output in ascending idx.  std::sort is unstable in PCL, so the
// in-voxel summation order is unspecified there; this oracle fixes it to
// ascending input index (one valid realisation).
// ----------------------------------------------------------------------------
int voxel_downsample(const P4* in, int n, float leaf, std::vector<P4>& out) {
    out.clear();
    if (n <= 0) return 0;
    const float inv = 1.0f / leaf;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool any = false;
    for (int i = 0; i < n; i++) {
        if (!std::isfinite(in[i].x) || !std::isfinite(in[i].y) || !std::isfinite(in[i].z)) continue;
        any = true;
        mn[0] = std::min(mn[0], in[i].x); mx[0] = std::max(mx[0], in[i].x);
        mn[1] = std::min(mn[1], in[i].y); mx[1] = std::max(mx[1], in[i].y);
        mn[2] = std::min(mn[2], in[i].z); mx[2] = std::max(mx[2], in[i].z);
    }
    if (!any) return 0;
    int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1;
    int64_t dy = (int64_t)((mx[1] - mn[1]) * inv) + 1;
    int64_t dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) {  // PCL: warn and output = input
        out.assign(in, in + n);
        return n;
    }
    int minb[3], maxb[3], divb[3];
    for (int a = 0; a < 3; a++) {
        minb[a] = (int)std::floor(mn[a] * inv);
        maxb[a] = (int)std::floor(mx[a] * inv);
        divb[a] = maxb[a] - minb[a] + 1;
    }
    const int mul1 = divb[0], mul2 = divb[0] * divb[1];
    std::vector<std::pair<int, int>> iv;  // (voxel idx, point idx)
    iv.reserve(n);
    for (int i = 0; i < n; i++) {
        if (!std::isfinite(in[i].x) || !std::isfinite(in[i].y) || !std::isfinite(in[i].z)) continue;
        int i0 = (int)(std::floor(in[i].x * inv) - (float)minb[0]);
        int i1 = (int)(std::floor(in[i].y * inv) - (float)minb[1]);
        int i2 = (int)(std::floor(in[i].z * inv) - (float)minb[2]);
        iv.emplace_back(i0 + i1 * mul1 + i2 * mul2, i);
    }
    std::sort(iv.begin(), iv.end());  // (idx, point index): in-voxel order = input order
    size_t a = 0;
    while (a < iv.size()) {
        size_t b = a;
        float sx = 0, sy = 0, sz = 0, sw = 0;
        while (b < iv.size() && iv[b].first == iv[a].first) {
            const P4& p = in[iv[b].second];
            sx = sx + p.x; sy = sy + p.y; sz = sz + p.z; sw = sw + p.w;
            b++;
        }
        const float cnt = (float)(b - a);
        out.push_back({sx / cnt, sy / cnt, sz / cnt, sw / cnt});
        a = b;
    }
    return (int)out.size();
}

// ----------------------------------------------------------------------------
// iVox (Faster-LIO linear iVox): ivox3d.h / ivox3d_node.hpp
// ----------------------------------------------------------------------------
struct Key3 {
    int x, y, z;
    bool operator==(const Key3& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct Key3Hash {  // eigen_types.h:73-76
    size_t operator()(const Key3& v) const {
        return size_t(((v.x) * 73856093) ^ ((v.y) * 471943) ^ ((v.z) * 83492791)) % 10000000;
    }
};
struct Cand {
    float d2;
    P4 p;
};
inline bool cand_less(const Cand& a, const Cand& b) {  // canonical total order
    if (a.d2 != b.d2) return a.d2 < b.d2;
    if (a.p.x != b.p.x) return a.p.x < b.p.x;
    if (a.p.y != b.p.y) return a.p.y < b.p.y;
    return a.p.z < b.p.z;
}
inline float dist2(const P4& a, const P4& b) {  // ivox3d_node.hpp:12-15
    // (pt1.getVector3fMap() - pt2.getVector3fMap()).squaredNorm(): fixed size 3 -> Eigen's unrolled redux tree x0 + (x1 + x2)
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + (dy * dy + dz * dz);
}

struct IVoxNode {
    double created;  // travel distance stamped at creation (ivox3d.h:240)
    std::vector<P4> pts;
};

class IVox {
   public:
    using Cache = std::list<std::pair<Key3, IVoxNode>>;
    float res, inv_res;
    size_t capacity;
    double max_distance;
    std::vector<Key3> nearby;
    std::unordered_map<Key3, Cache::iterator, Key3Hash> grids;
    Cache cache;
    int stencil_id = 0;

    IVox(float r, int stencil, size_t cap, double maxd) : res(r), inv_res(1.0f / r), capacity(cap), max_distance(maxd) {
        set_stencil(stencil);
    }
    // ivox3d.h:179-210
    void set_stencil(int s) {
        stencil_id = s;
        nearby.clear();
        static const int n18[19][3] = {{0, 0, 0},  {-1, 0, 0}, {1, 0, 0},   {0, 1, 0},  {0, -1, 0}, {0, 0, -1}, {0, 0, 1},
                                       {1, 1, 0},  {-1, 1, 0}, {1, -1, 0},  {-1, -1, 0}, {1, 0, 1},  {-1, 0, 1}, {1, 0, -1},
                                       {-1, 0, -1}, {0, 1, 1},  {0, -1, 1},  {0, 1, -1}, {0, -1, -1}};
        static const int n26x[8][3] = {{1, 1, 1}, {-1, 1, 1}, {1, -1, 1}, {1, 1, -1}, {-1, -1, 1}, {-1, 1, -1}, {1, -1, -1}, {-1, -1, -1}};
        if (s == 1) {
            nearby.push_back({0, 0, 0});
        } else if (s == 7) {
            for (int i = 0; i < 7; i++) nearby.push_back({n18[i][0], n18[i][1], n18[i][2]});
        } else if (s == 19) {
            for (int i = 0; i < 19; i++) nearby.push_back({n18[i][0], n18[i][1], n18[i][2]});
        } else if (s == 27) {
            for (int i = 0; i < 19; i++) nearby.push_back({n18[i][0], n18[i][1], n18[i][2]});
            for (int i = 0; i < 8; i++) nearby.push_back({n26x[i][0], n26x[i][1], n26x[i][2]});
        } else if (s == 75) {  // "NEARBY74" generates 5x5x3 = 75 offsets incl. centre
            for (int i = -2; i <= 2; i++)
                for (int j = -2; j <= 2; j++)
                    for (int k = -1; k <= 1; k++) nearby.push_back({i, j, k});
        }
    }
    Key3 pos2grid(const P4& p) const {  // ivox3d.h:258-261 (round half away from zero)
        return {(int)std::round(p.x * inv_res), (int)std::round(p.y * inv_res), (int)std::round(p.z * inv_res)};
    }
    // ivox3d.h:231-256
    void add_points(const P4* pts, int n, double distance) {
        for (int i = 0; i < n; i++) {
            Key3 key = pos2grid(pts[i]);
            auto it = grids.find(key);
            if (it == grids.end()) {
                cache.push_front({key, IVoxNode{distance, {}}});
                grids.insert({key, cache.begin()});
                cache.front().second.pts.push_back(pts[i]);
            } else {
                it->second->second.pts.push_back(pts[i]);
                cache.splice(cache.begin(), cache, it->second);
                grids[key] = cache.begin();
            }
            if (grids.size() > capacity && (distance - cache.back().second.created) > max_distance) {
                grids.erase(cache.back().first);
                cache.pop_back();
            }
        }
    }
    // ivox3d.h:139-171.  Returns false and leaves `closest` UNTOUCHED when no
    // candidate lies inside the stencil within the range (ivox3d.h:152-154):
    // the caller's stale content survives, as in the reference.
    //
    // WHICH five: the reference's.  Every stencil voxel, in nearby_grids_ order,
    // appends its in-range points in push_back order and is cut to max_num by
    // std::nth_element on `dist` alone (ivox3d_node.hpp:107-127); the whole list
    // is cut to max_num the same way (ivox3d.h:156-164).  Candidates exactly as
    // far as the max_num-th nearest are kept or dropped by what libstdc++'s
    // introselect does to that sequence -- this file is compiled against the
    // same libstdc++ as oracle/_ref, so the calls are made literally.
    // ORDER of the five: the canonical total order (the reference leaves the
    // nearest in front and the rest as introselect happened to leave them; no
    // caller depends on it, common_lib.h:236-268 fits a plane to the set).
    // tie_mode 0 restores the earlier definition (the five smallest in
    // (d2, x, y, z)), which differs only when the fifth and sixth distances are
    // equal; tie_mode 2 keeps the reference's ORDER as well (nearest first, the
    // rest as introselect leaves them): esti_plane's QR then sees the rows in
    // the reference's order and the whole path follows the reference's own
    // build to rounding of the f64 sums.
    int tie_mode = 1;
    struct DistPoint {  // ivox3d_node.hpp:71-84: ordered by dist alone
        double dist;
        P4 p;
        bool operator<(const DistPoint& o) const { return dist < o.dist; }
    };
    // the reference's selection, literally: fills `cand` with the list GetClosestPoint ends with; returns the number of in-range candidates seen
    size_t select_as_reference(const P4& q, int max_num, double max_sq, std::vector<DistPoint>& cand) const {
        cand.clear();
        cand.reserve((size_t)max_num * nearby.size());
        Key3 key = pos2grid(q);
        size_t seen = 0;
        for (const Key3& d : nearby) {
            auto it = grids.find(Key3{key.x + d.x, key.y + d.y, key.z + d.z});
            if (it == grids.end()) continue;
            const size_t old_size = cand.size();
            for (const P4& p : it->second->second.pts) {  // ivox3d_node.hpp:111-116
                double dd = (double)dist2(p, q);
                if (dd < max_sq) cand.push_back({dd, p});
            }
            seen += cand.size() - old_size;
            if (old_size + (size_t)max_num >= cand.size()) {  // ivox3d_node.hpp:119-124
            } else {
                std::nth_element(cand.begin() + old_size, cand.begin() + old_size + max_num - 1, cand.end());
                cand.resize(old_size + max_num);
            }
        }
        if (cand.empty()) return 0;
        if (cand.size() <= (size_t)max_num) {  // ivox3d.h:156-161
        } else {
            std::nth_element(cand.begin(), cand.begin() + max_num - 1, cand.end());
            cand.resize(max_num);
        }
        std::nth_element(cand.begin(), cand.begin(), cand.end());  // ivox3d.h:162
        return seen;
    }
    // (on return scratch.size() == the in-range candidates seen: callers read the count off it)
    bool closest(const P4& q, std::vector<P4>& closest_pt, int max_num, double max_sq, std::vector<Cand>& scratch) const {
        scratch.clear();
        if (tie_mode == 0) {
            Key3 key = pos2grid(q);
            for (const Key3& d : nearby) {
                auto it = grids.find(Key3{key.x + d.x, key.y + d.y, key.z + d.z});
                if (it == grids.end()) continue;
                for (const P4& p : it->second->second.pts) {
                    float d2 = dist2(p, q);
                    if ((double)d2 < max_sq) scratch.push_back({d2, p});
                }
            }
            if (scratch.empty()) return false;
            size_t k = std::min((size_t)max_num, scratch.size());
            std::partial_sort(scratch.begin(), scratch.begin() + k, scratch.end(), cand_less);
            closest_pt.clear();
            for (size_t i = 0; i < k; i++) closest_pt.push_back(scratch[i].p);
            return true;
        }
        std::vector<DistPoint> cand;
        const size_t seen = select_as_reference(q, max_num, max_sq, cand);
        if (cand.empty()) return false;
        for (const DistPoint& c : cand) scratch.push_back({(float)c.dist, c.p});
        if (tie_mode != 2) std::sort(scratch.begin(), scratch.end(), cand_less);  // (tie_mode 2: the list as the reference returns it, order included)
        closest_pt.clear();
        for (const Cand& c : scratch) closest_pt.push_back(c.p);
        scratch.resize(seen);  // (callers read the number of candidates off the scratch list)
        return true;
    }
    // the same query with the list exactly as the reference returns it (nearest first, the rest as introselect leaves them): pinned against
    // the compiled ivox3d.h element by element (tests/test_oracle_vs_ref.py)
    bool closest_as_reference(const P4& q, std::vector<P4>& closest_pt, int max_num, double max_sq) const {
        std::vector<DistPoint> cand;
        select_as_reference(q, max_num, max_sq, cand);
        if (cand.empty()) return false;
        closest_pt.clear();
        for (const DistPoint& c : cand) closest_pt.push_back(c.p);
        return true;
    }
    size_t num_points() const {
        size_t s = 0;
        for (auto& kv : cache) s += kv.second.pts.size();
        return s;
    }
};

// ----------------------------------------------------------------------------
// esti_plane (common_lib.h:236-268): solve A n = -1 (5x3, f32) by column-pivoted
// Householder QR, normalise, reject if any |n.q + d| > threshold.
// The QR follows Eigen's ColPivHouseholderQR algorithm (pivot on the largest
// remaining column norm with LAPACK-style norm down-dating, Householder
// reflector H = I - tau v v^T with v0 = 1, rank from the pivot threshold) written
// as explicit sequential f32 operations.
// ----------------------------------------------------------------------------
bool esti_plane(float pabcd[4], const P4* pt, float threshold) {
    const int R = 5, C = 3;
    float A[R][C];
    float b[R];
    for (int j = 0; j < R; j++) {
        A[j][0] = pt[j].x; A[j][1] = pt[j].y; A[j][2] = pt[j].z;
        b[j] = -1.0f;
    }
    const float eps = 1.1920929e-07f;  // FLT_EPSILON
    float normUpd[C], normDir[C];
    int perm[C] = {0, 1, 2};
    float hcoef[C];
    float maxnorm = 0.f;
    for (int k = 0; k < C; k++) {
        // m_qr.col(k).norm(): fixed size 5 -> Eigen's completely unrolled redux is a balanced tree
        // (redux_novec_unroller splits [0,5) into [0,2) and [2,5), the latter into [2,3) and [3,5))
        const float s = (A[0][k] * A[0][k] + A[1][k] * A[1][k]) + (A[2][k] * A[2][k] + (A[3][k] * A[3][k] + A[4][k] * A[4][k]));
        normDir[k] = sqrtf(s);
        normUpd[k] = normDir[k];
        maxnorm = fmaxf(maxnorm, normDir[k]);
    }
    const float thr_helper = ((maxnorm * eps) * (maxnorm * eps)) / (float)R;
    const float downdate_thr = sqrtf(eps);
    int nonzero = C;
    float maxpivot = 0.f;
    for (int k = 0; k < C; k++) {
        int big = k;
        float bigv = normUpd[k];
        for (int j = k + 1; j < C; j++)
            if (normUpd[j] > bigv) { bigv = normUpd[j]; big = j; }
        const float big_sq = bigv * bigv;
        if (nonzero == C && big_sq < thr_helper * (float)(R - k)) nonzero = k;
        if (big != k) {
            for (int r = 0; r < R; r++) { float t = A[r][k]; A[r][k] = A[r][big]; A[r][big] = t; }
            { float t = normUpd[k]; normUpd[k] = normUpd[big]; normUpd[big] = t; }
            { float t = normDir[k]; normDir[k] = normDir[big]; normDir[big] = t; }
            { int t = perm[k]; perm[k] = perm[big]; perm[big] = t; }
        }
        // Householder on A[k..R-1][k]
        const float c0 = A[k][k];
        float tail = 0.f;
        for (int r = k + 1; r < R; r++) tail = tail + A[r][k] * A[r][k];
        float tau, beta;
        if (tail <= 1.17549435e-38f) {
            tau = 0.f; beta = c0;
            for (int r = k + 1; r < R; r++) A[r][k] = 0.f;
        } else {
            beta = sqrtf(c0 * c0 + tail);
            if (c0 >= 0.f) beta = -beta;
            const float den = c0 - beta;
            for (int r = k + 1; r < R; r++) A[r][k] = A[r][k] / den;
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
        hcoef[k] = tau;
        if (fabsf(beta) > maxpivot) maxpivot = fabsf(beta);
        // apply H to trailing columns and to b later
        if (tau != 0.f) {
            for (int j = k + 1; j < C; j++) {
                float t = 0.f;  // tmp = essential^T * bottom; tmp += row0  (Eigen Householder.h order)
                for (int r = k + 1; r < R; r++) t = t + A[r][k] * A[r][j];
                t = t + A[k][j];
                A[k][j] = A[k][j] - tau * t;
                for (int r = k + 1; r < R; r++) A[r][j] = A[r][j] - (tau * A[r][k]) * t;
            }
        }
        for (int j = k + 1; j < C; j++) {
            if (normUpd[j] != 0.f) {
                float t = fabsf(A[k][j]) / normUpd[j];
                t = (1.f + t) * (1.f - t);
                if (t < 0.f) t = 0.f;
                const float ratio = normUpd[j] / normDir[j];
                const float t2 = t * (ratio * ratio);
                if (t2 <= downdate_thr) {
                    float s = 0.f;
                    for (int r = k + 1; r < R; r++) s = s + A[r][j] * A[r][j];
                    normDir[j] = sqrtf(s);
                    normUpd[j] = normDir[j];
                } else {
                    normUpd[j] = normUpd[j] * sqrtf(t);
                }
            }
        }
    }
    (void)maxpivot;
    // c = Q^T b
    for (int k = 0; k < nonzero; k++) {
        const float tau = hcoef[k];
        if (tau != 0.f) {
            float t = 0.f;
            for (int r = k + 1; r < R; r++) t = t + A[r][k] * b[r];
            t = t + b[k];
            b[k] = b[k] - tau * t;
            for (int r = k + 1; r < R; r++) b[r] = b[r] - (tau * A[r][k]) * t;
        }
    }
    // triangularView<Upper>().solveInPlace: Eigen's column-major vector solve is column oriented --
    // x_i = b_i / a_ii, then b_r -= x_i * a_ri for the rows above (not the row-oriented dot-product form)
    float xs[C] = {0.f, 0.f, 0.f};
    for (int i = nonzero - 1; i >= 0; i--) {
        xs[i] = b[i] / A[i][i];
        for (int r = 0; r < i; r++) b[r] = b[r] - xs[i] * A[r][i];
    }
    float nv[C] = {0.f, 0.f, 0.f};
    for (int i = 0; i < nonzero; i++) nv[perm[i]] = xs[i];
    const float n = sqrtf(nv[0] * nv[0] + (nv[1] * nv[1] + nv[2] * nv[2]));  // normvec.norm(): fixed size 3 -> tree x0 + (x1 + x2)
    pabcd[0] = nv[0] / n;
    pabcd[1] = nv[1] / n;
    pabcd[2] = nv[2] / n;
    pabcd[3] = 1.0f / n;
    for (int j = 0; j < R; j++) {
        const float v = ((pabcd[0] * pt[j].x + pabcd[1] * pt[j].y) + pabcd[2] * pt[j].z) + pabcd[3];
        if (fabsf(v) > threshold) return false;
    }
    return true;
}

// ----------------------------------------------------------------------------
// small dense f64 matrix helper (row-major)
// ----------------------------------------------------------------------------
struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
    static Mat eye(int n) { Mat m(n, n); for (int i = 0; i < n; i++) m(i, i) = 1; return m; }
};
Mat mul(const Mat& A, const Mat& B) {
    Mat C(A.r, B.c);
    for (int i = 0; i < A.r; i++)
        for (int k = 0; k < A.c; k++) {
            const double v = A(i, k);
            if (v == 0.0) continue;
            for (int j = 0; j < B.c; j++) C(i, j) += v * B(k, j);
        }
    return C;
}
Mat transpose(const Mat& A) {
    Mat T(A.c, A.r);
    for (int i = 0; i < A.r; i++) for (int j = 0; j < A.c; j++) T(j, i) = A(i, j);
    return T;
}
// inverse by LU with partial pivoting (what Eigen's inverse() does for n > 4)
Mat inverse(const Mat& A) {
    const int n = A.r;
    Mat LU = A;
    std::vector<int> piv(n);
    for (int i = 0; i < n; i++) piv[i] = i;
    for (int k = 0; k < n; k++) {
        int p = k;
        double best = std::fabs(LU(k, k));
        for (int i = k + 1; i < n; i++) if (std::fabs(LU(i, k)) > best) { best = std::fabs(LU(i, k)); p = i; }
        if (p != k) { for (int j = 0; j < n; j++) std::swap(LU(k, j), LU(p, j)); std::swap(piv[k], piv[p]); }
        for (int i = k + 1; i < n; i++) {
            LU(i, k) /= LU(k, k);
            const double f = LU(i, k);
            for (int j = k + 1; j < n; j++) LU(i, j) -= f * LU(k, j);
        }
    }
    Mat X(n, n);
    for (int col = 0; col < n; col++) {
        std::vector<double> y(n);
        for (int i = 0; i < n; i++) {
            double s = (piv[i] == col) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) s -= LU(i, j) * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = y[i];
            for (int j = i + 1; j < n; j++) s -= LU(i, j) * X(j, col);
            X(i, col) = s / LU(i, i);
        }
    }
    return X;
}

// 3x3 helpers
struct V3 { double v[3]; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };
inline V3 cross(const V3& a, const V3& b) { return {{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}}; }
struct Quat { double x, y, z, w; };  // Eigen coeffs order
inline Quat qmul(const Quat& a, const Quat& b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Quat qconj(const Quat& q) { return {-q.x, -q.y, -q.z, q.w}; }
// Eigen QuaternionBase::_transformVector: uv = 2 (q.vec x v); v + w uv + q.vec x uv
inline V3 qrot(const Quat& q, const V3& v) {
    V3 qv{{q.x, q.y, q.z}};
    V3 uv = cross(qv, v);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    V3 c2 = cross(qv, uv);
    return {{(v[0] + q.w * uv[0]) + c2[0], (v[1] + q.w * uv[1]) + c2[1], (v[2] + q.w * uv[2]) + c2[2]}};
}
// Eigen QuaternionBase::toRotationMatrix
inline void qtoR(const Quat& q, double R[9]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
inline void hat(const V3& v, double H[9]) {
    H[0] = 0; H[1] = -v[2]; H[2] = v[1];
    H[3] = v[2]; H[4] = 0; H[5] = -v[0];
    H[6] = -v[1]; H[7] = v[0]; H[8] = 0;
}
inline void mm3(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j]; C[i * 3 + j] = s; }
}

// ---- so3_math.h Exp(ang_vel, dt) and the per-point compensation of UndistortPcl, in Eigen's scalar evaluation order
// (3-term dot products associate as x0 + (x1 + x2); checked bit for bit against oracle/_ref, tests/test_oracle_vs_ref.py)
inline void mm3_eig(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + (A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j]);
}
inline void so3_Exp_rodrigues(const V3& w, double dt, double R[9]) {  // so3_math.h:36-60
    const double n = std::sqrt(w[0] * w[0] + (w[1] * w[1] + w[2] * w[2]));
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (n > 0.0000001) {
        V3 ax{{w[0] / n, w[1] / n, w[2] / n}};
        double K[9], sK[9], KK[9];
        hat(ax, K);
        const double th = n * dt, sn = std::sin(th), cs = 1.0 - std::cos(th);
        for (int i = 0; i < 9; i++) sK[i] = cs * K[i];
        mm3_eig(sK, K, KK);
        for (int i = 0; i < 9; i++) R[i] = (R[i] + sn * K[i]) + KK[i];
    }
}
struct P4;
inline void undistort_point(const double headR[9], const V3& hvel, const V3& hpos, const V3& tacc, const V3& tgyr, double dt, float pxyz[3],
                            const V3& epos, const Quat& erot, const Quat& ril, const V3& til) {  // IMU_Processing.hpp:386-394
    double Rd[9], Ri[9];
    so3_Exp_rodrigues(tgyr, dt, Rd);
    mm3_eig(headR, Rd, Ri);
    V3 Pi{{(double)pxyz[0], (double)pxyz[1], (double)pxyz[2]}};
    V3 T_ei;
    for (int a = 0; a < 3; a++) T_ei[a] = hpos[a] + hvel[a] * dt + 0.5 * tacc[a] * dt * dt - epos[a];
    V3 pl = qrot(ril, Pi);
    for (int a = 0; a < 3; a++) pl[a] += til[a];
    V3 pw;
    for (int a = 0; a < 3; a++) pw[a] = (Ri[a * 3] * pl[0] + (Ri[a * 3 + 1] * pl[1] + Ri[a * 3 + 2] * pl[2])) + T_ei[a];
    V3 pe = qrot(qconj(erot), pw);
    for (int a = 0; a < 3; a++) pe[a] -= til[a];
    V3 pc = qrot(qconj(ril), pe);
    pxyz[0] = (float)pc[0]; pxyz[1] = (float)pc[1]; pxyz[2] = (float)pc[2];
}

const double kTol = 1e-11;  // MTK::tolerance<double>()

// mtkmath.hpp:142-174
inline void cos_sinc_sqrt(double x2, double& c, double& s) {
    const double t0 = 2.220446049250313e-16;  // epsilon<double>
    const double t2 = std::sqrt(t0);
    const double tn = std::sqrt(t2);
    if (x2 >= tn) { double x = std::sqrt(x2); c = std::cos(x); s = std::sin(x) / x; return; }
    static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
    double cosi = 1., sinc = 1.;
    double term = -1 / 2. * x2;
    for (int i = 0; i < 3; ++i) {
        cosi += term; term *= inv[2 * i];
        sinc += term; term *= -inv[2 * i + 1] * x2;
    }
    c = cosi; s = sinc;
}
// mtkmath.hpp:249-256: returns w, writes vec part
inline Quat mtk_exp(const V3& v, double scale) {
    const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double c, s;
    cos_sinc_sqrt(scale * scale * n2, c, s);
    const double m = s * scale;
    return {m * v[0], m * v[1], m * v[2], c};
}
// mtkmath.hpp:268-288 with plus_minus_periodicity = true, scale 2 (SOn.hpp:293-297)
inline V3 so3_log(const Quat& q) {
    double nv = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (nv < kTol) nv = kTol;
    const double s = 2.0 / nv * std::atan(nv / q.w);
    return {{s * q.x, s * q.y, s * q.z}};
}
// mtkmath.hpp:235-247
inline void A_matrix(const V3& v, double A[9]) {
    const double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double n = std::sqrt(sq);
    for (int i = 0; i < 9; i++) A[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (n < kTol) return;
    double H[9], HH[9];
    hat(v, H);
    mm3(H, H, HH);
    const double a = (1 - std::cos(n)) / sq, b = (1 - std::sin(n) / n) / sq;
    for (int i = 0; i < 9; i++) A[i] += a * H[i] + b * HH[i];
}

// S2<double, 98090, 10000, 1> (use-ikfom.hpp:8): length 9.809, S2_typ = 1
const double kS2Len = 98090.0 / 10000.0;
inline void S2_Bx(const V3& vec, double Bx[6]) {  // 3x2 row-major; S2.hpp:179-245 (typ 1 branch)
    const double L = kS2Len;
    if (vec[0] + L > kTol) {
        Bx[0] = -vec[1]; Bx[1] = -vec[2];
        Bx[2] = L - vec[1] * vec[1] / (L + vec[0]); Bx[3] = -vec[2] * vec[1] / (L + vec[0]);
        Bx[4] = -vec[2] * vec[1] / (L + vec[0]); Bx[5] = L - vec[2] * vec[2] / (L + vec[0]);
        for (int i = 0; i < 6; i++) Bx[i] /= L;
    } else {
        for (int i = 0; i < 6; i++) Bx[i] = 0;
        Bx[3] = -1;  // res(1,1)
        Bx[4] = 1;   // res(2,0)
    }
}
inline void S2_boxplus(V3& vec, const double d[2]) {  // S2.hpp:136-142
    double Bx[6];
    S2_Bx(vec, Bx);
    V3 Bu{{Bx[0] * d[0] + Bx[1] * d[1], Bx[2] * d[0] + Bx[3] * d[1], Bx[4] * d[0] + Bx[5] * d[1]}};
    Quat e = mtk_exp(Bu, 0.5);
    double R[9];
    qtoR(e, R);
    V3 o;
    for (int i = 0; i < 3; i++) o[i] = R[i * 3] * vec[0] + R[i * 3 + 1] * vec[1] + R[i * 3 + 2] * vec[2];
    vec = o;
}
inline void S2_boxminus(const V3& vec, const V3& other, double res[2]) {  // S2.hpp:144-167
    double H[9];
    hat(vec, H);
    V3 hv{{H[0] * other[0] + H[1] * other[1] + H[2] * other[2], H[3] * other[0] + H[4] * other[1] + H[5] * other[2],
           H[6] * other[0] + H[7] * other[1] + H[8] * other[2]}};
    const double v_sin = std::sqrt(hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2]);
    const double v_cos = vec[0] * other[0] + vec[1] * other[1] + vec[2] * other[2];
    const double theta = std::atan2(v_sin, v_cos);
    if (v_sin < kTol) {
        if (std::fabs(theta) > kTol) { res[0] = 3.1415926; res[1] = 0; }
        else { res[0] = 0; res[1] = 0; }
    } else {
        double Bx[6];
        S2_Bx(other, Bx);
        double Ho[9];
        hat(other, Ho);
        V3 t{{Ho[0] * vec[0] + Ho[1] * vec[1] + Ho[2] * vec[2], Ho[3] * vec[0] + Ho[4] * vec[1] + Ho[5] * vec[2],
              Ho[6] * vec[0] + Ho[7] * vec[1] + Ho[8] * vec[2]}};
        const double f = theta / v_sin;
        res[0] = f * (Bx[0] * t[0] + Bx[2] * t[1] + Bx[4] * t[2]);
        res[1] = f * (Bx[1] * t[0] + Bx[3] * t[1] + Bx[5] * t[2]);
    }
}
inline void S2_Nx_yy(const V3& vec, double N[6]) {  // 2x3; S2.hpp:259-264
    double Bx[6], H[9];
    S2_Bx(vec, Bx);
    hat(vec, H);
    const double f = 1 / kS2Len / kS2Len;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += Bx[k * 2 + i] * H[k * 3 + j];
            N[i * 3 + j] = f * s;
        }
}
inline void S2_Mx(const V3& vec, const double delta[2], double M[6]) {  // 3x2; S2.hpp:266-280
    double Bx[6], H[9];
    S2_Bx(vec, Bx);
    hat(vec, H);
    const double dn = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
    if (dn < kTol) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 2; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += H[i * 3 + k] * Bx[k * 2 + j];
                M[i * 2 + j] = -s;
            }
    } else {
        V3 Bu{{Bx[0] * delta[0] + Bx[1] * delta[1], Bx[2] * delta[0] + Bx[3] * delta[1], Bx[4] * delta[0] + Bx[5] * delta[1]}};
        // exp_delta built with scale scalar(1/2) == 0 (integer division, S2.hpp:277) -> identity rotation
        Quat e = mtk_exp(Bu, 0.0);
        double E[9], A[9], T1[9], T2[9];
        qtoR(e, E);
        A_matrix(Bu, A);
        double At[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At[i * 3 + j] = A[j * 3 + i];
        mm3(E, H, T1);
        mm3(T1, At, T2);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 2; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += T2[i * 3 + k] * Bx[k * 2 + j];
                M[i * 2 + j] = -s;
            }
    }
}

// state_ikfom (use-ikfom.hpp:12-21); 23 DoF: pos0 rot3 R_il6 t_il9 vel12 bg15 ba18 grav21
struct State {
    V3 pos;
    Quat rot;
    Quat ril;
    V3 til;
    V3 vel, bg, ba;
    V3 grav;
};
void state_from(const double* s, State& x) {
    for (int i = 0; i < 3; i++) x.pos[i] = s[i];
    x.rot = {s[3], s[4], s[5], s[6]};
    x.ril = {s[7], s[8], s[9], s[10]};
    for (int i = 0; i < 3; i++) { x.til[i] = s[11 + i]; x.vel[i] = s[14 + i]; x.bg[i] = s[17 + i]; x.ba[i] = s[20 + i]; x.grav[i] = s[23 + i]; }
}
void state_to(const State& x, double* s) {
    for (int i = 0; i < 3; i++) s[i] = x.pos[i];
    s[3] = x.rot.x; s[4] = x.rot.y; s[5] = x.rot.z; s[6] = x.rot.w;
    s[7] = x.ril.x; s[8] = x.ril.y; s[9] = x.ril.z; s[10] = x.ril.w;
    for (int i = 0; i < 3; i++) { s[11 + i] = x.til[i]; s[14 + i] = x.vel[i]; s[17 + i] = x.bg[i]; s[20 + i] = x.ba[i]; s[23 + i] = x.grav[i]; }
}
void state_boxplus(State& x, const double* d) {
    for (int i = 0; i < 3; i++) x.pos[i] += d[i];
    x.rot = qmul(x.rot, mtk_exp(V3{{d[3], d[4], d[5]}}, 0.5));
    x.ril = qmul(x.ril, mtk_exp(V3{{d[6], d[7], d[8]}}, 0.5));
    for (int i = 0; i < 3; i++) { x.til[i] += d[9 + i]; x.vel[i] += d[12 + i]; x.bg[i] += d[15 + i]; x.ba[i] += d[18 + i]; }
    S2_boxplus(x.grav, d + 21);
}
void state_boxminus(const State& x, const State& o, double* d) {
    for (int i = 0; i < 3; i++) d[i] = x.pos[i] - o.pos[i];
    V3 r = so3_log(qmul(qconj(o.rot), x.rot));
    V3 r2 = so3_log(qmul(qconj(o.ril), x.ril));
    for (int i = 0; i < 3; i++) { d[3 + i] = r[i]; d[6 + i] = r2[i]; }
    for (int i = 0; i < 3; i++) { d[9 + i] = x.til[i] - o.til[i]; d[12 + i] = x.vel[i] - o.vel[i]; d[15 + i] = x.bg[i] - o.bg[i]; d[18 + i] = x.ba[i] - o.ba[i]; }
    S2_boxminus(x.grav, o.grav, d + 21);
}

// symmetric 3x3 eigen decomposition (cyclic Jacobi, f64); columns of V are eigenvectors.
void eig3(const double Ain[9], double w[3], double V[9]) {
    double A[9];
    std::memcpy(A, Ain, sizeof(A));
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                const double apq = A[p * 3 + q];
                if (apq == 0.0) continue;
                const double app = A[p * 3 + p], aqq = A[q * 3 + q];
                const double theta = (aqq - app) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; k++) {
                    const double akp = A[k * 3 + p], akq = A[k * 3 + q];
                    A[k * 3 + p] = c * akp - s * akq;
                    A[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {
                    const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                    A[p * 3 + k] = c * apk - s * aqk;
                    A[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - s * vkq;
                    V[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
    // ascending order like SelfAdjointEigenSolver
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (w[j] < w[i]) {
                std::swap(w[i], w[j]);
                for (int k = 0; k < 3; k++) std::swap(V[k * 3 + i], V[k * 3 + j]);
            }
}

// ----------------------------------------------------------------------------
// The LIO engine: file-scope globals of laserMapping.cpp gathered in a struct.
// ----------------------------------------------------------------------------
struct DynShare {  // esekfom.hpp:80-90
    bool valid = true, converge = true;
    Mat h_x;  // N x 15
    std::vector<double> h;
};

struct PassLog {
    int knn;          // this pass redid the neighbour search
    int n_eff;        // effct_feat_num
    int valid;        // dyn_share.valid after h_share_model
    int degenerate;
    double sum_abs_res;
    double JtJ[36];   // top-left 6x6 of HTH (after degeneracy projection)
    double Jtr[6];    // first 6 of h_x^T h
    double dx[23];
};

const int kMaxPts = 100000;  // fixed arrays, laserMapping.cpp:86,103,122-124

struct Lio {
    // constants (fastlio_init, laserMapping.cpp:1025-1124)
    int max_iter = 4;
    float leaf_surf = 0.5f, leaf_map = 0.5f;
    double init_time = 0.1;       // INIT_TIME
    double laser_cov = 0.001;     // LASER_POINT_COV
    bool degenerate_detect_en = true, extrinsic_est_en = false;
    double limit[23];
    int threads = 1;

    IVox ivox;
    State x;
    Mat P;
    double travel = 0, first_lidar_time = 0;
    V3 last_pos_lid{{0, 0, 0}};
    bool flg_first_scan = true, flg_EKF_inited = false, is_degenerate = false;
    double res_mean_last = 0.05, total_residual = 0;
    int effct_feat_num = 0;
    float last_contri[3] = {0, 0, 0}, last_strong[3] = {0, 0, 0};  // test visibility: the sums of laserMapping.cpp:946-964 of the last pass
    double last_eigval[3] = {0, 0, 0};

    std::vector<P4> ds_body, ds_world;
    std::vector<std::vector<P4>> nearest;  // Nearest_Points: persists across scans
    std::vector<uint8_t> selected;         // point_selected_surf (init true)
    std::vector<P4> normvec;
    std::vector<float> res_last;
    std::vector<PassLog> log;
    int n_ds = 0;

    Lio(float res, int stencil, size_t cap, double maxd) : ivox(res, stencil, cap, maxd), P(Mat::eye(23)) {
        for (int i = 0; i < 23; i++) limit[i] = 0.001;
        selected.assign(kMaxPts, 1);
        normvec.assign(kMaxPts, P4{0, 0, 0, 0});
        res_last.assign(kMaxPts, 0.f);
        x.pos = {{0, 0, 0}}; x.rot = {0, 0, 0, 1}; x.ril = {0, 0, 0, 1};
        x.til = x.vel = x.bg = x.ba = {{0, 0, 0}};
        x.grav = {{kS2Len, 0, 0}};
        odom_start = odom_end = x;  // start_state_point = state_ikfom() (IMU_Processing.hpp:103)
    }

    P4 body_to_world(const State& s, const P4& pb) const {  // laserMapping.cpp:189-198 / 831-836
        V3 p{{(double)pb.x, (double)pb.y, (double)pb.z}};
        V3 pi = qrot(s.ril, p);
        pi[0] += s.til[0]; pi[1] += s.til[1]; pi[2] += s.til[2];
        V3 pw = qrot(s.rot, pi);
        return {(float)(pw[0] + s.pos[0]), (float)(pw[1] + s.pos[1]), (float)(pw[2] + s.pos[2]), pb.w};
    }

    // laserMapping.cpp:813-982
    void h_share_model_geometric(const State& s, DynShare& d) {
        total_residual = 0.0;
        const int n = n_ds;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
        {
            std::vector<Cand> scratch;
            scratch.reserve(512);
#ifdef _OPENMP
#pragma omp for
#endif
            for (int i = 0; i < n; i++) {
                const P4& pb = ds_body[i];
                P4 pw = body_to_world(s, pb);
                ds_world[i] = pw;
                auto& near = nearest[i];
                if (d.converge) {
                    ivox.closest(pw, near, 5, 5.0, scratch);
                    selected[i] = near.size() < 5 ? 0 : 1;
                }
                if (!selected[i]) continue;
                float pabcd[4];
                selected[i] = 0;
                if (esti_plane(pabcd, near.data(), 0.1f)) {
                    const float pd2 = ((pabcd[0] * pw.x + pabcd[1] * pw.y) + pabcd[2] * pw.z) + pabcd[3];
                    const double pbn = std::sqrt((double)pb.x * pb.x + ((double)pb.y * pb.y + (double)pb.z * pb.z));  // V3D::norm(): tree x0 + (x1 + x2)
                    // float s = 1 - 0.9 * fabs(pd2) / sqrt(p_body.norm());  (evaluated in double, stored float)
                    const float sc = (float)(1 - 0.9 * std::fabs((double)pd2) / std::sqrt(pbn));
                    if ((double)sc > 0.9) {
                        selected[i] = 1;
                        normvec[i] = {pabcd[0], pabcd[1], pabcd[2], pd2};
                        res_last[i] = std::fabs(pd2);
                    }
                }
            }
        }
        effct_feat_num = 0;
        std::vector<int> sel;
        for (int i = 0; i < n; i++)
            if (selected[i]) { sel.push_back(i); total_residual += res_last[i]; effct_feat_num++; }
        if (effct_feat_num < 1) { d.valid = false; return; }
        res_mean_last = total_residual / effct_feat_num;

        d.h_x = Mat(effct_feat_num, 15);
        d.h.assign(effct_feat_num, 0.0);
        for (int r = 0; r < effct_feat_num; r++) {
            const P4& lp = ds_body[sel[r]];
            const P4& np = normvec[sel[r]];
            V3 pbe{{(double)lp.x, (double)lp.y, (double)lp.z}};
            V3 pthis = qrot(s.ril, pbe);
            pthis[0] += s.til[0]; pthis[1] += s.til[1]; pthis[2] += s.til[2];
            V3 nv{{(double)np.x, (double)np.y, (double)np.z}};
            V3 Cv = qrot(qconj(s.rot), nv);
            V3 Av = cross(pthis, Cv);  // point_crossmat * C
            d.h_x(r, 0) = np.x; d.h_x(r, 1) = np.y; d.h_x(r, 2) = np.z;
            d.h_x(r, 3) = Av[0]; d.h_x(r, 4) = Av[1]; d.h_x(r, 5) = Av[2];
            if (extrinsic_est_en) {
                V3 t = qrot(qconj(s.ril), Cv);
                V3 Bv = cross(pbe, t);
                for (int k = 0; k < 3; k++) { d.h_x(r, 6 + k) = Bv[k]; d.h_x(r, 9 + k) = Cv[k]; }
            }
            d.h[r] = -(double)np.w;
        }
        if (degenerate_detect_en) {
            is_degenerate = false;
            double HTH[9] = {0};
            for (int r = 0; r < effct_feat_num; r++)
                for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) HTH[a * 3 + b] += d.h_x(r, a) * d.h_x(r, b);
            double w[3], V[9];
            eig3(HTH, w, V);
            double V2[9];  // mat_v2 = V^T
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) V2[a * 3 + b] = V[b * 3 + a];
            for (int i = 0; i < 3; i++) {
                float contri = 0, strong = 0;
                for (int r = 0; r < effct_feat_num; r++) {
                    double f0 = d.h_x(r, 0), f1 = d.h_x(r, 1), f2 = d.h_x(r, 2);
                    const double nn = std::sqrt(f0 * f0 + f1 * f1 + f2 * f2);
                    if (nn > 0) { f0 /= nn; f1 /= nn; f2 /= nn; }
                    const float dotp = (float)std::fabs(f0 * V[0 * 3 + i] + f1 * V[1 * 3 + i] + f2 * V[2 * 3 + i]);
                    if (dotp > 0.1736f) contri += dotp;
                    if (dotp > 0.7070f) strong += dotp;
                }
                last_contri[i] = contri; last_strong[i] = strong; last_eigval[i] = w[i];
                if (contri < 250.0f && strong < 50.0f) {
                    for (int j = 0; j < 3; j++) V2[i * 3 + j] = 0;
                    is_degenerate = true;
                }
            }
            if (is_degenerate) {
                // mat_p = (V^T)^-1 * V2 ; h_x[:, :3] = (mat_p * h_x[:, :3]^T)^T
                Mat Vt(3, 3), V2m(3, 3);
                for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { Vt(a, b) = V[b * 3 + a]; V2m(a, b) = V2[a * 3 + b]; }
                Mat Pm = mul(inverse(Vt), V2m);
                for (int r = 0; r < effct_feat_num; r++) {
                    double o[3] = {d.h_x(r, 0), d.h_x(r, 1), d.h_x(r, 2)};
                    for (int a = 0; a < 3; a++) d.h_x(r, a) = Pm(a, 0) * o[0] + Pm(a, 1) * o[1] + Pm(a, 2) * o[2];
                }
            }
        }
    }

    // laserMapping.cpp:984-1023 (wheelspeed_en == false -> wheel block never valid)
    // measurement model supplied from outside (tests/test_ikfom_vs_ref.py drives the oracle, the product's host filter and
    // the reference's real esekf with the same rows): fn(ctx, state26, converge, &n, rows n x 6, h n, cap) -> valid
    typedef int (*meas_fn)(void* ctx, const double* s26, int converge, int* n, double* rows6, double* h, int cap);
    bool wheelspeed_en = false;   // laserMapping.cpp:83
    bool meas_ins_valid = false;  // Measures.ins.back() of the scan being registered, and its lidar_end_time
    double meas_ins_stamp = 0.0, meas_lidar_end = 0.0;
    V3 meas_ins_vel{{0, 0, 0}};
    meas_fn ext_fn = nullptr;
    void* ext_ctx = nullptr;
    int ext_cap = 0;
    void h_share_model(const State& s, DynShare& d) {
        DynShare geo = d;  // copy: stale h_x / h of the previous pass survive an early return
        if (ext_fn) {  // the point-to-plane part supplied from outside (filter pinning tests); the wheel-speed part below applies to it as well
            double s26[26];
            state_to(s, s26);
            std::vector<double> rows((size_t)ext_cap * 6), hv(ext_cap);
            int n = 0;
            const int rc_ext = ext_fn(ext_ctx, s26, d.converge ? 1 : 0, &n, rows.data(), hv.data(), ext_cap);
            if (!rc_ext) { d.valid = false; return; }
            if (rc_ext == 2) {
                // "No Effective Points" (laserMapping.cpp:888-893): h_share_model_geometric sets valid = false on the COPY and returns -- geo keeps
                // the h_x / h the previous pass left in the shared struct (its point-to-plane rows and the wheel-speed rows appended to them)
                effct_feat_num = 0;
            } else {
                geo.h_x = Mat(n, 15);
                geo.h.assign(hv.begin(), hv.begin() + n);
                for (int r = 0; r < n; r++) for (int c = 0; c < 6; c++) geo.h_x(r, c) = rows[(size_t)r * 6 + c];
                effct_feat_num = n;
            }
        }
        // h_share_model_wheelspeed (laserMapping.cpp:794-811): three rows dh/dv = I, h = rot * v_ins - vel, when wheelspeed_en and the last
        // INS sample of this scan is within 10 ms of its end.  (wheelspeed_en is a constant false in the reference: dead code there.)
        bool ws_valid = false;
        double ws_h[3] = {0, 0, 0};
        if (wheelspeed_en && meas_ins_valid && (meas_lidar_end - meas_ins_stamp) < 0.01) {
            const V3 vel = qrot(s.rot, meas_ins_vel);
            for (int a = 0; a < 3; a++) ws_h[a] = vel[a] - s.vel[a];
            ws_valid = true;
        }
        if (!ext_fn) h_share_model_geometric(s, geo);
        const int n_geo = (int)geo.h.size();
        const int n_terms = n_geo + (ws_valid ? 3 : 0);
        if (ws_valid) {  // :994-1012
            const float weight = !is_degenerate ? (float)(0.0001 * n_geo) : (float)(0.001 * n_geo);
            Mat hx(n_terms, 15);
            for (int r = 0; r < n_geo; r++)
                for (int c = 0; c < 15; c++) hx(r, c) = geo.h_x(r, c);
            for (int a = 0; a < 3; a++) hx(n_geo + a, 12 + a) = 1.0;
            d.h_x = hx;
            d.h = geo.h;
            for (int a = 0; a < 3; a++) d.h.push_back(ws_h[a] * weight);
        } else {
            d.h_x = geo.h_x;
            d.h = geo.h;
        }
        if (n_terms == 0) d.valid = false;
    }

    // esekfom.hpp:1619-1931
    void update_iterated(double R) {
        DynShare dyn;
        dyn.valid = true;
        dyn.converge = true;
        int t = 0;
        const State x_prop = x;
        const Mat P_prop = P;
        const int n = 23;
        Mat K_x(n, n);
        std::vector<double> K_h(n, 0.0);
        double dx_new[23] = {0};
        log.clear();
        for (int i = -1; i < max_iter; i++) {
            dyn.valid = true;
            PassLog pl;
            std::memset(&pl, 0, sizeof(pl));
            pl.knn = dyn.converge ? 1 : 0;
            h_share_model(x, dyn);
            pl.valid = dyn.valid;
            pl.n_eff = effct_feat_num;
            pl.degenerate = is_degenerate;
            pl.sum_abs_res = total_residual;
            if (!dyn.valid) { log.push_back(pl); continue; }
            const Mat& hx = dyn.h_x;
            const int dof = hx.r;
            double dx[23];
            state_boxminus(x, x_prop, dx);
            for (int k = 0; k < 23; k++) dx_new[k] = dx[k];
            P = P_prop;
            const int so3_idx[2] = {3, 6};
            for (int si = 0; si < 2; si++) {
                const int idx = so3_idx[si];
                double A[9], J[9];
                A_matrix(V3{{dx[idx], dx[idx + 1], dx[idx + 2]}}, A);
                for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) J[a * 3 + b] = A[b * 3 + a];
                apply_left3(J, dx_new + idx);
                apply_rows3(P, idx, J);
                apply_cols3(P, idx, J);
            }
            {
                const int idx = 21;
                double Nx[6], Mx[6], J[4];
                double seg[2] = {dx[idx], dx[idx + 1]};
                S2_Nx_yy(x.grav, Nx);
                S2_Mx(x_prop.grav, seg, Mx);
                for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { double s = 0; for (int k = 0; k < 3; k++) s += Nx[a * 3 + k] * Mx[k * 2 + b]; J[a * 2 + b] = s; }
                double t0 = J[0] * dx_new[idx] + J[1] * dx_new[idx + 1], t1 = J[2] * dx_new[idx] + J[3] * dx_new[idx + 1];
                dx_new[idx] = t0; dx_new[idx + 1] = t1;
                apply_rows2(P, idx, J);
                apply_cols2(P, idx, J);
            }
            if (n > dof) {  // esekfom.hpp:1715-1744
                Mat hc(dof, n);
                for (int r = 0; r < dof; r++) for (int c = 0; c < 15; c++) hc(r, c) = hx(r, c);
                Mat hct = transpose(hc);
                Mat S = mul(mul(hc, P), hct);
                for (int a = 0; a < dof; a++) for (int b = 0; b < dof; b++) S(a, b) = S(a, b) / R + (a == b ? 1.0 : 0.0);
                Mat K = mul(mul(P, hct), inverse(S));
                for (auto& v : K.a) v /= R;
                for (int a = 0; a < n; a++) { double s = 0; for (int r = 0; r < dof; r++) s += K(a, r) * dyn.h[r]; K_h[a] = s; }
                K_x = mul(K, hc);
            } else {  // esekfom.hpp:1782-1809
                Mat Pt = P;
                for (auto& v : Pt.a) v /= R;
                Mat P_temp = inverse(Pt);
                Mat hxt = transpose(hx);
                Mat HTH = mul(hxt, hx);  // 15x15
                for (int a = 0; a < 15; a++) for (int b = 0; b < 15; b++) P_temp(a, b) += HTH(a, b);
                Mat P_inv = inverse(P_temp);
                Mat Pi15(n, 15);
                for (int a = 0; a < n; a++) for (int b = 0; b < 15; b++) Pi15(a, b) = P_inv(a, b);
                Mat T = mul(Pi15, hxt);  // 23 x N
                for (int a = 0; a < n; a++) { double s = 0; for (int r = 0; r < dof; r++) s += T(a, r) * dyn.h[r]; K_h[a] = s; }
                Mat Kx15 = mul(Pi15, HTH);
                K_x = Mat(n, n);
                for (int a = 0; a < n; a++) for (int b = 0; b < 15; b++) K_x(a, b) = Kx15(a, b);
            }
            // test visibility (both branches): the leading 6 x 6 of h_x^T h_x and the first six of h_x^T h, after the degeneracy projection
            for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { double s = 0; for (int r = 0; r < dof; r++) s += hx(r, a) * hx(r, b); pl.JtJ[a * 6 + b] = s; }
            for (int a = 0; a < 6; a++) { double s = 0; for (int r = 0; r < dof; r++) s += hx(r, a) * dyn.h[r]; pl.Jtr[a] = s; }
            double dx_[23];
            for (int a = 0; a < n; a++) {
                double s = K_h[a];
                for (int b = 0; b < n; b++) s += (K_x(a, b) - (a == b ? 1.0 : 0.0)) * dx_new[b];
                dx_[a] = s;
            }
            for (int a = 0; a < n; a++) pl.dx[a] = dx_[a];
            log.push_back(pl);
            state_boxplus(x, dx_);
            dyn.converge = true;
            for (int a = 0; a < n; a++)
                if (std::fabs(dx_[a]) > limit[a]) { dyn.converge = false; break; }
            if (dyn.converge) t++;
            if (!t && i == max_iter - 2) dyn.converge = true;
            if (t > 1 || i == max_iter - 1) {
                Mat L = P;
                for (int si = 0; si < 2; si++) {
                    const int idx = so3_idx[si];
                    double A[9], J[9];
                    A_matrix(V3{{dx_[idx], dx_[idx + 1], dx_[idx + 2]}}, A);
                    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) J[a * 3 + b] = A[b * 3 + a];
                    // L rows from P rows (esekfom.hpp:1847-1849)
                    for (int c = 0; c < n; c++) {
                        double v[3] = {P(idx, c), P(idx + 1, c), P(idx + 2, c)};
                        for (int a = 0; a < 3; a++) L(idx + a, c) = J[a * 3] * v[0] + J[a * 3 + 1] * v[1] + J[a * 3 + 2] * v[2];
                    }
                    for (int c = 0; c < 15; c++) {
                        double v[3] = {K_x(idx, c), K_x(idx + 1, c), K_x(idx + 2, c)};
                        for (int a = 0; a < 3; a++) K_x(idx + a, c) = J[a * 3] * v[0] + J[a * 3 + 1] * v[1] + J[a * 3 + 2] * v[2];
                    }
                    apply_cols3(L, idx, J);
                    apply_cols3(P, idx, J);
                }
                {
                    const int idx = 21;
                    double Nx[6], Mx[6], J[4];
                    double seg[2] = {dx_[idx], dx_[idx + 1]};
                    S2_Nx_yy(x.grav, Nx);
                    S2_Mx(x_prop.grav, seg, Mx);
                    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { double s = 0; for (int k = 0; k < 3; k++) s += Nx[a * 3 + k] * Mx[k * 2 + b]; J[a * 2 + b] = s; }
                    for (int c = 0; c < n; c++) {
                        double v[2] = {P(idx, c), P(idx + 1, c)};
                        L(idx, c) = J[0] * v[0] + J[1] * v[1];
                        L(idx + 1, c) = J[2] * v[0] + J[3] * v[1];
                    }
                    for (int c = 0; c < 15; c++) {
                        double v[2] = {K_x(idx, c), K_x(idx + 1, c)};
                        K_x(idx, c) = J[0] * v[0] + J[1] * v[1];
                        K_x(idx + 1, c) = J[2] * v[0] + J[3] * v[1];
                    }
                    apply_cols2(L, idx, J);
                    apply_cols2(P, idx, J);
                }
                Mat Pn(n, n);
                for (int a = 0; a < n; a++)
                    for (int b = 0; b < n; b++) {
                        double s = 0;
                        for (int k = 0; k < 15; k++) s += K_x(a, k) * P(k, b);
                        Pn(a, b) = L(a, b) - s;
                    }
                P = Pn;
                return;
            }
        }
    }
    static void apply_left3(const double J[9], double* v) {
        double t[3] = {v[0], v[1], v[2]};
        for (int a = 0; a < 3; a++) v[a] = J[a * 3] * t[0] + J[a * 3 + 1] * t[1] + J[a * 3 + 2] * t[2];
    }
    static void apply_rows3(Mat& M, int idx, const double J[9]) {  // M[idx:idx+3, :] = J M[idx:idx+3, :]
        for (int c = 0; c < M.c; c++) {
            double v[3] = {M(idx, c), M(idx + 1, c), M(idx + 2, c)};
            for (int a = 0; a < 3; a++) M(idx + a, c) = J[a * 3] * v[0] + J[a * 3 + 1] * v[1] + J[a * 3 + 2] * v[2];
        }
    }
    static void apply_cols3(Mat& M, int idx, const double J[9]) {  // M[:, idx:idx+3] = M[:, idx:idx+3] J^T
        for (int r = 0; r < M.r; r++) {
            double v[3] = {M(r, idx), M(r, idx + 1), M(r, idx + 2)};
            for (int a = 0; a < 3; a++) M(r, idx + a) = v[0] * J[a * 3] + v[1] * J[a * 3 + 1] + v[2] * J[a * 3 + 2];
        }
    }
    static void apply_rows2(Mat& M, int idx, const double J[4]) {
        for (int c = 0; c < M.c; c++) {
            double v[2] = {M(idx, c), M(idx + 1, c)};
            M(idx, c) = J[0] * v[0] + J[1] * v[1];
            M(idx + 1, c) = J[2] * v[0] + J[3] * v[1];
        }
    }
    static void apply_cols2(Mat& M, int idx, const double J[4]) {
        for (int r = 0; r < M.r; r++) {
            double v[2] = {M(r, idx), M(r, idx + 1)};
            M(r, idx) = v[0] * J[0] + v[1] * J[1];
            M(r, idx + 1) = v[0] * J[2] + v[1] * J[3];
        }
    }

    // laserMapping.cpp:523-576
    int map_incremental() {
        std::vector<P4> to_add, no_ds;
        to_add.reserve(n_ds);
        no_ds.reserve(n_ds);
        const float fs = leaf_map;
        for (int i = 0; i < n_ds; i++) {
            ds_world[i] = body_to_world(x, ds_body[i]);
            const P4& pw = ds_world[i];
            if (!nearest[i].empty() && flg_EKF_inited) {
                const auto& near = nearest[i];
                bool need_add = true;
                P4 mid;
                // floor(x / 0.5) * 0.5 + 0.5 * 0.5 evaluated in double (filter_size_map_min is double), stored float
                mid.x = (float)(std::floor((double)pw.x / (double)fs) * (double)fs + 0.5 * (double)fs);
                mid.y = (float)(std::floor((double)pw.y / (double)fs) * (double)fs + 0.5 * (double)fs);
                mid.z = (float)(std::floor((double)pw.z / (double)fs) * (double)fs + 0.5 * (double)fs);
                mid.w = 0;
                const float dist = calc_dist(pw, mid);
                const double half = 0.5 * (double)fs;
                if (std::fabs((double)(near[0].x - mid.x)) > half && std::fabs((double)(near[0].y - mid.y)) > half &&
                    std::fabs((double)(near[0].z - mid.z)) > half) {
                    no_ds.push_back(pw);
                    continue;
                }
                for (int k = 0; k < 5; k++) {
                    if (near.size() < 5) break;
                    if (calc_dist(near[k], mid) < dist) { need_add = false; break; }
                }
                if (need_add) to_add.push_back(pw);
            } else {
                to_add.push_back(pw);
            }
        }
        ivox.add_points(to_add.data(), (int)to_add.size(), travel);
        ivox.add_points(no_ds.data(), (int)no_ds.size(), travel);
        return (int)(to_add.size() + no_ds.size());
    }
    static float calc_dist(const P4& a, const P4& b) {  // common_lib.h:231-234
        return ((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y)) + (a.z - b.z) * (a.z - b.z);
    }

    void set_ds(const P4* pts, int n) {
        n_ds = n;
        ds_body.assign(pts, pts + n);
        ds_world.resize(n);
        nearest.resize(n);  // Nearest_Points.resize: surviving entries keep their content (laserMapping.cpp:1274)
    }

    // ------------------------------------------------------------------------------------------
    // IMU front half of fastlio_main: buffers + sync_packages (laserMapping.cpp:397-415,445-520), ImuProcess
    // (IMU_Processing.hpp: ctor :84-104, IMU_init :164-236, UndistortPcl :238-406, Process :408-450) and
    // esekf::predict (esekfom.hpp:279-383) with the process model of use-ikfom.hpp:36-88.
    // IMU accelerations are in units of g inside (fastlio_imu_enqueue divides by 9.81, laserMapping.cpp:410).
    // ------------------------------------------------------------------------------------------
    struct Imu { double stamp; V3 acc, gyr; };
    struct ScanIn { std::vector<P4> pts; std::vector<float> t_ms; double beg; };
    struct Pose6 { double off; V3 acc, gyr, vel, pos; double R[9]; };
    std::deque<Imu> imu_buffer;
    std::deque<ScanIn> lidar_buffer;
    std::deque<std::pair<double, V3>> ins_buffer;  // (stamp, velocity in the IMU frame) -- fastlio_ins_enqueue after its rotations
    double lidar_mean_scantime = 0.1;  // scan_period (laserMapping.cpp:1121)
    // ImuProcess members
    bool b_first_frame = true, imu_need_init = true, state_init_done = false;
    int init_iter_num = 1;
    V3 mean_acc{{0, 0, -1.0}}, mean_gyr{{0, 0, 0}}, cov_acc{{0.1, 0.1, 0.1}}, cov_gyr{{0.1, 0.1, 0.1}};
    V3 cov_acc_scale{{0.1, 0.1, 0.1}}, cov_gyr_scale{{0.1, 0.1, 0.1}}, cov_bias_gyr{{0.0001, 0.0001, 0.0001}}, cov_bias_acc{{0.0001, 0.0001, 0.0001}};
    V3 vel_last{{0, 0, 0}}, angvel_last{{0, 0, 0}}, acc_s_last{{0, 0, 0}};
    Imu last_imu{0, {{0, 0, 0}}, {{0, 0, 0}}};
    double last_lidar_end_time = 0;
    double Qd[12] = {1e-4, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4, 1e-5, 1e-5, 1e-5, 1e-5, 1e-5, 1e-5};  // process_noise_cov(): ng, na, nbg, nba
    bool undistort_en = true;
    double blind = 0.1;
    V3 Lidar_T{{0, 0, 0}};
    Quat Lidar_R{0, 0, 0, 1};
    int point_filter_num = 1, max_point_num = -1;
    std::vector<P4> feats_undistort;
    State odom_start, odom_end;  // states before / after the scan (fastlio_odometry)

    void imu_enqueue(double stamp, const double gyr[3], const double acc_ms2[3]) {
        Imu m;
        m.stamp = stamp;
        for (int i = 0; i < 3; i++) { m.gyr[i] = gyr[i]; m.acc[i] = acc_ms2[i] / 9.81; }
        imu_buffer.push_back(m);
    }
    // Preprocess::velodyne_handler (preprocess.cpp:395-451, feature extraction off, point_filter_num 1): curvature =
    // per-point time in ms (float), points inside the blind radius dropped
    void pcl_enqueue(const P4* pts, const uint32_t* t_us, int n, double stamp) {
        ScanIn sc;
        sc.beg = stamp;
        if (max_point_num > 0) point_filter_num = std::max(1, n / max_point_num);  // preprocess.cpp:396-398: overwrites the member
        for (int i = 0; i < n; i++) {
            const P4& p = pts[i];
            if (i % point_filter_num == 0 && (double)(p.x * p.x + p.y * p.y + p.z * p.z) > blind * blind) {
                sc.pts.push_back(p);
                sc.t_ms.push_back(t_us[i] / 1000.0f);  // uint32 -> float, then the f32 division (preprocess.cpp:406)
            }
        }
        lidar_buffer.push_back(std::move(sc));
    }

    // esekf::predict.  State dims (24): pos0 rot3 ril6 til9 vel12 bg15 ba18 grav21(3); dofs (23): ... grav21(2)
    void predict(double dt, const V3& acc, const V3& gyro) {
        const int n = 23;
        // f (get_f)
        double f[24] = {0};
        V3 amb{{acc[0] - x.ba[0], acc[1] - x.ba[1], acc[2] - x.ba[2]}};
        V3 a_in = qrot(x.rot, amb);
        for (int i = 0; i < 3; i++) { f[i] = x.vel[i]; f[3 + i] = gyro[i] - x.bg[i]; f[12 + i] = a_in[i] + x.grav[i]; }
        // f_x (24 x 23), f_w (24 x 12)
        Mat fx(24, 23), fw(24, 12);
        double R[9], H[9], RH[9];
        qtoR(x.rot, R);
        hat(amb, H);
        mm3(R, H, RH);
        for (int i = 0; i < 3; i++) {
            fx(i, 12 + i) = 1.0;
            fx(3 + i, 15 + i) = -1.0;
            for (int j = 0; j < 3; j++) { fx(12 + i, 3 + j) = -RH[i * 3 + j]; fx(12 + i, 18 + j) = -R[i * 3 + j]; fw(12 + i, 3 + j) = -R[i * 3 + j]; }
            fw(3 + i, i) = -1.0;
            fw(15 + i, 6 + i) = 1.0;
            fw(18 + i, 9 + i) = 1.0;
        }
        {
            double zero2[2] = {0, 0}, Mx[6];
            S2_Mx(x.grav, zero2, Mx);
            for (int i = 0; i < 3; i++) for (int j = 0; j < 2; j++) fx(12 + i, 21 + j) = Mx[i * 2 + j];
        }
        const State x_before = x;
        // x_.oplus(f_, dt): vect += dt * f; SO3 *= exp(f, dt) (SOn.hpp:242-245, exp(vec, scale) -> mtk exp(vec, scale / 2));
        // S2::oplus with f == 0 is the identity rotation; til / bg / ba / ril have f == 0
        for (int i = 0; i < 3; i++) { x.pos[i] += dt * f[i]; x.til[i] += dt * f[9 + i]; x.vel[i] += dt * f[12 + i]; x.bg[i] += dt * f[15 + i]; x.ba[i] += dt * f[18 + i]; }
        x.rot = qmul(x.rot, mtk_exp(V3{{f[3], f[4], f[5]}}, dt / 2));
        x.ril = qmul(x.ril, mtk_exp(V3{{f[6], f[7], f[8]}}, dt / 2));
        Mat F1 = Mat::eye(n), fxf(n, n), fwf(n, 12);
        // vect states: idx == dim for the first 21 dims except the SO3 ones handled below
        const int vect_idx[5] = {0, 9, 12, 15, 18};
        for (int v = 0; v < 5; v++)
            for (int j = 0; j < 3; j++) {
                for (int i = 0; i < n; i++) fxf(vect_idx[v] + j, i) = fx(vect_idx[v] + j, i);
                for (int i = 0; i < 12; i++) fwf(vect_idx[v] + j, i) = fw(vect_idx[v] + j, i);
            }
        const int so3_idx[2] = {3, 6};
        for (int si = 0; si < 2; si++) {
            const int idx = so3_idx[si];
            V3 seg{{-f[idx] * dt, -f[idx + 1] * dt, -f[idx + 2] * dt}};
            // F_x1 block = exp(seg, scalar(1/2) == 0).toRotationMatrix() == identity (esekfom.hpp:312)
            double A[9];
            A_matrix(seg, A);
            for (int i = 0; i < n; i++) {
                double v[3] = {fx(idx, i), fx(idx + 1, i), fx(idx + 2, i)};
                for (int a = 0; a < 3; a++) fxf(idx + a, i) = A[a * 3] * v[0] + A[a * 3 + 1] * v[1] + A[a * 3 + 2] * v[2];
            }
            for (int i = 0; i < 12; i++) {
                double v[3] = {fw(idx, i), fw(idx + 1, i), fw(idx + 2, i)};
                for (int a = 0; a < 3; a++) fwf(idx + a, i) = A[a * 3] * v[0] + A[a * 3 + 1] * v[1] + A[a * 3 + 2] * v[2];
            }
        }
        {  // S2 at idx 21, dim 21
            V3 seg{{f[21] * dt, f[22] * dt, f[23] * dt}};
            double Nx[6], Mx[6], zero2[2] = {0, 0};
            S2_Nx_yy(x.grav, Nx);
            S2_Mx(x_before.grav, zero2, Mx);
            // res = exp(seg, 0) == identity
            for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { double s2 = 0; for (int k = 0; k < 3; k++) s2 += Nx[a * 3 + k] * Mx[k * 2 + b]; F1(21 + a, 21 + b) = s2; }
            double Hb[9], A[9], T[9], res23[6];
            hat(x_before.grav, Hb);
            A_matrix(seg, A);
            // res_temp_S2 = -Nx * I * x_before_hat * A^T
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s2 = 0; for (int k = 0; k < 3; k++) s2 += Hb[i * 3 + k] * A[j * 3 + k]; T[i * 3 + j] = s2; }
            for (int a = 0; a < 2; a++) for (int j = 0; j < 3; j++) { double s2 = 0; for (int k = 0; k < 3; k++) s2 += Nx[a * 3 + k] * T[k * 3 + j]; res23[a * 3 + j] = -s2; }
            for (int i = 0; i < n; i++) {
                double v[3] = {fx(21, i), fx(22, i), fx(23, i)};
                for (int a = 0; a < 2; a++) fxf(21 + a, i) = res23[a * 3] * v[0] + res23[a * 3 + 1] * v[1] + res23[a * 3 + 2] * v[2];
            }
            for (int i = 0; i < 12; i++) {
                double v[3] = {fw(21, i), fw(22, i), fw(23, i)};
                for (int a = 0; a < 2; a++) fwf(21 + a, i) = res23[a * 3] * v[0] + res23[a * 3 + 1] * v[1] + res23[a * 3 + 2] * v[2];
            }
        }
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) F1(i, j) += fxf(i, j) * dt;
        Mat FP = mul(F1, P);
        Mat Pn = mul(FP, transpose(F1));
        Mat W(n, 12);
        for (int i = 0; i < n; i++) for (int j = 0; j < 12; j++) W(i, j) = dt * fwf(i, j);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                double s2 = 0;
                for (int k = 0; k < 12; k++) s2 += W(i, k) * Qd[k] * W(j, k);
                Pn(i, j) += s2;
            }
        P = Pn;
    }

    void set_Q() {
        for (int i = 0; i < 3; i++) { Qd[i] = cov_gyr[i]; Qd[3 + i] = cov_acc[i]; Qd[6 + i] = cov_bias_gyr[i]; Qd[9 + i] = cov_bias_acc[i]; }
    }

    void imu_init(const std::vector<Imu>& imu, double lidar_beg, double lidar_end, const V3* ins_vel = nullptr) {
        int& N = init_iter_num;
        if (b_first_frame) {
            // Reset()
            mean_acc = {{0, 0, -1.0}}; mean_gyr = {{0, 0, 0}}; vel_last = {{0, 0, 0}}; angvel_last = {{0, 0, 0}};
            imu_need_init = true; state_init_done = false; init_iter_num = 1; last_imu = Imu{0, {{0, 0, 0}}, {{0, 0, 0}}};
            N = 1;
            b_first_frame = false;
            mean_acc = imu.front().acc;
            mean_gyr = imu.front().gyr;
        }
        for (const Imu& m : imu) {
            for (int i = 0; i < 3; i++) {
                mean_acc[i] += (m.acc[i] - mean_acc[i]) / N;
                mean_gyr[i] += (m.gyr[i] - mean_gyr[i]) / N;
                cov_acc[i] = cov_acc[i] * (N - 1.0) / N + (m.acc[i] - mean_acc[i]) * (m.acc[i] - mean_acc[i]) * (N - 1.0) / (N * N);
                cov_gyr[i] = cov_gyr[i] * (N - 1.0) / N + (m.gyr[i] - mean_gyr[i]) * (m.gyr[i] - mean_gyr[i]) * (N - 1.0) / (N * N);
            }
            N++;
        }
        if (ins_vel) vel_last = *ins_vel;  // IMU_Processing.hpp:201-204
        const double na = std::sqrt(mean_acc[0] * mean_acc[0] + (mean_acc[1] * mean_acc[1] + mean_acc[2] * mean_acc[2]));
        const double ng = std::sqrt(mean_gyr[0] * mean_gyr[0] + (mean_gyr[1] * mean_gyr[1] + mean_gyr[2] * mean_gyr[2]));
        if (std::fabs(na - 1.0) > 0.1 || ng > (10.0 / 180.0 * M_PI)) { b_first_frame = true; return; }
        {   // init_state.grav = S2(-mean_acc / |mean_acc| * G_m_s2): the S2 ctor re-normalises to length 9.809 (S2.hpp:124-127)
            V3 g{{-mean_acc[0] / na * 9.81, -mean_acc[1] / na * 9.81, -mean_acc[2] / na * 9.81}};
            const double z = g[0] * g[0] + (g[1] * g[1] + g[2] * g[2]);
            if (z > 0) { const double nz = std::sqrt(z); for (int i = 0; i < 3; i++) g[i] /= nz; }
            for (int i = 0; i < 3; i++) x.grav[i] = g[i] * kS2Len;
        }
        for (int i = 0; i < 3; i++) { x.vel[i] = vel_last[i]; x.bg[i] = 0; x.ba[i] = 0; }
        x.til = Lidar_T; x.ril = Lidar_R;  // init_state.offset_T_L_I / offset_R_L_I = Lidar_*_wrt_IMU
        P = Mat::eye(23);
        for (int i : {6, 7, 8, 9, 10, 11}) P(i, i) = 0.00001;
        for (int i : {15, 16, 17}) P(i, i) = 0.0001;
        for (int i : {18, 19, 20}) P(i, i) = 0.001;
        P(21, 21) = P(22, 22) = 0.00001;
        last_imu = imu.back();
        last_lidar_end_time = lidar_end;
        odom_start = x;  // start_state_point = kf_state.get_x() (IMU_Processing.hpp:234)
        (void)lidar_beg;
    }

    void pose6(std::vector<Pose6>& out, double off) {
        Pose6 p;
        p.off = off; p.acc = acc_s_last; p.gyr = angvel_last; p.vel = x.vel; p.pos = x.pos;
        qtoR(x.rot, p.R);
        out.push_back(p);
    }
    void after_predict(const V3& angvel_avr, const V3& acc_avr) {
        for (int i = 0; i < 3; i++) angvel_last[i] = angvel_avr[i] - x.bg[i];
        V3 amb{{acc_avr[0] - x.ba[0], acc_avr[1] - x.ba[1], acc_avr[2] - x.ba[2]}};
        acc_s_last = qrot(x.rot, amb);
        for (int i = 0; i < 3; i++) acc_s_last[i] += x.grav[i];
    }

    void undistort_pcl(const ScanIn& sc, const std::vector<Imu>& meas_imu, double lidar_end, std::vector<P4>& out) {
        const double na = std::sqrt(mean_acc[0] * mean_acc[0] + (mean_acc[1] * mean_acc[1] + mean_acc[2] * mean_acc[2]));
        set_Q();
        if (sc.beg > last_lidar_end_time) {
            V3 acc_avr{{last_imu.acc[0] * 9.81 / na, last_imu.acc[1] * 9.81 / na, last_imu.acc[2] * 9.81 / na}};
            predict(sc.beg - last_lidar_end_time, acc_avr, last_imu.gyr);
            after_predict(last_imu.gyr, acc_avr);
            last_lidar_end_time = sc.beg;
        }
        odom_start = x;
        std::vector<Imu> v_imu;
        v_imu.push_back(last_imu);
        for (const Imu& m : meas_imu) v_imu.push_back(m);
        const double imu_end_time = v_imu.back().stamp, pcl_beg = sc.beg, pcl_end = lidar_end;
        std::vector<Pose6> poses;
        pose6(poses, 0.0);
        for (size_t k = 0; k + 1 < v_imu.size(); k++) {
            const Imu &head = v_imu[k], &tail = v_imu[k + 1];
            if (tail.stamp < last_lidar_end_time) continue;
            V3 angvel_avr, acc_avr;
            for (int i = 0; i < 3; i++) { angvel_avr[i] = 0.5 * (head.gyr[i] + tail.gyr[i]); acc_avr[i] = 0.5 * (head.acc[i] + tail.acc[i]) * 9.81 / na; }
            double dt = head.stamp < last_lidar_end_time ? tail.stamp - last_lidar_end_time : tail.stamp - head.stamp;
            dt = std::min(1.0, dt);
            predict(dt, acc_avr, angvel_avr);
            after_predict(angvel_avr, acc_avr);
            pose6(poses, tail.stamp - pcl_beg);
        }
        {
            V3 angvel_avr = v_imu.back().gyr, acc_avr;
            for (int i = 0; i < 3; i++) acc_avr[i] = v_imu.back().acc[i] * 9.81 / na;
            const double note = pcl_end > imu_end_time ? 1.0 : -1.0;
            double dt = std::min(1.0, note * (pcl_end - imu_end_time));
            predict(dt, acc_avr, angvel_avr);
            after_predict(angvel_avr, acc_avr);
            pose6(poses, pcl_end - pcl_beg);
        }
        last_imu = meas_imu.back();
        last_lidar_end_time = pcl_end;
        // backward propagation.  The reference sorts the cloud by time (std::sort, unstable) and walks it from the back;
        // every point is compensated with the IMU segment [head, tail) that contains its time -- restated per point, in
        // input order (the downstream VoxelGrid does not depend on the order of distinct voxels).  One quirk of the walk is
        // kept: when the earliest point has t > 0 the iterator parks on it (`if (it_pcl == begin) break`) and every earlier
        // segment compensates it AGAIN, on the already compensated coordinates (IMU_Processing.hpp:371-404).  With several
        // points sharing the minimum time std::sort decides which one sits at begin(); the lowest input index is used here.
        out = sc.pts;
        if (!undistort_en || out.empty()) return;
        size_t first = 0;
        for (size_t i = 1; i < out.size(); i++)
            if (sc.t_ms[i] < sc.t_ms[first]) first = i;
        auto compensate = [&](P4& p, double t, int h) {
            const Pose6 &head = poses[h], &tail = poses[h + 1];
            float q[3] = {p.x, p.y, p.z};
            undistort_point(head.R, head.vel, head.pos, tail.acc, tail.gyr, t - head.off, q, x.pos, x.rot, x.ril, x.til);
            p.x = q[0]; p.y = q[1]; p.z = q[2];
        };
        for (size_t i = 0; i < out.size(); i++) {
            const double t = (double)sc.t_ms[i] / 1000.0;
            int h = -1;  // the last head with head.off < t (points with t <= poses[0].off == 0 stay untouched)
            for (int k = (int)poses.size() - 2; k >= 0; k--)
                if (t > poses[k].off) { h = k; break; }
            if (h < 0) continue;
            compensate(out[i], t, h);
            if (i == first)
                for (int k = h - 1; k >= 0; k--)
                    if (t > poses[k].off) compensate(out[i], t, k);
        }
    }

    // fastlio_main (laserMapping.cpp:1126-1310): returns 5 nothing to do, 0 first-scan latch, 4 IMU initialising,
    // 1 map seeded, 2 too few points, 3 state updated
    int frontend_main() {
        if (lidar_buffer.empty() || imu_buffer.empty()) return 5;
        ScanIn sc = std::move(lidar_buffer.front());
        lidar_buffer.pop_front();
        const double lidar_end = sc.beg + lidar_mean_scantime;
        std::vector<Imu> meas_imu;
        while (!imu_buffer.empty() && !(imu_buffer.front().stamp > lidar_end)) { meas_imu.push_back(imu_buffer.front()); imu_buffer.pop_front(); }
        bool have_ins = false;
        V3 ins_vel{{0, 0, 0}};
        while (!ins_buffer.empty() && !(ins_buffer.front().first > lidar_end)) {
            ins_vel = ins_buffer.front().second;
            meas_ins_stamp = ins_buffer.front().first;
            have_ins = true;
            ins_buffer.pop_front();
        }
        meas_ins_valid = have_ins;  // meas.ins is rebuilt for every scan (laserMapping.cpp:495-503)
        meas_ins_vel = ins_vel;
        meas_lidar_end = lidar_end;
        if (flg_first_scan) { first_lidar_time = sc.beg; flg_first_scan = false; return 0; }
        // Process() returns at once when no IMU sample fell into the scan: feats_undistort keeps the PREVIOUS scan's cloud
        // and fastlio_main registers that again (IMU_Processing.hpp:413, laserMapping.cpp:1189-1197)
        if (meas_imu.empty()) {
            if (feats_undistort.empty()) return 4;
            const int rc0 = process_scan_core(feats_undistort.data(), (int)feats_undistort.size(), sc.beg);
            odom_end = x;
            return rc0;
        }
        if (imu_need_init) {
            imu_init(meas_imu, sc.beg, lidar_end, have_ins ? &ins_vel : nullptr);
            imu_need_init = true;
            last_imu = meas_imu.back();
            if (init_iter_num > 100) {  // MAX_INI_COUNT (IMU_Processing.hpp:25)
                imu_need_init = false;
                cov_acc = cov_acc_scale;
                cov_gyr = cov_gyr_scale;
            }
            return 4;
        }
        state_init_done = true;  // IMU_Processing.hpp:443
        undistort_pcl(sc, meas_imu, lidar_end, feats_undistort);
        const int rc = process_scan_core(feats_undistort.data(), (int)feats_undistort.size(), sc.beg);
        odom_end = x;
        return rc;
    }

    // fastlio_main after p_imu->Process (laserMapping.cpp:1189-1304).  `raw` plays
    // feats_undistort; the caller has already put the propagated state/covariance in x/P.
    // returns: 0 first-scan latch, 1 seeded map, 2 too few points, 3 updated, <0 error
    int process_scan(const P4* raw, int n_raw, double lidar_beg_time) {
        if (flg_first_scan) {
            first_lidar_time = lidar_beg_time;
            flg_first_scan = false;
            return 0;
        }
        return process_scan_core(raw, n_raw, lidar_beg_time);
    }
    int process_scan_core(const P4* raw, int n_raw, double lidar_beg_time) {
        if (n_raw <= 0) return 2;
        flg_EKF_inited = (lidar_beg_time - first_lidar_time) < init_time ? false : true;
        std::vector<P4> ds;
        voxel_downsample(raw, n_raw, leaf_surf, ds);
        if ((int)ds.size() > kMaxPts) return -1;
        n_ds = (int)ds.size();
        ds_body = ds;
        if (ivox.grids.empty()) {
            if (n_ds > 5) {
                ds_world.resize(n_ds);
                for (int i = 0; i < n_ds; i++) ds_world[i] = body_to_world(x, ds_body[i]);
                ivox.add_points(ds_world.data(), n_ds, travel);
            }
            return 1;
        }
        if (ivox.stencil_id != 19 && (lidar_beg_time - first_lidar_time) > 10 * init_time) ivox.set_stencil(19);
        if (n_ds < 5) return 2;
        ds_world.resize(n_ds);
        nearest.resize(n_ds);
        update_iterated(laser_cov);
        V3 off = qrot(x.rot, x.til);
        V3 pos_lid{{x.pos[0] + off[0], x.pos[1] + off[1], x.pos[2] + off[2]}};
        const double dxp = pos_lid[0] - last_pos_lid[0], dyp = pos_lid[1] - last_pos_lid[1], dzp = pos_lid[2] - last_pos_lid[2];
        travel = travel + std::sqrt(dxp * dxp + dyp * dyp + dzp * dzp);
        last_pos_lid = pos_lid;
        map_incremental();
        return 3;
    }
};

}  // namespace

// =============================================================================
// C ABI (ctypes)
// =============================================================================
extern "C" {

int orc_voxel_downsample(const float* in_xyzi, int n, float leaf, float* out_xyzi, int cap) {
    std::vector<P4> out;
    int m = voxel_downsample(reinterpret_cast<const P4*>(in_xyzi), n, leaf, out);
    if (m > cap) return -m;
    std::memcpy(out_xyzi, out.data(), sizeof(P4) * (size_t)m);
    return m;
}

int orc_esti_plane(const float* five_xyzi, float threshold, float* pabcd) {
    return esti_plane(pabcd, reinterpret_cast<const P4*>(five_xyzi), threshold) ? 1 : 0;
}

void* orc_ivox_create(float res, int stencil, uint64_t capacity, double max_distance) { return new IVox(res, stencil, (size_t)capacity, max_distance); }
void orc_ivox_destroy(void* h) { delete static_cast<IVox*>(h); }
void orc_ivox_set_stencil(void* h, int stencil) { static_cast<IVox*>(h)->set_stencil(stencil); }
void orc_ivox_add(void* h, const float* pts, int n, double travel) { static_cast<IVox*>(h)->add_points(reinterpret_cast<const P4*>(pts), n, travel); }
void orc_ivox_set_tie_mode(void* h, int mode) { static_cast<IVox*>(h)->tie_mode = mode; }
// GetClosestPoint(pt, out, 5, 5.0) with the list as the reference returns it; out_pts n x 5 x 4
void orc_ivox_knn_as_reference(void* h, const float* q_xyzi, int n, float* out_pts, int* out_cnt) {
    IVox* iv = static_cast<IVox*>(h);
    const P4* q = reinterpret_cast<const P4*>(q_xyzi);
    std::vector<P4> near;
    for (int i = 0; i < n; i++) {
        near.clear();
        iv->closest_as_reference(q[i], near, 5, 5.0);
        out_cnt[i] = (int)near.size();
        for (int k = 0; k < 5; k++) {
            P4 o = k < (int)near.size() ? near[k] : P4{0.f, 0.f, 0.f, 0.f};
            memcpy(out_pts + ((size_t)i * 5 + k) * 4, &o, 16);
        }
    }
}
uint64_t orc_ivox_num_voxels(void* h) { return static_cast<IVox*>(h)->grids.size(); }
uint64_t orc_ivox_num_points(void* h) { return static_cast<IVox*>(h)->num_points(); }
int64_t orc_ivox_dump(void* h, float* out, uint64_t cap) {  // all stored points, list order (most recently touched voxel first)
    IVox* v = static_cast<IVox*>(h);
    const uint64_t n = v->num_points();
    if (n > cap) return -(int64_t)n;
    uint64_t k = 0;
    for (auto& kv : v->cache)
        for (auto& p : kv.second.pts) { std::memcpy(out + k * 4, &p, sizeof(P4)); k++; }
    return (int64_t)n;
}
// pure kNN (no stale-content semantics): out_pts is n x 5 x 4 floats, out_cnt n ints.
// returns the total number of in-range candidates visited (for the C-bar statistic of SURVEY 8d)
uint64_t orc_ivox_knn(void* h, const float* q_xyzi, int n, float* out_pts, int* out_cnt, int threads) {
    IVox* iv = static_cast<IVox*>(h);
    const P4* q = reinterpret_cast<const P4*>(q_xyzi);
    uint64_t visited = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 0 ? threads : 1) reduction(+ : visited)
#endif
    {
        std::vector<Cand> scratch;
        std::vector<P4> near;
#ifdef _OPENMP
#pragma omp for
#endif
        for (int i = 0; i < n; i++) {
            near.clear();
            bool ok = iv->closest(q[i], near, 5, 5.0, scratch);
            visited += scratch.size();
            out_cnt[i] = ok ? (int)near.size() : 0;
            for (int k = 0; k < 5; k++) {
                P4 p = (k < (int)near.size()) ? near[k] : P4{0, 0, 0, 0};
                std::memcpy(out_pts + ((size_t)i * 5 + k) * 4, &p, sizeof(P4));
            }
        }
    }
    return visited;
}
// number of map points resident in the stencil voxels of each query (no range test): C-bar of SURVEY 8d
uint64_t orc_ivox_stencil_points(void* h, const float* q_xyzi, int n) {
    IVox* iv = static_cast<IVox*>(h);
    const P4* q = reinterpret_cast<const P4*>(q_xyzi);
    uint64_t tot = 0;
    for (int i = 0; i < n; i++) {
        Key3 key = iv->pos2grid(q[i]);
        for (const Key3& d : iv->nearby) {
            auto it = iv->grids.find(Key3{key.x + d.x, key.y + d.y, key.z + d.z});
            if (it != iv->grids.end()) tot += it->second->second.pts.size();
        }
    }
    return tot;
}

void* orc_lio_create(float res, int stencil, uint64_t capacity, double max_distance, int threads) {
    Lio* l = new Lio(res, stencil, (size_t)capacity, max_distance);
    l->threads = threads > 0 ? threads : 1;
    return l;
}
void orc_lio_destroy(void* h) { delete static_cast<Lio*>(h); }
void orc_lio_set_state(void* h, const double* s26) { state_from(s26, static_cast<Lio*>(h)->x); }
void orc_lio_get_state(void* h, double* s26) { state_to(static_cast<Lio*>(h)->x, s26); }
void orc_lio_set_cov(void* h, const double* P529) { Lio* l = static_cast<Lio*>(h); for (int i = 0; i < 529; i++) l->P.a[i] = P529[i]; }
void orc_lio_get_cov(void* h, double* P529) { Lio* l = static_cast<Lio*>(h); for (int i = 0; i < 529; i++) P529[i] = l->P.a[i]; }
void orc_lio_set_flags(void* h, int ekf_inited, int first_scan, double travel, double first_lidar_time) {
    Lio* l = static_cast<Lio*>(h);
    l->flg_EKF_inited = ekf_inited != 0;
    l->flg_first_scan = first_scan != 0;
    l->travel = travel;
    l->first_lidar_time = first_lidar_time;
}
void orc_lio_set_tie_mode(void* h, int mode) { static_cast<Lio*>(h)->ivox.tie_mode = mode; }
void orc_lio_set_stencil(void* h, int s) { static_cast<Lio*>(h)->ivox.set_stencil(s); }
void orc_lio_map_add(void* h, const float* pts, int n, double travel) { static_cast<Lio*>(h)->ivox.add_points(reinterpret_cast<const P4*>(pts), n, travel); }
uint64_t orc_lio_map_num_points(void* h) { return static_cast<Lio*>(h)->ivox.num_points(); }
uint64_t orc_lio_map_num_voxels(void* h) { return static_cast<Lio*>(h)->ivox.grids.size(); }
// dump all map points (unordered); returns count (or -needed if cap too small)
int64_t orc_lio_map_dump(void* h, float* out, uint64_t cap) {
    Lio* l = static_cast<Lio*>(h);
    uint64_t n = l->ivox.num_points();
    if (n > cap) return -(int64_t)n;
    uint64_t k = 0;
    for (auto& kv : l->ivox.cache)
        for (auto& p : kv.second.pts) { std::memcpy(out + k * 4, &p, sizeof(P4)); k++; }
    return (int64_t)n;
}
void orc_lio_set_ds(void* h, const float* ds_xyzi, int n) { static_cast<Lio*>(h)->set_ds(reinterpret_cast<const P4*>(ds_xyzi), n); }
// fastlio_init's reset of Nearest_Points / point_selected_surf (laserMapping.cpp:1045-1047)
void orc_lio_reset_cache(void* h) {
    Lio* l = static_cast<Lio*>(h);
    l->nearest.clear();
    l->selected.assign(kMaxPts, 1);
}
int orc_lio_get_ds(void* h, float* out, int cap) {
    Lio* l = static_cast<Lio*>(h);
    if (l->n_ds > cap) return -l->n_ds;
    std::memcpy(out, l->ds_body.data(), sizeof(P4) * (size_t)l->n_ds);
    return l->n_ds;
}
int orc_lio_get_ds_world(void* h, float* out, int cap) {  // feats_down_world as the last linearisation left it
    Lio* l = static_cast<Lio*>(h);
    if (l->n_ds > cap || (int)l->ds_world.size() < l->n_ds) return -l->n_ds;
    std::memcpy(out, l->ds_world.data(), sizeof(P4) * (size_t)l->n_ds);
    return l->n_ds;
}
// one call of h_share_model_geometric at the current state; converge != 0 -> redo kNN.
// outputs per ds point: selected[n], normvec[n*4] (n, pd2), nn_cnt[n], nn_pts[n*5*4]; JtJ36/Jtr6 before the
// degeneracy projection; returns n_eff
int orc_lio_linearize(void* h, int converge, uint8_t* selected, float* normvec, int* nn_cnt, float* nn_pts,
                      double* JtJ36, double* Jtr6, double* sum_abs_res, int* degenerate) {
    Lio* l = static_cast<Lio*>(h);
    DynShare d;
    d.converge = converge != 0;
    const bool deg_en = l->degenerate_detect_en;
    l->degenerate_detect_en = false;
    l->h_share_model_geometric(l->x, d);
    l->degenerate_detect_en = deg_en;
    const int n = l->n_ds;
    for (int i = 0; i < n; i++) {
        if (selected) selected[i] = l->selected[i];
        if (normvec) std::memcpy(normvec + (size_t)i * 4, &l->normvec[i], sizeof(P4));
        if (nn_cnt) nn_cnt[i] = (int)l->nearest[i].size();
        if (nn_pts)
            for (int k = 0; k < 5; k++) {
                P4 p = k < (int)l->nearest[i].size() ? l->nearest[i][k] : P4{0, 0, 0, 0};
                std::memcpy(nn_pts + ((size_t)i * 5 + k) * 4, &p, sizeof(P4));
            }
    }
    if (JtJ36) for (int i = 0; i < 36; i++) JtJ36[i] = 0;
    if (Jtr6) for (int i = 0; i < 6; i++) Jtr6[i] = 0;
    if (l->effct_feat_num >= 1) {
        for (int r = 0; r < d.h_x.r; r++) {
            for (int a = 0; a < 6; a++) {
                if (Jtr6) Jtr6[a] += d.h_x(r, a) * d.h[r];
                if (JtJ36) for (int b = 0; b < 6; b++) JtJ36[a * 6 + b] += d.h_x(r, a) * d.h_x(r, b);
            }
        }
    }
    if (sum_abs_res) *sum_abs_res = l->total_residual;
    if (degenerate) {
        // run the degeneracy logic on a scratch copy to report the decision
        *degenerate = 0;
        if (deg_en && l->effct_feat_num >= 1) {
            DynShare d2;
            d2.converge = false;
            l->h_share_model_geometric(l->x, d2);
            *degenerate = l->is_degenerate ? 1 : 0;
        }
    }
    return l->effct_feat_num;
}
// run esekf::update_iterated_dyn_share_modified on the ds points set by orc_lio_set_ds; returns #passes logged
int orc_lio_update(void* h) {
    Lio* l = static_cast<Lio*>(h);
    l->ds_world.resize(l->n_ds);
    l->update_iterated(l->laser_cov);
    return (int)l->log.size();
}
int orc_lio_pass_log(void* h, int i, int* knn, int* n_eff, int* valid, int* degenerate, double* sum_abs_res, double* JtJ36,
                     double* Jtr6, double* dx23) {
    Lio* l = static_cast<Lio*>(h);
    if (i < 0 || i >= (int)l->log.size()) return -1;
    const PassLog& p = l->log[i];
    *knn = p.knn; *n_eff = p.n_eff; *valid = p.valid; *degenerate = p.degenerate; *sum_abs_res = p.sum_abs_res;
    std::memcpy(JtJ36, p.JtJ, sizeof(p.JtJ));
    std::memcpy(Jtr6, p.Jtr, sizeof(p.Jtr));
    std::memcpy(dx23, p.dx, sizeof(p.dx));
    return 0;
}
int orc_lio_map_incremental(void* h) { return static_cast<Lio*>(h)->map_incremental(); }
int orc_lio_process_scan(void* h, const float* raw_xyzi, int n_raw, double lidar_beg_time) {
    return static_cast<Lio*>(h)->process_scan(reinterpret_cast<const P4*>(raw_xyzi), n_raw, lidar_beg_time);
}
double orc_lio_travel(void* h) { return static_cast<Lio*>(h)->travel; }
int orc_lio_is_degenerate(void* h) { return static_cast<Lio*>(h)->is_degenerate ? 1 : 0; }
void orc_lio_last_degeneracy(void* h, float* contri3, float* strong3, double* eigval3) {
    Lio* l = static_cast<Lio*>(h);
    for (int i = 0; i < 3; i++) { contri3[i] = l->last_contri[i]; strong3[i] = l->last_strong[i]; eigval3[i] = l->last_eigval[i]; }
}

// IMU front half
void orc_lio_imu_enqueue(void* h, double stamp, const double* gyr, const double* acc_ms2) { static_cast<Lio*>(h)->imu_enqueue(stamp, gyr, acc_ms2); }
void orc_lio_pcl_enqueue(void* h, const float* xyzi, const uint32_t* t_us, int n, double stamp) {
    static_cast<Lio*>(h)->pcl_enqueue(reinterpret_cast<const P4*>(xyzi), t_us, n, stamp);
}
void orc_lio_frontend_config(void* h, const double* extT, const double* extR_xyzw, int filter_num, double scan_period, int undistort) {
    Lio* l = static_cast<Lio*>(h);
    l->Lidar_T = V3{{extT[0], extT[1], extT[2]}};
    l->Lidar_R = Quat{extR_xyzw[0], extR_xyzw[1], extR_xyzw[2], extR_xyzw[3]};
    l->point_filter_num = filter_num > 0 ? filter_num : 1;
    l->lidar_mean_scantime = scan_period;
    l->undistort_en = undistort != 0;
}
void orc_lio_set_max_point_num(void* h, int n) { static_cast<Lio*>(h)->max_point_num = n; }  // p_pre->max_point_num (laserMapping.cpp:1100)
void orc_lio_set_wheelspeed(void* h, int on) { static_cast<Lio*>(h)->wheelspeed_en = on != 0; }
void orc_lio_ins_enqueue(void* h, double stamp, const double* v) { static_cast<Lio*>(h)->ins_buffer.push_back({stamp, V3{{v[0], v[1], v[2]}}}); }
int orc_lio_frontend_main(void* h) { return static_cast<Lio*>(h)->frontend_main(); }
void orc_lio_predict(void* h, double dt, const double* acc, const double* gyro) {
    static_cast<Lio*>(h)->predict(dt, V3{{acc[0], acc[1], acc[2]}}, V3{{gyro[0], gyro[1], gyro[2]}});
}
int orc_lio_get_undistorted(void* h, float* out, int cap) {
    Lio* l = static_cast<Lio*>(h);
    const int n = (int)l->feats_undistort.size();
    if (n > cap) return -n;
    std::memcpy(out, l->feats_undistort.data(), sizeof(P4) * (size_t)n);
    return n;
}
int orc_lio_is_init(void* h) { return static_cast<Lio*>(h)->state_init_done ? 1 : 0; }  // ImuProcess::IsInit
void orc_lio_get_odometry(void* h, double* s26_start, double* s26_end) {
    Lio* l = static_cast<Lio*>(h);
    state_to(l->odom_start, s26_start);
    state_to(l->odom_end, s26_end);
}

// ---- undistortPoints(delta_pose, points, scan_period), slam/common/slam_utils.cpp:163-191 (localisation mode), f32 throughout.
// Per scan: Quaternionf(delta.block<3,3>(0,0)) (Eigen Quaternion.h quaternionbase_assign_impl<Other,3,3>), AngleAxisf(q) (AngleAxis.h:170-190;
// stableNorm below epsilon: StableNorm.h:18-50 on one dynamic 3-segment, summed in sequence).  Per point: r = (stamp / 1e6) / period as float,
// log = r * [t; angle * axis], rotation = Quaternionf(AngleAxisf(|w|, w / |w|)).toRotationMatrix(), p' = [R t] [p; 1] (Transform * vector is the
// 3x4 affine part times the homogeneous vector: fixed-size sum of four terms (x0 + x1) + (x2 + x3); pcl::transformPoint of PCL 1.9.1).
struct DeltaMotion {  // what undistortPoints derives once per delta pose: translation, angle * axis (f32)
    float tr[3], aa[3];
};
static DeltaMotion delta_motion(const float* D) {
    const float m[3][3] = {{D[0], D[1], D[2]}, {D[4], D[5], D[6]}, {D[8], D[9], D[10]}};
    float q[4];
    float t = m[0][0] + (m[1][1] + m[2][2]);
    if (t > 0.f) {
        t = std::sqrt(t + 1.0f); q[3] = 0.5f * t; t = 0.5f / t;
        q[0] = (m[2][1] - m[1][2]) * t; q[1] = (m[0][2] - m[2][0]) * t; q[2] = (m[1][0] - m[0][1]) * t;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0f); q[i] = 0.5f * t; t = 0.5f / t;
        q[3] = (m[k][j] - m[j][k]) * t; q[j] = (m[j][i] + m[i][j]) * t; q[k] = (m[k][i] + m[i][k]) * t;
    }
    float nrm = std::sqrt(q[0] * q[0] + (q[1] * q[1] + q[2] * q[2]));
    if (nrm < std::numeric_limits<float>::epsilon()) {
        const float mx = std::max(std::fabs(q[0]), std::max(std::fabs(q[1]), std::fabs(q[2])));
        float scale = 0.f, inv = 1.f, ssq = 0.f;
        if (mx > scale) {
            const float tmp = 1.f / mx;
            if (tmp > std::numeric_limits<float>::max()) { inv = std::numeric_limits<float>::max(); scale = 1.f / inv; }
            else { scale = mx; inv = tmp; }
        }
        if (scale > 0.f) { const float a = q[0] * inv, b = q[1] * inv, c = q[2] * inv; ssq = (a * a + b * b) + c * c; }
        nrm = scale * std::sqrt(ssq);
    }
    float angle = 0.f, axis[3] = {1.f, 0.f, 0.f};
    if (nrm != 0.f) {
        angle = 2.f * std::atan2(nrm, std::fabs(q[3]));
        if (q[3] < 0.f) nrm = -nrm;
        for (int i = 0; i < 3; i++) axis[i] = q[i] / nrm;
    }
    DeltaMotion M;
    for (int i = 0; i < 3; i++) { M.tr[i] = D[4 * i + 3]; M.aa[i] = angle * axis[i]; }
    return M;
}
static void undistort_one(const DeltaMotion& M, float* p, uint32_t stamp, double scan_period) {
    const float r = (float)(((double)stamp / 1000000.0) / scan_period);
    const float tx = r * M.tr[0], ty = r * M.tr[1], tz = r * M.tr[2];
    const float wx = r * M.aa[0], wy = r * M.aa[1], wz = r * M.aa[2];
    const float norm = std::sqrt(wx * wx + (wy * wy + wz * wz));
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (!((double)norm < 1e-8)) {
        const float ax = wx / norm, ay = wy / norm, az = wz / norm;
        const float ha = 0.5f * norm;
        const float w = std::cos(ha), sh = std::sin(ha);
        const float x = sh * ax, y = sh * ay, z = sh * az;
        const float t2x = 2.f * x, t2y = 2.f * y, t2z = 2.f * z;
        const float twx = t2x * w, twy = t2y * w, twz = t2z * w;
        const float txx = t2x * x, txy = t2y * x, txz = t2z * x;
        const float tyy = t2y * y, tyz = t2z * y, tzz = t2z * z;
        R[0] = 1.f - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
        R[3] = txy + twz; R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
        R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.f - (txx + tyy);
    }
    const float px = p[0], py = p[1], pz = p[2];
    p[0] = (R[0] * px + R[1] * py) + (R[2] * pz + tx);
    p[1] = (R[3] * px + R[4] * py) + (R[5] * pz + ty);
    p[2] = (R[6] * px + R[7] * py) + (R[8] * pz + tz);
}
void orc_undistort_delta(const float* D, float* xyzi, const uint32_t* stamp_us, int n, double scan_period) {
    const DeltaMotion M = delta_motion(D);
    for (int i = 0; i < n; i++) undistort_one(M, xyzi + 4 * (size_t)i, stamp_us[i], scan_period);
}
// undistortPoints(std::vector<PoseType>& poses, PointCloudAttrPtr&), slam_utils.cpp:193-228: poses[i].T (i >= 1) is the motion since
// poses[0]; the cloud is walked ONCE, a point belongs to the first pose interval (from the one the previous point used onwards) whose
// end, poses[i].timestamp - header.stamp in unsigned 64-bit arithmetic, is not before the point's stamp; points past the last pose and
// everything after them stay as they are
void orc_undistort_poses(const uint64_t* pose_stamp_us, const double* pose_T16, int n_poses, float* xyzi, const uint32_t* stamp_us, int n,
                         uint64_t header_us) {
    int idx = 0;
    for (int i = 1; i < n_poses; i++) {
        const double scan_period = (double)(pose_stamp_us[i] - pose_stamp_us[0]) / 1000000.0;
        float D[16];
        for (int k = 0; k < 16; k++) D[k] = (float)pose_T16[16 * (size_t)i + k];
        const DeltaMotion M = delta_motion(D);
        for (; idx < n; idx++) {
            if ((uint64_t)stamp_us[idx] > (uint64_t)(pose_stamp_us[i] - header_us)) break;
            undistort_one(M, xyzi + 4 * (size_t)idx, stamp_us[idx], scan_period);
        }
    }
}

void orc_so3_Exp(const double* w3, double dt, double* R9) { so3_Exp_rodrigues(V3{{w3[0], w3[1], w3[2]}}, dt, R9); }
void orc_undistort_point(const double* R_imu9, const double* vel3, const double* pos3, const double* acc3, const double* gyr3, double dt,
                         const float* p_xyz, const double* end_pos3, const double* end_rot_xyzw, const double* ril_xyzw, const double* til3, float* out_xyz) {
    auto v3 = [](const double* p) { return V3{{p[0], p[1], p[2]}}; };
    auto q4 = [](const double* p) { return Quat{p[0], p[1], p[2], p[3]}; };
    float q[3] = {p_xyz[0], p_xyz[1], p_xyz[2]};
    undistort_point(R_imu9, v3(vel3), v3(pos3), v3(acc3), v3(gyr3), dt, q, v3(end_pos3), q4(end_rot_xyzw), q4(ril_xyzw), v3(til3));
    out_xyz[0] = q[0]; out_xyz[1] = q[1]; out_xyz[2] = q[2];
}

// esekf::update_iterated_dyn_share_modified with an externally supplied measurement model
void orc_kf_update_cb(const double* s26, const double* P, double R, int max_iter, Lio::meas_fn fn, void* ctx, int cap, double* s26_out, double* P_out) {
    Lio l(0.5f, 19, 1000, 100.0);
    state_from(s26, l.x);
    for (int i = 0; i < 23; i++) for (int j = 0; j < 23; j++) l.P(i, j) = P[i * 23 + j];
    l.max_iter = max_iter;
    l.ext_fn = fn; l.ext_ctx = ctx; l.ext_cap = cap;
    l.update_iterated(R);
    state_to(l.x, s26_out);
    for (int i = 0; i < 23; i++) for (int j = 0; j < 23; j++) P_out[i * 23 + j] = l.P(i, j);
}

// the same with the wheel-speed rows of laserMapping.cpp:794-811 appended in every pass: ins_vel = Measures.ins.back() in the IMU frame
void orc_kf_update_ws_cb(const double* s26, const double* P, double R, int max_iter, Lio::meas_fn fn, void* ctx, int cap, const double* ins_vel, int degenerate,
                         double* s26_out, double* P_out) {
    Lio l(0.5f, 19, 1000, 100.0);
    state_from(s26, l.x);
    for (int i = 0; i < 23; i++) for (int j = 0; j < 23; j++) l.P(i, j) = P[i * 23 + j];
    l.max_iter = max_iter;
    l.ext_fn = fn; l.ext_ctx = ctx; l.ext_cap = cap;
    l.wheelspeed_en = true;
    l.meas_ins_valid = true;
    l.meas_ins_stamp = l.meas_lidar_end = 0.0;
    l.meas_ins_vel = V3{{ins_vel[0], ins_vel[1], ins_vel[2]}};
    l.is_degenerate = degenerate != 0;
    l.update_iterated(R);
    state_to(l.x, s26_out);
    for (int i = 0; i < 23; i++) for (int j = 0; j < 23; j++) P_out[i * 23 + j] = l.P(i, j);
}

// manifold helpers exposed for the known-answer tests
void orc_state_boxplus(const double* s26, const double* d23, double* out26) { State x; state_from(s26, x); state_boxplus(x, d23); state_to(x, out26); }
void orc_state_boxminus(const double* a26, const double* b26, double* d23) { State a, b; state_from(a26, a); state_from(b26, b); state_boxminus(a, b, d23); }
void orc_A_matrix(const double* v3, double* A9) { A_matrix(V3{{v3[0], v3[1], v3[2]}}, A9); }
void orc_S2_Bx(const double* v3, double* Bx6) { S2_Bx(V3{{v3[0], v3[1], v3[2]}}, Bx6); }
void orc_S2_Nx_yy(const double* v3, double* N6) { S2_Nx_yy(V3{{v3[0], v3[1], v3[2]}}, N6); }
void orc_S2_Mx(const double* v3, const double* d2, double* M6) { S2_Mx(V3{{v3[0], v3[1], v3[2]}}, d2, M6); }
void orc_eig3(const double* A9, double* w3, double* V9) { eig3(A9, w3, V9); }
void orc_inverse(const double* A, int n, double* out) {
    Mat M(n, n);
    for (int i = 0; i < n * n; i++) M.a[i] = A[i];
    Mat X = inverse(M);
    for (int i = 0; i < n * n; i++) out[i] = X.a[i];
}

}  // extern "C"
