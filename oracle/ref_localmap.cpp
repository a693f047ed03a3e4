// ref_localmap.cpp -- the reference's OWN text for two stages that cannot be compiled as whole translation units here (their classes pull in
// PCL / g2o / the whole localisation module), cut out of the files where it lies at build time and compiled inside this harness:
//   * the body of Localization::runUpdateLocalMap's loop -- /root/reference/slam/localization/src/localization.cpp:305-312 (constants, VoxelGrid
//     set-up) and :325-372 (skip test, key-frame radius search, nearest-first thinning, concatenation to 200 000 points, VoxelGrid, the far /
//     out-of-map branches, the hand-over to the localizer) -> _ref/obj/localmap_consts.inc, _ref/obj/localmap_body.inc   
//   * OverlapDetector::filter and ::calc_fitness_score -- /root/reference/slam/localization/include/overlap_merge.hpp:213-263
//     -> _ref/obj/overlap_fitness.inc                                                                                     
// (oracle/Makefile `_ref/libref_localmap.so` writes the excerpts with sed; nothing of the reference is copied into this repository.)
// Around them: the members / types the excerpts name, as plain globals -- key frames with mTransfromPoints, the key-frame position tree
// (pcl::KdTreeFLANN stand-in: exact, sorted by distance as FLANN returns it), mConfig, mLocalMap, a localizer that records what it is handed.
// pcl::VoxelGrid is routed to the oracle's restatement (ref_shims/pcl/filters/voxel_grid.h: pcl::VoxelGrid itself is not in the tree; the restatement is pinned by ref_voxelgrid_cov.cpp),
// pcl::transformPointCloud / pcl::search::KdTree are the stand-ins the GICP harness uses.  Test infrastructure only.
#include <unistd.h>

#include <Eigen/Geometry>
#include <cmath>
#include <limits>
#include <memory>
#include <mutex>
#include <vector>

#include <pcl/filters/voxel_grid.h>
#include <pcl/kdtree/kdtree_flann.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/registration.h>  // transformPointCloud stand-in
#include <pcl/search/kdtree.h>

typedef pcl::PointXYZI Point;
typedef pcl::PointCloud<Point> PointCloud;

static int g_warn_far = 0, g_warn_few = 0, g_err_out = 0, g_info = 0;
#define LOG_INFO(...) (g_info++)
#define LOG_ERROR(...) (g_err_out++)
// the two LOG_WARN call sites are told apart by their format strings (both are string literals in the excerpt)
static void count_warn(const char* fmt) { (fmt[14] == 'n' ? g_warn_far : g_warn_few)++; }  // "Localization: nearest ..." / "Localization: local ..."
#define LOG_WARN(fmt, ...) count_warn(fmt)
#define usleep(x) ((void)0)

namespace {
struct KeyFrame {
    PointCloud::Ptr mTransfromPoints;
};
struct Config {
    double resolution = 0.2;
    double key_frame_distance = 1.0;
} mConfig;
struct Localizer {
    int calls = 0;
    PointCloud::Ptr last;
    void updateLocalMap(PointCloud::Ptr& m) { calls++; last = m; }
};
std::vector<std::shared_ptr<KeyFrame>> mKeyFrames;
pcl::KdTreeFLANN<pcl::PointXYZ>::Ptr mGraphKDTree;
pcl::PointCloud<pcl::PointXYZ>::Ptr g_positions;
PointCloud::Ptr mLocalMap;
std::unique_ptr<Localizer> mLocalizer;
Eigen::Isometry3d lastPose = Eigen::Isometry3d::Identity();  // localization.cpp:314 (a local of the thread function: lives across loop turns)
std::vector<int> g_last_selection;

// one turn of the loop body for a dequeued pose
int loop_turn(const Eigen::Isometry3d& pose) {
#include "_ref/obj/localmap_consts.inc"
    (void)min_local_map_points_num;
#include "_ref/obj/localmap_body.inc"
    return 0;
}

struct OverlapExcerpt {
    typedef pcl::PointXYZI PointT;
#include "_ref/obj/overlap_fitness.inc"
};
}  // namespace

extern "C" {

void ref_lm_reset(double resolution, double key_frame_distance) {
    mKeyFrames.clear();
    g_positions.reset(new pcl::PointCloud<pcl::PointXYZ>());
    mGraphKDTree.reset(new pcl::KdTreeFLANN<pcl::PointXYZ>());
    mGraphKDTree->setInputCloud(g_positions);
    mLocalMap = nullptr;
    mLocalizer.reset(new Localizer());
    lastPose = Eigen::Isometry3d::Identity();
    mConfig.resolution = resolution;
    mConfig.key_frame_distance = key_frame_distance;
    g_warn_far = g_warn_few = g_err_out = g_info = 0;
}

int ref_lm_add_keyframe(const float* xyzi, int n, const float pos[3]) {
    auto kf = std::make_shared<KeyFrame>();
    kf->mTransfromPoints.reset(new PointCloud());
    kf->mTransfromPoints->points.resize(n);
    for (int i = 0; i < n; i++) {
        Point p;
        p.x = xyzi[4 * i]; p.y = xyzi[4 * i + 1]; p.z = xyzi[4 * i + 2]; p.intensity = xyzi[4 * i + 3];
        kf->mTransfromPoints->points[i] = p;
    }
    mKeyFrames.push_back(kf);
    pcl::PointXYZ q;
    q.x = pos[0]; q.y = pos[1]; q.z = pos[2];
    g_positions->points.push_back(q);
    mGraphKDTree->setInputCloud(g_positions);
    return (int)mKeyFrames.size() - 1;
}

// one loop turn; returns the product's codes -- 0 nothing to do (moved less than update_distance), 1 local map replaced, 2 out of map,
// 3 nearest key frame >= 20 m away -- derived from what the excerpt did: the localizer's call count, the map it was handed, the log calls
int ref_lm_update(const double pose_xyz[3]) {
    Eigen::Isometry3d pose = Eigen::Isometry3d::Identity();
    pose.translation() = Eigen::Vector3d(pose_xyz[0], pose_xyz[1], pose_xyz[2]);
    const int calls0 = mLocalizer->calls, out0 = g_err_out, far0 = g_warn_far;
    loop_turn(pose);
    if (mLocalizer->calls == calls0) return 0;
    if (g_err_out > out0) return 2;
    if (g_warn_far > far0) return 3;
    return 1;
}

int ref_lm_local_map(float* out_xyzi, int cap) {  // what the localizer was handed last (-1: a null map)
    if (!mLocalizer->last) return -1;
    const int n = (int)mLocalizer->last->points.size();
    if (n > cap) return -n - 2;
    for (int i = 0; i < n; i++) {
        const Point& p = mLocalizer->last->points[i];
        out_xyzi[4 * i] = p.x; out_xyzi[4 * i + 1] = p.y; out_xyzi[4 * i + 2] = p.z; out_xyzi[4 * i + 3] = p.intensity;
    }
    return n;
}

// OverlapDetector::calc_fitness_score(cloud1, cloud2, relpose, max_range) -> (score, inlier ratio)
void ref_overlap_fitness(const float* c1, int n1, const float* c2, int n2, const float relpose16[16], double max_range, double out[2]) {
    auto mk = [](const float* c, int n) {
        PointCloud::Ptr pc(new PointCloud());
        pc->points.resize(n);
        for (int i = 0; i < n; i++) {
            Point p;
            p.x = c[4 * i]; p.y = c[4 * i + 1]; p.z = c[4 * i + 2]; p.intensity = c[4 * i + 3];
            pc->points[i] = p;
        }
        pc->width = n; pc->height = 1;
        return pc;
    };
    const Eigen::Matrix4f M = Eigen::Map<const Eigen::Matrix<float, 4, 4, Eigen::RowMajor>>(relpose16);
    OverlapExcerpt ex;
    const std::pair<double, double> r = ex.calc_fitness_score(mk(c1, n1), mk(c2, n2), M, max_range);
    out[0] = r.first;
    out[1] = r.second;
}

}  // extern "C"
