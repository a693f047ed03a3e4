// oracle/ref_gicp.cpp -- the reference's fine matcher fast_gicp::FastGICP<PointXYZI, PointXYZI> (slam/thirdparty/fast_gicp/include/fast_gicp/
// gicp/{fast_gicp.hpp, impl/fast_gicp_impl.hpp, lsq_registration.hpp, impl/lsq_registration_impl.hpp}, so3/so3.hpp) compiled from where
// the sources lie, configured as select_registration_method("FAST_GICP") configures it (backend/hdl_graph_slam/src/hdl_graph_slam/
// registrations.cpp:33-42).  Nothing is copied or edited.  PCL's Registration base class, PointCloud and the kd-tree are stand-ins
// (oracle/ref_shims: the kd-tree stand-in is an exact k-NN search like PCL's, see its header).
// TEST INFRASTRUCTURE ONLY: pins lio_gicp_* (csrc/gicp.hip).  Built into oracle/_ref/libref_gicp.so by `make -C oracle ref`.
#include <omp.h>
#include <iostream>
#include <pcl/search/kdtree.h>
#include <fast_gicp/gicp/fast_gicp.hpp>
#include <fast_gicp/gicp/impl/fast_gicp_impl.hpp>
#include <fast_gicp/gicp/impl/lsq_registration_impl.hpp>
// ... and its voxelised sibling fast_gicp::FastVGICP (fast_vgicp.hpp, impl/fast_vgicp_impl.hpp, fast_vgicp_voxel.hpp), configured as
// select_registration_method("FAST_VGICP") (registrations.cpp:56-66); boost::hash_combine is a stand-in
#include <fast_gicp/gicp/fast_vgicp.hpp>
#include <fast_gicp/gicp/impl/fast_vgicp_impl.hpp>

using PointT = pcl::PointXYZI;
struct RefGicp : public fast_gicp::FastGICP<PointT, PointT> {
    using Base = fast_gicp::FastGICP<PointT, PointT>;
    using Base::compute_error;
    using Base::correspondences_;
    using Base::linearize;
    using Base::mahalanobis_;
    using Base::nr_iterations_;
    using Base::sq_distances_;
    pcl::PointCloud<PointT>::Ptr src, tgt;  // keep the clouds alive
};

struct RefVgicp : public fast_gicp::FastVGICP<PointT, PointT> {
    using Base = fast_gicp::FastVGICP<PointT, PointT>;
    using Base::compute_error;
    using Base::linearize;
    using Base::nr_iterations_;
    using Base::voxel_correspondences_;
    using Base::voxelmap_;
    pcl::PointCloud<PointT>::Ptr src, tgt;
};

static pcl::PointCloud<PointT>::Ptr to_pcl(const float* xyzi, int n) {
    pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>());
    c->points.resize(n);
    for (int i = 0; i < n; i++) { c->points[i].x = xyzi[4 * i]; c->points[i].y = xyzi[4 * i + 1]; c->points[i].z = xyzi[4 * i + 2]; c->points[i].intensity = xyzi[4 * i + 3]; }
    c->width = n; c->height = 1;
    return c;
}
static Eigen::Isometry3d to_iso(const double* T16) {
    Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) T.matrix()(r, c) = T16[4 * r + c];
    return T;
}
static void put_covs(const std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>>& covs, double* cov9) {
    for (size_t i = 0; i < covs.size(); i++)
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) cov9[i * 9 + r * 3 + c] = covs[i](r, c);
}

extern "C" {
void* ref_gicp_create(int k, double max_corr_dist, double transformation_epsilon, int max_iterations, int num_threads, float kdtree_cell) {
    pcl::search::KdTree<PointT>::cell_size() = kdtree_cell > 0 ? kdtree_cell : 1.0f;
    RefGicp* g = new RefGicp();
    g->setNumThreads(num_threads);
    g->setTransformationEpsilon(transformation_epsilon);
    g->setMaximumIterations(max_iterations);
    g->setMaxCorrespondenceDistance(max_corr_dist);
    g->setCorrespondenceRandomness(k);
    return g;
}
void ref_gicp_destroy(void* h) { delete static_cast<RefGicp*>(h); }
// setInputTarget / setInputSource (calculate_covariances); cov9 = the upper-left 3 x 3 of every 4 x 4 covariance, row-major, input order
void ref_gicp_set_target(void* h, const float* xyzi, int n, double* cov9) {
    RefGicp* g = static_cast<RefGicp*>(h);
    g->tgt = to_pcl(xyzi, n);
    g->setInputTarget(g->tgt);
    if (cov9) put_covs(g->getTargetCovariances(), cov9);
}
void ref_gicp_set_source(void* h, const float* xyzi, int n, double* cov9) {
    RefGicp* g = static_cast<RefGicp*>(h);
    g->src = to_pcl(xyzi, n);
    g->setInputSource(g->src);
    if (cov9) put_covs(g->getSourceCovariances(), cov9);
}
// linearize(trans, &H, &b) (update_correspondences inside); corr / sq_dist / maha9 (optional) receive the per-source-point state
double ref_gicp_linearize(void* h, const double* T16, double* H36, double* b6, int* corr, float* sq_dist, double* maha9) {
    RefGicp* g = static_cast<RefGicp*>(h);
    Eigen::Matrix<double, 6, 6> H;
    Eigen::Matrix<double, 6, 1> b;
    const double e = g->linearize(to_iso(T16), &H, &b);
    for (int r = 0; r < 6; r++) {
        for (int c = 0; c < 6; c++) H36[r * 6 + c] = H(r, c);
        b6[r] = b(r);
    }
    const size_t n = g->correspondences_.size();
    for (size_t i = 0; i < n; i++) {
        if (corr) corr[i] = g->correspondences_[i];
        if (sq_dist) sq_dist[i] = g->sq_distances_[i];
        if (maha9 && g->correspondences_[i] >= 0)
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) maha9[i * 9 + r * 3 + c] = g->mahalanobis_[i](r, c);
    }
    return e;
}
double ref_gicp_compute_error(void* h, const double* T16) { return static_cast<RefGicp*>(h)->compute_error(to_iso(T16)); }
// registration->align(aligned, guess)
int ref_gicp_align(void* h, const float* guess16, float* T16, int* iterations) {
    RefGicp* g = static_cast<RefGicp*>(h);
    Eigen::Matrix4f G;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) G(r, c) = guess16[4 * r + c];
    pcl::PointCloud<PointT> aligned;
    g->align(aligned, G);
    const Eigen::Matrix4f T = g->getFinalTransformation();
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) T16[4 * r + c] = T(r, c);
    if (iterations) *iterations = g->nr_iterations_;
    return g->hasConverged() ? 1 : 0;
}
// trans.cast<float>() * [x y z 1] exactly as update_correspondences evaluates it (for the operation-order check of the device kernel)
void ref_gicp_transform_f(const double* T16, const float* xyz, float* out) {
    const Eigen::Isometry3f trans_f = to_iso(T16).cast<float>();
    PointT p; p.x = xyz[0]; p.y = xyz[1]; p.z = xyz[2];
    PointT q;
    q.getVector4fMap() = trans_f * p.getVector4fMap();
    out[0] = q.x; out[1] = q.y; out[2] = q.z;
}

// ---- FastVGICP ----
void* ref_vgicp_create(int k, double resolution, int search_method, double transformation_epsilon, double rotation_epsilon, int max_iterations, int num_threads,
                       float kdtree_cell) {
    pcl::search::KdTree<PointT>::cell_size() = kdtree_cell > 0 ? kdtree_cell : 1.0f;
    RefVgicp* g = new RefVgicp();
    g->setNumThreads(num_threads);
    g->setResolution(resolution);
    g->setTransformationEpsilon(transformation_epsilon);
    g->setRotationEpsilon(rotation_epsilon);
    g->setMaximumIterations(max_iterations);
    g->setCorrespondenceRandomness(k);
    g->setNeighborSearchMethod(search_method == 7 ? fast_gicp::NeighborSearchMethod::DIRECT7
                               : search_method == 27 ? fast_gicp::NeighborSearchMethod::DIRECT27 : fast_gicp::NeighborSearchMethod::DIRECT1);
    return g;
}
void ref_vgicp_destroy(void* h) { delete static_cast<RefVgicp*>(h); }
void ref_vgicp_set_target(void* h, const float* xyzi, int n) {
    RefVgicp* g = static_cast<RefVgicp*>(h);
    g->tgt = to_pcl(xyzi, n);
    g->setInputTarget(g->tgt);
}
void ref_vgicp_set_source(void* h, const float* xyzi, int n) {
    RefVgicp* g = static_cast<RefVgicp*>(h);
    g->src = to_pcl(xyzi, n);
    g->setInputSource(g->src);
}
// linearize(trans, &H, &b); returns the error; n_corr = voxel correspondences found
double ref_vgicp_linearize(void* h, const double* T16, double* H36, double* b6, int* n_corr) {
    RefVgicp* g = static_cast<RefVgicp*>(h);
    Eigen::Matrix<double, 6, 6> H;
    Eigen::Matrix<double, 6, 1> b;
    const double e = g->linearize(to_iso(T16), &H, &b);
    for (int r = 0; r < 6; r++) {
        for (int c = 0; c < 6; c++) H36[r * 6 + c] = H(r, c);
        b6[r] = b(r);
    }
    if (n_corr) *n_corr = (int)g->voxel_correspondences_.size();
    return e;
}
double ref_vgicp_compute_error(void* h, const double* T16) { return static_cast<RefVgicp*>(h)->compute_error(to_iso(T16)); }
// the Gaussian voxel holding p (after a linearize built the map): number of points, mean[3], cov9; 0 = none
int ref_vgicp_voxel_at(void* h, const float* p, double* mean3, double* cov9) {
    RefVgicp* g = static_cast<RefVgicp*>(h);
    if (!g->voxelmap_) return -1;
    const Eigen::Vector4d x(p[0], p[1], p[2], 1.0);
    auto v = g->voxelmap_->lookup_voxel(g->voxelmap_->voxel_coord(x));
    if (!v) return 0;
    for (int r = 0; r < 3; r++) {
        mean3[r] = v->mean[r];
        for (int c = 0; c < 3; c++) cov9[r * 3 + c] = v->cov(r, c);
    }
    return v->num_points;
}
int ref_vgicp_align(void* h, const float* guess16, float* T16, int* iterations) {
    RefVgicp* g = static_cast<RefVgicp*>(h);
    Eigen::Matrix4f G;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) G(r, c) = guess16[4 * r + c];
    pcl::PointCloud<PointT> aligned;
    g->align(aligned, G);
    const Eigen::Matrix4f T = g->getFinalTransformation();
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) T16[4 * r + c] = T(r, c);
    if (iterations) *iterations = g->nr_iterations_;
    return g->hasConverged() ? 1 : 0;
}
}
