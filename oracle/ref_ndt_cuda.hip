// oracle/ref_ndt_cuda.hip -- the reference's OWN GPU matcher core, fast_gicp::cuda::NDTCudaCore (slam/thirdparty/fast_gicp/src/fast_gicp/
// cuda/{ndt_cuda, gaussian_voxelmap, covariance_regularization, find_voxel_correspondences, ndt_compute_derivatives}.cu), compiled from
// where the sources lie for gfx950 with hipcc: the files are CUDA + Thrust; rocThrust provides thrust::, and the six CUDA runtime names
// they use are spelled in HIP by a force-included header (oracle/ref_shims/cuda/cuda_to_hip.h).  Nothing is copied or edited.
// TEST INFRASTRUCTURE ONLY (it needs a GPU to run: used by the -m gpu tests to pin lio_ndt_* against the reference's own kernels);
// built into oracle/_ref/libref_ndt_cuda.so by `make -C oracle ref`.
#include <fast_gicp/cuda/gaussian_voxelmap.cuh>
#include <fast_gicp/cuda/ndt_cuda.cuh>
// ... and the registration object the reference instantiates around it: fast_gicp::NDTCuda<PointT, PointT> with LsqRegistration's
// Levenberg-Marquardt loop (ndt_cuda_impl.hpp, lsq_registration_impl.hpp); PCL's Registration base class, PointCloud and
// boost::format are shims (oracle/ref_shims)
#include <fast_gicp/ndt/impl/ndt_cuda_impl.hpp>
#include <fast_gicp/gicp/impl/lsq_registration_impl.hpp>

#include <thrust/host_vector.h>

using fast_gicp::cuda::NDTCudaCore;

static Eigen::Isometry3d to_iso(const double* T16) {
    Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) T.matrix()(r, c) = T16[4 * r + c];
    return T;
}
static std::vector<Eigen::Vector3f, Eigen::aligned_allocator<Eigen::Vector3f>> to_cloud(const float* xyzi, int n) {
    std::vector<Eigen::Vector3f, Eigen::aligned_allocator<Eigen::Vector3f>> c(n);
    for (int i = 0; i < n; i++) c[i] = Eigen::Vector3f(xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2]);
    return c;
}

using RefNdt = fast_gicp::NDTCuda<pcl::PointXYZI, pcl::PointXYZI>;
static pcl::PointCloud<pcl::PointXYZI>::Ptr to_pcl(const float* xyzi, int n) {
    pcl::PointCloud<pcl::PointXYZI>::Ptr c(new pcl::PointCloud<pcl::PointXYZI>());
    c->points.resize(n);
    for (int i = 0; i < n; i++) { c->points[i].x = xyzi[4 * i]; c->points[i].y = xyzi[4 * i + 1]; c->points[i].z = xyzi[4 * i + 2]; c->points[i].intensity = xyzi[4 * i + 3]; }
    c->width = n; c->height = 1;
    return c;
}

extern "C" {
// select_registration_method("NDT_CUDA") of backend/hdl_graph_slam/src/hdl_graph_slam/registrations.cpp:107-118, same setter calls
void* ref_ndtreg_create(double resolution, int search_method, double max_process_time_ms) {
    RefNdt* r = new RefNdt();
    r->setTransformationEpsilon(0.01);
    r->setRotationEpsilon(0.1);
    r->setMaximumIterations(64);
    r->setResolution(resolution);
    r->setDistanceMode(fast_gicp::NDTDistanceMode::P2D);
    r->setNeighborSearchMethod(search_method == 1 ? fast_gicp::NeighborSearchMethod::DIRECT1
                               : search_method == 27 ? fast_gicp::NeighborSearchMethod::DIRECT27 : fast_gicp::NeighborSearchMethod::DIRECT7, 0.0);
    r->setMaxProcessTime((int64_t)max_process_time_ms);
    return r;
}
void ref_ndtreg_destroy(void* h) { delete static_cast<RefNdt*>(h); }
void ref_ndtreg_set_target(void* h, const float* xyzi, int n) { static_cast<RefNdt*>(h)->setInputTarget(to_pcl(xyzi, n)); }
void ref_ndtreg_set_source(void* h, const float* xyzi, int n) { static_cast<RefNdt*>(h)->setInputSource(to_pcl(xyzi, n)); }
// registration->align(aligned, guess); returns hasConverged, fills the final transformation (row-major) and the iteration count
int ref_ndtreg_align(void* h, const float* guess16, float* T16, int* iterations) {
    RefNdt* r = static_cast<RefNdt*>(h);
    Eigen::Matrix4f G;
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) G(a, b) = guess16[4 * a + b];
    pcl::PointCloud<pcl::PointXYZI> aligned;
    r->align(aligned, G);
    const Eigen::Matrix4f T = r->getFinalTransformation();
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) T16[4 * a + b] = T(a, b);
    if (iterations) *iterations = static_cast<pcl::Registration<pcl::PointXYZI, pcl::PointXYZI, float>*>(r)->nr_iterations_;
    return r->hasConverged() ? 1 : 0;
}

// search_method: 1 / 7 / 27 (DIRECT1 / DIRECT7 / DIRECT27); distance mode P2D as select_registration_method("NDT_CUDA") sets it
void* ref_ndt_create(double resolution, int search_method) {
    NDTCudaCore* c = new NDTCudaCore();
    c->set_resolution(resolution);
    c->set_distance_mode(fast_gicp::NDTDistanceMode::P2D);
    c->set_neighbor_search_method(search_method == 1 ? fast_gicp::NeighborSearchMethod::DIRECT1
                                  : search_method == 27 ? fast_gicp::NeighborSearchMethod::DIRECT27 : fast_gicp::NeighborSearchMethod::DIRECT7, 0.0);
    return c;
}
void ref_ndt_destroy(void* h) { delete static_cast<NDTCudaCore*>(h); }
void ref_ndt_set_target(void* h, const float* xyzi, int n) { static_cast<NDTCudaCore*>(h)->set_target_cloud(to_cloud(xyzi, n)); }
void ref_ndt_set_source(void* h, const float* xyzi, int n) {
    NDTCudaCore* c = static_cast<NDTCudaCore*>(h);
    c->set_source_cloud(to_cloud(xyzi, n));
    c->create_source_voxelmap();  // no-op in P2D
}
int ref_ndt_num_voxels(void* h) { return static_cast<NDTCudaCore*>(h)->target_voxelmap->voxelmap_info.num_voxels; }
// every target voxel: integer coordinate, point count, mean, covariance as the derivative kernels read it (after PLANE regularisation)
int ref_ndt_voxels(void* h, int* coord3, int* num_points, float* mean3, float* cov9, int cap) {
    NDTCudaCore* c = static_cast<NDTCudaCore*>(h);
    const auto& vm = *c->target_voxelmap;
    thrust::host_vector<thrust::pair<Eigen::Vector3i, int>> buckets = vm.buckets;
    thrust::host_vector<int> np = vm.num_points;
    thrust::host_vector<Eigen::Vector3f> means = vm.voxel_means;
    thrust::host_vector<Eigen::Matrix3f> covs = vm.voxel_covs;
    int out = 0;
    for (size_t b = 0; b < buckets.size(); b++) {
        const int v = buckets[b].second;
        if (v < 0) continue;
        if (out < cap) {
            for (int k = 0; k < 3; k++) { coord3[3 * out + k] = buckets[b].first[k]; mean3[3 * out + k] = means[v][k]; }
            num_points[out] = np[v];
            for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) cov9[9 * out + 3 * r + k] = covs[v](r, k);
        }
        out++;
    }
    return out;
}
// NDTCuda::linearize = update_correspondences(T) then compute_error(T, H, b) (ndt_cuda_impl.hpp:82-86); returns the number of pairs
int ref_ndt_update_correspondences(void* h, const double* T16) {
    NDTCudaCore* c = static_cast<NDTCudaCore*>(h);
    c->update_correspondences(to_iso(T16));
    return (int)c->correspondences->size();
}
double ref_ndt_compute_error(void* h, const double* T16, double* H36, double* b6) {
    NDTCudaCore* c = static_cast<NDTCudaCore*>(h);
    Eigen::Matrix<double, 6, 6> H;
    Eigen::Matrix<double, 6, 1> b;
    const double e = c->compute_error(to_iso(T16), H36 ? &H : nullptr, H36 ? &b : nullptr);
    if (H36)
        for (int r = 0; r < 6; r++) {
            for (int k = 0; k < 6; k++) H36[6 * r + k] = H(r, k);
            b6[r] = b(r);
        }
    return e;
}
// the valid pairs as (source index, target voxel index) -> count of pairs whose voxel is >= 0
int ref_ndt_valid_pairs(void* h) {
    NDTCudaCore* c = static_cast<NDTCudaCore*>(h);
    thrust::host_vector<thrust::pair<int, int>> p = *c->correspondences;
    int n = 0;
    for (size_t i = 0; i < p.size(); i++) n += (p[i].first >= 0 && p[i].second >= 0) ? 1 : 0;
    return n;
}
}
