// stand-in for <pcl/filters/voxel_grid.h> for ONE harness (ref_voxelgrid_cov.cpp): only the BASE the reference's own
// pclomp::VoxelGridCovariance (slam/thirdparty/ndt_omp, PCL's VoxelGridCovariance with OpenMP) derives from -- the data members its
// applyFilter names and setLeafSize as PCL 1.9.1 states it (inverse_leaf_size_ = 1 / leaf_size_ in f32).  No filtering happens here: the
// statements that are pinned (bounding box, overflow guard, keys, per-leaf sums) are the reference file's own.
#pragma once
#include <cfloat>
#include <memory>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <pcl/common/common.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
namespace pcl {
namespace traits {
template <typename PointT>
struct fieldList { struct type { static constexpr int value = 4; }; };  // PointXYZI: x, y, z, intensity
}  // namespace traits
// the two field functors of <pcl/point_traits.h> for the one point type of the harness: every float field, in declaration order
template <typename PointT>
struct NdCopyPointEigenFunctor {
    const PointT& p1_;
    Eigen::VectorXf& p2_;
    NdCopyPointEigenFunctor(const PointT& p1, Eigen::VectorXf& p2) : p1_(p1), p2_(p2) {}
    void all() { p2_[0] = p1_.x; p2_[1] = p1_.y; p2_[2] = p1_.z; p2_[3] = p1_.intensity; }
};
template <typename PointT>
struct NdCopyEigenPointFunctor {
    const Eigen::VectorXf& p1_;
    PointT& p2_;
    NdCopyEigenPointFunctor(const Eigen::VectorXf& p1, PointT& p2) : p1_(p1), p2_(p2) {}
    void all() { p2_.x = p1_[0]; p2_.y = p1_[1]; p2_.z = p1_[2]; p2_.intensity = p1_[3]; }
};
template <typename FieldList, typename F>
inline void for_each_type(F f) { f.all(); }
// (named by getNeighborhoodAtPoint, which the harness never calls)
inline Eigen::MatrixXi getAllNeighborCellIndices() { return Eigen::MatrixXi::Zero(4, 27); }

template <typename PointT>
class Filter {
   public:
    using PointCloud = pcl::PointCloud<PointT>;
    using PointCloudPtr = typename PointCloud::Ptr;
    using PointCloudConstPtr = typename PointCloud::ConstPtr;
    virtual ~Filter() {}
    void setInputCloud(const PointCloudConstPtr& c) { input_ = c; }
    void filter(PointCloud& out) { applyFilter(out); }

   protected:
    virtual void applyFilter(PointCloud& out) = 0;
    const std::string& getClassName() const { return filter_name_; }
    std::string filter_name_;
    PointCloudConstPtr input_;
    std::shared_ptr<std::vector<int>> indices_;
};
template <typename PointT>
class VoxelGrid : public Filter<PointT> {
   public:
    using PointCloud = typename Filter<PointT>::PointCloud;
    VoxelGrid() : leaf_size_(Eigen::Vector4f::Zero()), inverse_leaf_size_(Eigen::Array4f::Zero()), downsample_all_data_(true), save_leaf_layout_(false),
                  min_b_(Eigen::Vector4i::Zero()), max_b_(Eigen::Vector4i::Zero()), div_b_(Eigen::Vector4i::Zero()), divb_mul_(Eigen::Vector4i::Zero()),
                  filter_limit_min_(-FLT_MAX), filter_limit_max_(FLT_MAX), filter_limit_negative_(false) { this->filter_name_ = "VoxelGrid"; }
    // voxel_grid.h (PCL 1.9.1): leaf_size_ = (lx, ly, lz, 1); inverse_leaf_size_ = Eigen::Array4f::Ones () / leaf_size_.array ()
    void setLeafSize(float lx, float ly, float lz) {
        leaf_size_[0] = lx; leaf_size_[1] = ly; leaf_size_[2] = lz;
        if (leaf_size_[3] == 0) leaf_size_[3] = 1;
        inverse_leaf_size_ = Eigen::Array4f::Ones() / leaf_size_.array();
    }
    void setDownsampleAllData(bool d) { downsample_all_data_ = d; }
    Eigen::Vector3i getMinBoxCoordinates() const { return min_b_.template head<3>(); }
    Eigen::Vector3i getNrDivisions() const { return div_b_.template head<3>(); }

   protected:
    void applyFilter(PointCloud&) override {}
    Eigen::Vector4f leaf_size_;
    Eigen::Array4f inverse_leaf_size_;
    bool downsample_all_data_, save_leaf_layout_;
    std::vector<int> leaf_layout_;
    Eigen::Vector4i min_b_, max_b_, div_b_, divb_mul_;
    std::string filter_field_name_;
    double filter_limit_min_, filter_limit_max_;
    bool filter_limit_negative_;
};
}  // namespace pcl
