// stand-in for <pcl/registration/registration.h> (PCL is not installed; third-party source not in the reference tree): the base class
// fast_gicp::LsqRegistration derives from, reduced to the members and the align() -> computeTransformation() hand-over that class and
// its callers use (PCL 1.9.1 registration.h / impl/registration.hpp: align() resets converged_ / the transformations and calls
// computeTransformation(output, guess)).
#pragma once
#include <Eigen/Core>
#include <memory>
#include <string>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#ifndef PCL_VERSION_CALC
#define PCL_VERSION_CALC(a, b, c) ((a) * 100000 + (b) * 100 + (c))
#define PCL_VERSION PCL_VERSION_CALC(1, 10, 0)
#endif
namespace pcl {
template <typename T>
using shared_ptr = std::shared_ptr<T>;
template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
   public:
    using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
    using PointCloudSource = pcl::PointCloud<PointSource>;
    using PointCloudSourcePtr = typename PointCloudSource::Ptr;
    using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
    using PointCloudTarget = pcl::PointCloud<PointTarget>;
    using PointCloudTargetPtr = typename PointCloudTarget::Ptr;
    using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
    Registration() : final_transformation_(Matrix4::Identity()) {}
    virtual ~Registration() {}
    virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; }
    virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) { target_ = cloud; }
    void setMaximumIterations(int n) { max_iterations_ = n; }
    void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
    void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
    double getMaxCorrespondenceDistance() const { return corr_dist_threshold_; }
    Matrix4 getFinalTransformation() const { return final_transformation_; }
    bool hasConverged() const { return converged_; }
    virtual double getFitnessScore(double /*max_range*/) { return 0.0; }  // PCL: kd-tree nearest neighbours of the aligned cloud; mocks override
    using Ptr = std::shared_ptr<Registration<PointSource, PointTarget, Scalar>>;
    const std::string& getClassName() const { return reg_name_; }
    void align(PointCloudSource& output, const Matrix4& guess = Matrix4::Identity()) {
        converged_ = false;
        final_transformation_ = Matrix4::Identity();
        computeTransformation(output, guess);
    }
    int nr_iterations_ = 0;  // public here: the harness reports it

   protected:
    virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
    std::string reg_name_;
    PointCloudSourceConstPtr input_;
    PointCloudTargetConstPtr target_;
    int max_iterations_ = 10;
    Matrix4 final_transformation_;
    double transformation_epsilon_ = 0.0;
    double corr_dist_threshold_ = 0.0;
    bool converged_ = false;
};
// pcl::transformPointCloud(in, out, Matrix4f) as PCL 1.9.1 computes it for XYZ points: out = in, then xyz = M(0..2, 0..3) * [x y z 1]
// accumulated left to right in f32
template <typename PointT, typename Scalar>
void transformPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, const Eigen::Matrix<Scalar, 4, 4>& M) {
    if (&in != &out) out = in;
    for (size_t i = 0; i < in.points.size(); i++) {
        const PointT& p = in.points[i];
        PointT q = p;
        q.x = static_cast<float>(M(0, 0) * p.x + M(0, 1) * p.y + M(0, 2) * p.z + M(0, 3));
        q.y = static_cast<float>(M(1, 0) * p.x + M(1, 1) * p.y + M(1, 2) * p.z + M(1, 3));
        q.z = static_cast<float>(M(2, 0) * p.x + M(2, 1) * p.y + M(2, 2) * p.z + M(2, 3));
        out.points[i] = q;
    }
}
}  // namespace pcl
