// ref_voxelgrid_cov.cpp -- TEST INFRASTRUCTURE: the reference tree's own PCL-derived voxel filter, compiled from where it lies.
//
// pcl::VoxelGrid itself is third-party source that is neither installed nor in the reference tree.  But the reference vendors ndt_omp, whose
// pclomp::VoxelGridCovariance (slam/thirdparty/ndt_omp/include/pclomp/voxel_grid_covariance_omp_impl.hpp:49-330) is PCL's
// VoxelGridCovariance::applyFilter -- and that function carries, statement for statement, the parts of pcl::VoxelGrid::applyFilter the
// oracle's restatement has to get right:
//   * the bounding box (getMinMax3D), the int64 overflow guard on (max - min) * inverse_leaf_size + 1 and the INT32_MAX test (:73-83);
//   * min_b_ / max_b_ = floor(min_p * inverse_leaf_size), div_b_, divb_mul_ (:85-101);
//   * the voxel key of a point, int(floor(x * inverse_leaf_size) - float(min_b)) and its dot product with divb_mul_ (:217-222), the skip of
//     non-finite points in a cloud that is not dense (:211-215);
//   * per-leaf f32 sums of ALL fields in INPUT order (downsample_all_data_: NdCopyPointEigenFunctor + `leaf.centroid += centroid`, :248-262),
//     divided by float(nr_points) (:289), leaves visited in ascending key order (std::map).
// What it does not carry: pcl::VoxelGrid sorts an index vector by key and walks it (an unstable std::sort: the order of the addends inside a
// voxel is then implementation-defined; the oracle fixes input order, which is what this class does), and on the overflow guard VoxelGrid
// returns the input cloud where this class returns an empty one.  Base-class data members, setLeafSize and getMinMax3D are stand-ins
// (ref_shims_vgc/): plain member storage, 1 / leaf in f32, min / max.
#include <cstdint>
#include <cstring>

#include <pclomp/voxel_grid_covariance_omp_impl.hpp>

extern "C" {
// returns the number of leaves (every leaf, whatever its point count), or -1 when the overflow guard fired.  Leaves in ascending key order:
// key, point count, the four f32 centroids (x, y, z, intensity).  min_b3 / div_b3: the box the keys refer to.
int ref_vgc_filter(const float* xyzi, int n, float leaf, int is_dense, int64_t* keys, int* counts, float* centroids4, int cap, int* min_b3, int* div_b3) {
    typedef pcl::PointXYZI P;
    pcl::PointCloud<P>::Ptr cloud(new pcl::PointCloud<P>());
    cloud->points.resize(n);
    for (int i = 0; i < n; i++) {
        P p;
        p.x = xyzi[4 * i]; p.y = xyzi[4 * i + 1]; p.z = xyzi[4 * i + 2]; p.intensity = xyzi[4 * i + 3];
        cloud->points[i] = p;
    }
    cloud->width = (uint32_t)n; cloud->height = 1;
    cloud->is_dense = is_dense != 0;
    pclomp::VoxelGridCovariance<P> vg;
    vg.setDownsampleAllData(true);  // pcl::VoxelGrid's default (the class turns it off in its constructor): every field is averaged, intensity included
    vg.setLeafSize(leaf, leaf, leaf);
    vg.setInputCloud(cloud);
    pcl::PointCloud<P> out;
    vg.filter(out, false);
    const auto& leaves = vg.getLeaves();
    if (leaves.empty() && n > 0) {
        // the guard (or a cloud without a finite point): tell them apart by the guard's own arithmetic being re-run by the caller; here: -1
        return -1;
    }
    const Eigen::Vector3i mb = vg.getMinBoxCoordinates(), db = vg.getNrDivisions();
    for (int k = 0; k < 3; k++) { min_b3[k] = mb[k]; div_b3[k] = db[k]; }
    int m = 0;
    for (auto it = leaves.begin(); it != leaves.end(); ++it, ++m) {
        if (m >= cap) return -2;
        keys[m] = (int64_t)it->first;
        counts[m] = it->second.nr_points;
        for (int c = 0; c < 4; c++) centroids4[4 * m + c] = it->second.centroid[c];
    }
    return m;
}
}
