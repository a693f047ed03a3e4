// undistort.hip -- Preprocess::velodyne_handler's point filter + ImuProcess::UndistortPcl's backward propagation
// (/root/reference/slam/mapping/fastlio/src/preprocess.cpp:395-451, IMU_Processing.hpp:371-404) as one pass over the
// raw scan in HBM.
//
// The reference sorts the cloud by per-point time and walks it backwards with a moving IMU-segment cursor; on the
// device every point finds its segment itself (<= 128 IMU poses, staged in LDS) and is compensated independently:
//   P_c = R_LI^T ( R_e^T ( R_h Exp(w_t dt) (R_LI P + t_LI) + p_h + v_h dt + a_t dt^2 / 2 - p_e ) - t_LI )
// in f64 with Eigen's scalar evaluation order (-ffp-contract=off; sin / cos are the device libm, so results agree with
// the CPU to the last f32 bit except where a 1-ulp f64 difference crosses an f32 rounding boundary).
// Points the reference would not have pushed into the cloud (blind radius, point_filter_num) are written as NaN: the
// VoxelGrid kernels drop non-finite points, which is the same thing as their not being there.
// 20 B in (xyzi + stamp) and 16 B out per point: HBM-bound, ~4.5 MB per 120 k-point scan.
#include "lio_common.h"

namespace lio {

namespace {

constexpr int kThreads = 256;

__device__ inline void cross3d(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
// Eigen QuaternionBase::_transformVector
__device__ inline void qrot_d(const double q[4], const double v[3], double o[3]) {
    double uv[3], c2[3];
    cross3d(q, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    cross3d(q, uv, c2);
    for (int a = 0; a < 3; a++) o[a] = (v[a] + q[3] * uv[a]) + c2[a];
}

// sin(x) and 1 - cos(x).  Within a sweep |x| = |w| dt stays far below 1/2 rad: there both come from their Taylor series
// in Horner form (11 and 10 terms, truncation < 1e-20, evaluated with fma: a few f64 ulp at most), which is ~20 x cheaper
// than the library's general-range sincos; 1 - cos(x) is summed directly instead of cancelling.  Larger arguments take
// the library path.
__device__ inline void sin_versin(double x, double& s, double& v) {
    if (fabs(x) < 0.5) {
        const double z = x * x;
        double p = -1.0 / 51090942171709440000.0;         // -1/21!
        p = fma(p, z, 1.0 / 121645100408832000.0);         // 1/19!
        p = fma(p, z, -1.0 / 355687428096000.0);           // -1/17!
        p = fma(p, z, 1.0 / 1307674368000.0);              // 1/15!
        p = fma(p, z, -1.0 / 6227020800.0);                // -1/13!
        p = fma(p, z, 1.0 / 39916800.0);                   // 1/11!
        p = fma(p, z, -1.0 / 362880.0);                    // -1/9!
        p = fma(p, z, 1.0 / 5040.0);                       // 1/7!
        p = fma(p, z, -1.0 / 120.0);                       // -1/5!
        p = fma(p, z, 1.0 / 6.0);                          // 1/3!
        s = fma(-(x * z), p, x);                           // x - x^3 (1/3! - x^2/5! + ...)
        double q = 1.0 / 2432902008176640000.0;            // 1/20!
        q = fma(q, z, -1.0 / 6402373705728000.0);          // -1/18!
        q = fma(q, z, 1.0 / 20922789888000.0);             // 1/16!
        q = fma(q, z, -1.0 / 87178291200.0);               // -1/14!
        q = fma(q, z, 1.0 / 479001600.0);                  // 1/12!
        q = fma(q, z, -1.0 / 3628800.0);                   // -1/10!
        q = fma(q, z, 1.0 / 40320.0);                      // 1/8!
        q = fma(q, z, -1.0 / 720.0);                       // -1/6!
        q = fma(q, z, 1.0 / 24.0);                         // 1/4!
        q = fma(q, z, -0.5);                               // -1/2!
        v = -(z * q);                                      // 1 - cos x = x^2 (1/2! - x^2/4! + ...)
    } else {
        double c;
        sincos(x, &s, &c);
        v = 1.0 - c;
    }
}

__device__ inline void compensate(const UndistortArgs& A, const ImuPoseDev* poses, int h, double t, float& px, float& py, float& pz) {
    const ImuPoseDev& head = poses[h];
    const ImuPoseDev& tail = poses[h + 1];
    const double dt = t - head.off;
    // so3_math.h Exp(ang_vel, dt)
    double Rd[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double w0 = tail.gyr[0], w1 = tail.gyr[1], w2 = tail.gyr[2];
    const double n = sqrt(w0 * w0 + (w1 * w1 + w2 * w2));
    if (n > 0.0000001) {
        const double ax = w0 / n, ay = w1 / n, az = w2 / n;
        const double K[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
        const double th = n * dt;
        double sn, cs;
        sin_versin(th, sn, cs);
        double sK[9];
        for (int i = 0; i < 9; i++) sK[i] = cs * K[i];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                const double kk = sK[i * 3] * K[j] + (sK[i * 3 + 1] * K[3 + j] + sK[i * 3 + 2] * K[6 + j]);
                Rd[i * 3 + j] = (Rd[i * 3 + j] + sn * K[i * 3 + j]) + kk;
            }
    }
    double Ri[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Ri[i * 3 + j] = head.R[i * 3] * Rd[j] + (head.R[i * 3 + 1] * Rd[3 + j] + head.R[i * 3 + 2] * Rd[6 + j]);
    const double Pi[3] = {(double)px, (double)py, (double)pz};
    double T_ei[3], pl[3], pw[3], pe[3], pc[3];
    for (int a = 0; a < 3; a++) T_ei[a] = head.pos[a] + head.vel[a] * dt + 0.5 * tail.acc[a] * dt * dt - A.pos_e[a];
    qrot_d(A.ril, Pi, pl);
    for (int a = 0; a < 3; a++) pl[a] += A.til[a];
    for (int a = 0; a < 3; a++) pw[a] = (Ri[a * 3] * pl[0] + (Ri[a * 3 + 1] * pl[1] + Ri[a * 3 + 2] * pl[2])) + T_ei[a];
    const double rc[4] = {-A.rot_e[0], -A.rot_e[1], -A.rot_e[2], A.rot_e[3]};
    qrot_d(rc, pw, pe);
    for (int a = 0; a < 3; a++) pe[a] -= A.til[a];
    const double lc[4] = {-A.ril[0], -A.ril[1], -A.ril[2], A.ril[3]};
    qrot_d(lc, pe, pc);
    px = (float)pc[0]; py = (float)pc[1]; pz = (float)pc[2];
}

// the last head whose offset lies strictly before t; -1 when t <= poses[0].off (the point stays as it is)
__device__ inline int find_segment(const ImuPoseDev* poses, int n_poses, double t) {
    int h = -1;
    for (int k = n_poses - 2; k >= 0; k--)
        if (t > poses[k].off) { h = k; break; }
    return h;
}

__global__ __launch_bounds__(kThreads) void undistort_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ stamp_us, uint32_t n,
                                                             float4* __restrict__ out, const ImuPoseDev* __restrict__ g_poses, UndistortArgs A,
                                                             unsigned long long* __restrict__ block_min) {
    __shared__ ImuPoseDev poses[kMaxImuPoses];
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    // the point and its time stamp are requested before the pose table is staged (their addresses depend on i alone): one memory round trip for
    // the three instead of three in a row
    const uint32_t ic = i < n ? i : (n ? n - 1u : 0u);
    float4 p_in = in[ic];
    uint32_t stamp_i = A.undistort ? stamp_us[ic] : 0u;
    {
        const double* src = reinterpret_cast<const double*>(g_poses);
        double* dst = reinterpret_cast<double*>(poses);
        const int words = A.n_poses * (int)(sizeof(ImuPoseDev) / sizeof(double));
        for (int i = threadIdx.x; i < words; i += kThreads) dst[i] = src[i];
    }
    pin_loaded(p_in);
    pin_loaded(stamp_i);
    __syncthreads();
    unsigned long long key = ~0ull;  // (time bits, index) of a kept point: the minimum is the sorted cloud's begin()
    if (i < n) {
        float4 p = p_in;
        const float nanv = __int_as_float(0x7fc00000);
        bool keep = (A.filter_num <= 1) || (i % (uint32_t)A.filter_num == 0);
        keep = keep && ((double)(p.x * p.x + p.y * p.y + p.z * p.z) > A.blind2);
        if (!keep) {
            p = make_float4(nanv, nanv, nanv, p.w);
        } else if (A.undistort) {
            const float t_ms = (float)stamp_i / 1000.0f;  // added_pt.curvature = attr.stamp / 1000.0f (ms)
            const double t = (double)t_ms / double(1000);
            const int h = find_segment(poses, A.n_poses, t);
            if (h >= 0) compensate(A, poses, h, t, p.x, p.y, p.z);
            key = ((unsigned long long)__float_as_uint(t_ms) << 32) | i;
        }
        out[i] = p;
    }
    if (A.undistort) {  // workgroup minimum -> block_min[blockIdx.x]; no atomics (1 800 same-address atomics cost ~18 us)
        __shared__ unsigned long long wmin[kThreads / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t lo = __shfl_xor((uint32_t)key, off), hi = __shfl_xor((uint32_t)(key >> 32), off);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            key = o < key ? o : key;
        }
        if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = key;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long m = wmin[0];
            for (int w = 1; w < kThreads / 64; w++) m = wmin[w] < m ? wmin[w] : m;
            block_min[blockIdx.x] = m;
        }
    }
}

// The reference's backward walk parks its iterator on the earliest point (`if (it_pcl == begin) break`) and lets every
// earlier IMU segment compensate that one point AGAIN, on the already compensated coordinates (IMU_Processing.hpp:371-404).
// It only happens when that point's time is > 0; kept for parity.  One wave: it first folds the workgroup minima.
__global__ void __launch_bounds__(64) undistort_first_kernel(float4* out, const ImuPoseDev* __restrict__ poses, UndistortArgs A,
                                                             const unsigned long long* __restrict__ block_min, uint32_t n_blocks) {
    unsigned long long key = ~0ull;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 64) key = block_min[b] < key ? block_min[b] : key;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)key, off), hi = __shfl_xor((uint32_t)(key >> 32), off);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        key = o < key ? o : key;
    }
    if (key == ~0ull) return;
    const uint32_t i = (uint32_t)key;
    const float t_ms = __uint_as_float((uint32_t)(key >> 32));
    const double t = (double)t_ms / double(1000);
    // segment search across the lanes (<= 128 poses: two per lane), not a chain of dependent loads
    int h = -1;
    for (int k = (int)threadIdx.x; k < A.n_poses - 1; k += 64)
        if (t > poses[k].off) h = k;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h = max(h, __shfl_xor(h, off));
    if (h <= 0 || threadIdx.x != 0) return;
    float4 p = out[i];
    for (int k = h - 1; k >= 0; k--)
        if (t > poses[k].off) compensate(A, poses, k, t, p.x, p.y, p.z);
    out[i] = p;
}

// ---- localisation mode: constant-velocity compensation, undistortPoints(delta_pose, points, scan_period) of
// slam/common/slam_utils.cpp:163-191.  Everything is f32 as in the reference (Eigen Matrix<float,6,1> / AngleAxisf / Quaternionf /
// Affine3f, pcl::transformPoint); sums follow Eigen's fixed-size orders x0 + (x1 + x2) and (x0 + x1) + (x2 + x3).  sin / cos of the half angle are evaluated in
// f64 and rounded to f32, which is what a correctly rounded sinf / cosf returns except within ~1e-8 of a rounding boundary.
__device__ inline float4 delta_point(const float4 p, uint32_t stamp, const DeltaArgs& A) {
    const float ratio = (float)(((double)stamp / 1000000.0) / A.scan_period);  // float t_diff_ratio = (stamp / 1000000.0) / scan_period
    const float tx = ratio * A.t[0], ty = ratio * A.t[1], tz = ratio * A.t[2];
    const float wx = ratio * A.aa[0], wy = ratio * A.aa[1], wz = ratio * A.aa[2];
    const float norm = sqrtf(wx * wx + (wy * wy + wz * wz));
    float r00 = 1.f, r01 = 0.f, r02 = 0.f, r10 = 0.f, r11 = 1.f, r12 = 0.f, r20 = 0.f, r21 = 0.f, r22 = 1.f;
    if (!((double)norm < 1e-8)) {
        const float ax = wx / norm, ay = wy / norm, az = wz / norm;
        const float ha = 0.5f * norm;
        const float w = (float)cos((double)ha), sh = (float)sin((double)ha);
        const float x = sh * ax, y = sh * ay, z = sh * az;
        // QuaternionBase::toRotationMatrix
        const float t2x = 2.f * x, t2y = 2.f * y, t2z = 2.f * z;
        const float twx = t2x * w, twy = t2y * w, twz = t2z * w;
        const float txx = t2x * x, txy = t2y * x, txz = t2z * x;
        const float tyy = t2y * y, tyz = t2z * y, tzz = t2z * z;
        r00 = 1.f - (tyy + tzz); r01 = txy - twz; r02 = txz + twy;
        r10 = txy + twz; r11 = 1.f - (txx + tzz); r12 = tyz - twx;
        r20 = txz - twy; r21 = tyz + twx; r22 = 1.f - (txx + tyy);
    }
    // transform * v = [R t] [v; 1]: four terms in Eigen's fixed-size order (x0 + x1) + (x2 + x3)
    float4 o = p;
    o.x = (r00 * p.x + r01 * p.y) + (r02 * p.z + tx);
    o.y = (r10 * p.x + r11 * p.y) + (r12 * p.z + ty);
    o.z = (r20 * p.x + r21 * p.y) + (r22 * p.z + tz);
    return o;
}

__global__ __launch_bounds__(kThreads) void undistort_delta_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ stamp_us, uint32_t n,
                                                                   float4* __restrict__ out, DeltaArgs A) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    out[i] = delta_point(in[i], stamp_us[i], A);
}

// ---- the pose-list variant, undistortPoints(std::vector<PoseType>&, PointCloudAttrPtr&) of slam_utils.cpp:193-228 ---------------------
// The reference walks the cloud once: a point is compensated with the first pose interval, from the one the previous point used onwards,
// whose end (poses[i].timestamp - header.stamp, unsigned) is not before its stamp.  With interval ends that do not decrease that is
// seg(idx) = max over j <= idx of need(j), need(j) = first interval whose end is not before stamp j: a prefix maximum.
// Kernel 1: maxima of need() per tile of 2048 points.  Kernel 2: prefix over the tiles before, ordered in-tile prefix, transform.
constexpr int kPoseItems = 8;
constexpr int kPoseTile = kThreads * kPoseItems;

__device__ inline uint32_t pose_need(uint32_t stamp, const PoseListArgs& A) {
    uint32_t i = 1;
    while (i < (uint32_t)A.n_poses && (unsigned long long)stamp > A.limit[i]) i++;
    return i;  // == n_poses: past the last pose
}

__global__ __launch_bounds__(kThreads) void undistort_poses_need_kernel(const uint32_t* __restrict__ stamp_us, uint32_t n, PoseListArgs A,
                                                                        uint32_t* __restrict__ tile_max) {
    const uint32_t base = blockIdx.x * kPoseTile;
    uint32_t mx = 0;
#pragma unroll
    for (int r = 0; r < kPoseItems; r++) {
        const uint32_t i = base + r * kThreads + threadIdx.x;
        if (i < n) mx = max(mx, pose_need(stamp_us[i], A));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
    __shared__ uint32_t w[kThreads / 64];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) tile_max[blockIdx.x] = max(max(w[0], w[1]), max(w[2], w[3]));
}

__global__ __launch_bounds__(kThreads) void undistort_poses_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ stamp_us, uint32_t n,
                                                                   float4* __restrict__ out, PoseListArgs A, const uint32_t* __restrict__ tile_max) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // segment reached by the tiles before this one
    uint32_t pre = 0;
    for (uint32_t b = tid; b < blockIdx.x; b += kThreads) pre = max(pre, tile_max[b]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pre = max(pre, __shfl_xor(pre, off));
    __shared__ uint32_t wred[kThreads / 64], wtot[kThreads / 64];
    if (lane == 0) wred[wave] = pre;
    __syncthreads();
    pre = max(max(wred[0], wred[1]), max(wred[2], wred[3]));
    // thread t owns kPoseItems CONSECUTIVE points, so that (thread, item) order is the cloud's order
    const uint32_t base = blockIdx.x * kPoseTile + tid * kPoseItems;
    uint32_t need[kPoseItems], stamp[kPoseItems], mine = 0;
#pragma unroll
    for (int r = 0; r < kPoseItems; r++) {
        const uint32_t i = base + r;
        stamp[r] = i < n ? stamp_us[i] : 0u;
        need[r] = i < n ? pose_need(stamp[r], A) : 0u;
        mine = max(mine, need[r]);
    }
    // exclusive prefix maximum over the threads of the tile: inclusive wave scan, shifted, plus the waves before
    uint32_t inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc = max(inc, t);
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t run = __shfl_up(inc, 1);
    if (lane == 0) run = 0;
    for (int w = 0; w < wave; w++) run = max(run, wtot[w]);
    run = max(run, pre);
#pragma unroll
    for (int r = 0; r < kPoseItems; r++) {
        const uint32_t i = base + r;
        if (i >= n) break;
        run = max(run, need[r]);
        float4 p = in[i];
        if (run < (uint32_t)A.n_poses) p = delta_point(p, stamp[r], A.d[run]);
        out[i] = p;
    }
}

}  // namespace

int undistort_delta_launch(hipStream_t stream, const float4* d_in, const uint32_t* d_stamp_us, uint32_t n, float4* d_out, const DeltaArgs& args) {
    if (n == 0) return LIO_OK;
    hipLaunchKernelGGL(undistort_delta_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, d_in, d_stamp_us, n, d_out, args);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

int undistort_poses_launch(hipStream_t stream, const float4* d_in, const uint32_t* d_stamp_us, uint32_t n, float4* d_out, const PoseListArgs& args,
                           uint32_t* d_tile_max /* ceil(n / 2048) words */) {
    if (n == 0) return LIO_OK;
    const uint32_t tiles = (n + kPoseTile - 1) / kPoseTile;
    hipLaunchKernelGGL(undistort_poses_need_kernel, dim3(tiles), dim3(kThreads), 0, stream, d_stamp_us, n, args, d_tile_max);
    hipLaunchKernelGGL(undistort_poses_kernel, dim3(tiles), dim3(kThreads), 0, stream, d_in, d_stamp_us, n, d_out, args, d_tile_max);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

int undistort_launch(hipStream_t stream, const float4* d_in, const uint32_t* d_stamp_us, uint32_t n, float4* d_out, const ImuPoseDev* d_poses,
                     const UndistortArgs& args, unsigned long long* d_block_min) {
    if (n == 0) return LIO_OK;
    if (args.undistort && (args.n_poses < 2 || args.n_poses > kMaxImuPoses)) {
        set_error("undistort: %d IMU poses (2..%d supported)", args.n_poses, kMaxImuPoses);
        return LIO_E_CAPACITY;
    }
    const uint32_t blocks = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(undistort_kernel, dim3(blocks), dim3(kThreads), 0, stream, d_in, d_stamp_us, n, d_out, d_poses, args, d_block_min);
    if (args.undistort) hipLaunchKernelGGL(undistort_first_kernel, dim3(1), dim3(64), 0, stream, d_out, d_poses, args, d_block_min, blocks);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

}  // namespace lio
