// gicp.hip -- Generalized-ICP on the device: the fine matcher of the reference's map merge / loop closure and the CPU fallback
// matcher family of its localisation.
//
// Replaces fast_gicp::FastGICP<PointXYZI, PointXYZI> as select_registration_method("FAST_GICP") configures it
// (/root/reference/slam/backend/hdl_graph_slam/src/hdl_graph_slam/registrations.cpp:33-42: 20 neighbours, transformation epsilon 0.01,
// 64 iterations, max correspondence distance 2.0; overlap_merge.hpp:56-58 tightens it to 0.5 / 0.001 for the fine alignment), i.e.
//   calculate_covariances   fast_gicp_impl.hpp:244-303   per point: the k nearest neighbours (kd-tree there), their covariance, PLANE
//                                                        regularisation (SVD, singular values replaced by (1, 1, 1e-3))
//   update_correspondences  fast_gicp_impl.hpp:118-157   per source point: nearest target point of the transformed point within the
//                                                        correspondence distance, Mahalanobis matrix (C_B + R C_A R^T)^-1
//   linearize / compute_error  :159-242                  e = b - T a, J = [ [T a]x | -I ], H = sum J^T M J, b = sum J^T M e, E = sum e^T M e
//   LsqRegistration         lsq_registration_impl.hpp:71-208   the LM loop on SE(3) shared with the NDT matcher (lsq.h)
// Mapping to the machine: both clouds live in a hash grid of `grid_resolution` cells (the map's brick-coherent table, the points of a cell
// contiguous in the pool) -- the exact k-NN and 1-NN searches walk rings of cells around the query's cell and stop once the k-th / best
// distance is within the radius already covered (after ring r every unseen point is at least r cells away), so they return what an exact
// kd-tree returns.  One lane per point; the k-NN heap of a lane sits in LDS ([slot][lane]: conflict free).  All sums in f64, folded in a
// fixed order (run-to-run identical).  The PLANE regularisation needs only the eigenvector of the smallest eigenvalue:
// U diag(1, 1, 1e-3) V^T = I - (1 - 1e-3) v0 v0^T for a symmetric positive semi-definite covariance.
#include <sched.h>

#include <vector>

#include "hashgrid.h"
#include "lio_common.h"
#include "lsq.h"

namespace lio {

constexpr int kGicpThreads = 128;
constexpr int kGicpMaxK = 32;
constexpr int kGicpAcc = 29;  // 21 H (upper), 6 b, err, count

struct GicpXform {
    double R[9], t[3];   // trans (double)
    float Rf[9], tf[3];  // trans.cast<float>()
};

__device__ inline bool grid_find_slot(const Slot* __restrict__ table, uint32_t mask, int cx, int cy, int cz, uint32_t& ptr, uint32_t& cnt, uint32_t& slot) {
    const unsigned long long want = pack_key(cx, cy, cz);
    BrickProbe bp = brick_probe(cx, cy, cz);
    for (uint32_t probe = 0; probe <= (mask >> 6); probe++) {
        const uint32_t h = brick_slot(bp, mask);
        const Slot sl = table[h];
        if (sl.key == want) { ptr = sl.ptr; cnt = sl.cnt; slot = h; return cnt > 0; }
        if (sl.key == kEmptyKey) return false;
        brick_next(bp);
    }
    return false;
}

// k nearest neighbours of every point of the cloud within the cloud itself (the point is its own nearest), their covariance, PLANE
// regularisation.  cov6 = (xx, xy, xz, yy, yz, zz) of the regularised matrix, pool order.
// The sorted list of the k <= 32 best lives in two registers per lane (position = lane, 16 + lane); an insertion is one shift by a lane.
__global__ void __launch_bounds__(kGrpThreads) gicp_cov_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool, uint32_t n,
                                                               float res, int k, double* __restrict__ cov6) {
    const int lane = threadIdx.x & (kGrp - 1);
    const uint32_t i = (blockIdx.x * kGrpThreads + threadIdx.x) / kGrp;
    if (i >= n) return;  // whole groups leave together
    const float4 p = pool[i];
    int kx, ky, kz;
    pos2grid_ndt(p.x, p.y, p.z, res, kx, ky, kz);
    const float gap = cell_gap3(p.x, p.y, p.z, res, kx, ky, kz);
    const unsigned long long kNone = ~0ull;
    unsigned long long e0 = kNone, e1 = kNone, kth = kNone;
    for (int r = 0;; r++) {
        const int n_cells = shell_cells(r);
        for (int t0 = 0; t0 < n_cells; t0 += kGrp) {
            const int t = t0 + lane;
            uint32_t ptr = 0, cnt = 0;
            bool found = false;
            if (t < n_cells) {
                int dx, dy, dz;
                shell_cell(r, t, dx, dy, dz);
                found = grid_find(table, mask, kx + dx, ky + dy, kz + dz, ptr, cnt);
            }
            uint32_t hits = grp_ballot(found);
            while (hits) {
                const int b = __ffs((int)hits) - 1;
                hits &= hits - 1;
                const uint32_t cptr = __shfl(ptr, b, kGrp), ccnt = __shfl(cnt, b, kGrp);
                for (uint32_t j0 = 0; j0 < ccnt; j0 += kGrp) {
                    const uint32_t j = j0 + lane;
                    unsigned long long c = kNone;
                    if (j < ccnt) {
                        const float4 q = pool[cptr + j];
                        const float ex = q.x - p.x, ey = q.y - p.y, ez = q.z - p.z;
                        const float d2 = (ex * ex + ey * ey) + ez * ez;
                        if (d2 == d2) c = cand_key(d2, cptr + j);
                    }
                    uint32_t take = grp_ballot(c < kth);
                    while (take) {
                        const int bb = __ffs((int)take) - 1;
                        take &= take - 1;
                        const unsigned long long cc = __shfl(c, bb, kGrp);
                        unsigned long long l0 = __shfl_up(e0, 1, kGrp), l1 = __shfl_up(e1, 1, kGrp);
                        const unsigned long long carry = __shfl(e0, kGrp - 1, kGrp);
                        if (lane == 0) { l0 = 0; l1 = carry; }
                        e0 = e0 <= cc ? e0 : (l0 <= cc ? cc : l0);
                        e1 = e1 <= cc ? e1 : (l1 <= cc ? cc : l1);
                    }
                    kth = k <= kGrp ? __shfl(e0, k - 1, kGrp) : __shfl(e1, k - 1 - kGrp, kGrp);
                }
            }
        }
        const float reach = (float)r * res + gap;
        if ((kth != kNone && __uint_as_float((uint32_t)(kth >> 32)) <= reach * reach) || r > 64) break;
    }
    // neighbors.colwise() -= neighbors.rowwise().mean(); cov = neighbors * neighbors^T / k   (f64, as the reference casts); the lanes hold
    // one or two neighbours each, sums by a fixed butterfly over the group
    const bool v0 = lane < k && e0 != kNone, v1 = kGrp + lane < k && e1 != kNone;
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0;
    if (v0) q0 = pool[(uint32_t)e0];
    if (v1) q1 = pool[(uint32_t)e1];
    double m[3] = {(double)q0.x + (double)q1.x, (double)q0.y + (double)q1.y, (double)q0.z + (double)q1.z};
#pragma unroll
    for (int off = kGrp / 2; off > 0; off >>= 1)
        for (int a = 0; a < 3; a++) m[a] += __shfl_xor(m[a], off, kGrp);
    for (int a = 0; a < 3; a++) m[a] /= (double)k;
    double C6[6] = {0, 0, 0, 0, 0, 0};
    if (v0) {
        const double d[3] = {(double)q0.x - m[0], (double)q0.y - m[1], (double)q0.z - m[2]};
        C6[0] += d[0] * d[0]; C6[1] += d[0] * d[1]; C6[2] += d[0] * d[2]; C6[3] += d[1] * d[1]; C6[4] += d[1] * d[2]; C6[5] += d[2] * d[2];
    }
    if (v1) {
        const double d[3] = {(double)q1.x - m[0], (double)q1.y - m[1], (double)q1.z - m[2]};
        C6[0] += d[0] * d[0]; C6[1] += d[0] * d[1]; C6[2] += d[0] * d[2]; C6[3] += d[1] * d[1]; C6[4] += d[1] * d[2]; C6[5] += d[2] * d[2];
    }
#pragma unroll
    for (int off = kGrp / 2; off > 0; off >>= 1)
        for (int a = 0; a < 6; a++) C6[a] += __shfl_xor(C6[a], off, kGrp);
    if (lane != 0) return;
    double C[9] = {C6[0], C6[1], C6[2], C6[1], C6[3], C6[4], C6[2], C6[4], C6[5]};
    for (int a = 0; a < 9; a++) C[a] /= (double)k;
    double w[3], V[9];
    ek_eig3_sym(C, w, V);  // ascending: column 0 = the direction of least spread (the surface normal)
    const double n0[3] = {V[0], V[3], V[6]};
    const double g = 1.0 - 1e-3;
    double* o = cov6 + (size_t)i * 6;
    o[0] = 1.0 - g * n0[0] * n0[0];
    o[1] = -g * n0[0] * n0[1];
    o[2] = -g * n0[0] * n0[2];
    o[3] = 1.0 - g * n0[1] * n0[1];
    o[4] = -g * n0[1] * n0[2];
    o[5] = 1.0 - g * n0[2] * n0[2];
}

// update_correspondences: nearest target point of trans_f * a within the correspondence distance, and the Mahalanobis matrix of the pair
__global__ void __launch_bounds__(kGrpThreads) gicp_corr_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ tpool, float res,
                                                                const double* __restrict__ tcov, const float4* __restrict__ spool, const double* __restrict__ scov,
                                                                uint32_t n_src, GicpXform X, float max_d2, int32_t* __restrict__ corr, double* __restrict__ maha) {
    const int lane = threadIdx.x & (kGrp - 1);
    const uint32_t i = (blockIdx.x * kGrpThreads + threadIdx.x) / kGrp;
    if (i >= n_src) return;
    const float4 a = spool[i];
    // trans_f * [x y z 1]: Eigen folds the four products of a row pairwise, (r0 x + r1 y) + (r2 z + t) (checked against the reference build)
    const float tx = (X.Rf[0] * a.x + X.Rf[1] * a.y) + (X.Rf[2] * a.z + X.tf[0]);
    const float ty = (X.Rf[3] * a.x + X.Rf[4] * a.y) + (X.Rf[5] * a.z + X.tf[1]);
    const float tz = (X.Rf[6] * a.x + X.Rf[7] * a.y) + (X.Rf[8] * a.z + X.tf[2]);
    int kx, ky, kz;
    pos2grid_ndt(tx, ty, tz, res, kx, ky, kz);
    const float gap = cell_gap3(tx, ty, tz, res, kx, ky, kz);
    unsigned long long bk = ~0ull;  // (squared distance, index) of the nearest so far, the same in all lanes of the group after every ring
    for (int r = 0;; r++) {
        const int n_cells = shell_cells(r);
        for (int t0 = 0; t0 < n_cells; t0 += kGrp) {
            const int t = t0 + lane;
            uint32_t ptr = 0, cnt = 0;
            bool found = false;
            if (t < n_cells) {
                int dx, dy, dz;
                shell_cell(r, t, dx, dy, dz);
                found = grid_find(table, mask, kx + dx, ky + dy, kz + dz, ptr, cnt);
            }
            if (found)
                for (uint32_t j = 0; j < cnt; j++) {
                    const float4 q = tpool[ptr + j];
                    const float ex = q.x - tx, ey = q.y - ty, ez = q.z - tz;
                    const float d2 = (ex * ex + ey * ey) + ez * ez;
                    if (d2 == d2) {
                        const unsigned long long c = cand_key(d2, ptr + j);
                        if (c < bk) bk = c;
                    }
                }
        }
#pragma unroll
        for (int off = kGrp / 2; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(bk, off, kGrp);
            if (o < bk) bk = o;
        }
        const float reach = (float)r * res + gap;
        if ((bk != ~0ull && __uint_as_float((uint32_t)(bk >> 32)) <= reach * reach) || reach * reach > max_d2) break;
    }
    if (lane != 0) return;
    const float best = bk != ~0ull ? __uint_as_float((uint32_t)(bk >> 32)) : INFINITY;
    const uint32_t bi = bk != ~0ull ? (uint32_t)bk : 0xFFFFFFFFu;
    const bool ok = bi != 0xFFFFFFFFu && best < max_d2;
    corr[i] = ok ? (int32_t)bi : -1;
    if (!ok) return;
    // RCR = cov_B + R cov_A R^T ; mahalanobis = RCR^-1 (3 x 3: the fourth row / column of the reference's 4 x 4 is the unit one)
    const double* ca = scov + (size_t)i * 6;
    const double* cb = tcov + (size_t)bi * 6;
    const double A[9] = {ca[0], ca[1], ca[2], ca[1], ca[3], ca[4], ca[2], ca[4], ca[5]};
    double RA[9], M[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { double s = 0; for (int k2 = 0; k2 < 3; k2++) s += X.R[r * 3 + k2] * A[k2 * 3 + c]; RA[r * 3 + c] = s; }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { double s = 0; for (int k2 = 0; k2 < 3; k2++) s += RA[r * 3 + k2] * X.R[c * 3 + k2]; M[r * 3 + c] = s; }
    M[0] += cb[0]; M[1] += cb[1]; M[2] += cb[2]; M[3] += cb[1]; M[4] += cb[3]; M[5] += cb[4]; M[6] += cb[2]; M[7] += cb[4]; M[8] += cb[5];
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
    const double id = 1.0 / det;
    double* o = maha + (size_t)i * 6;  // symmetric inverse
    o[0] = c00 * id;
    o[1] = (M[2] * M[7] - M[1] * M[8]) * id;
    o[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    o[3] = (M[0] * M[8] - M[2] * M[6]) * id;
    o[4] = (M[2] * M[3] - M[0] * M[5]) * id;
    o[5] = (M[0] * M[4] - M[1] * M[3]) * id;
}

// linearize (DERIV) / compute_error on the cached correspondences and Mahalanobis matrices at transform X
template <bool DERIV>
__global__ void __launch_bounds__(kGicpThreads) gicp_cost_kernel(const float4* __restrict__ tpool, const float4* __restrict__ spool, uint32_t n_src,
                                                                 const int32_t* __restrict__ corr, const double* __restrict__ maha, GicpXform X,
                                                                 double* __restrict__ partial) {
    const uint32_t i = blockIdx.x * kGicpThreads + threadIdx.x;
    double acc[kGicpAcc];
#pragma unroll
    for (int a = 0; a < kGicpAcc; a++) acc[a] = 0.0;
    if (i < n_src && corr[i] >= 0) {
        const float4 a = spool[i];
        const float4 b = tpool[corr[i]];
        const double ax = (double)a.x, ay = (double)a.y, az = (double)a.z;
        const double ta[3] = {(X.R[0] * ax + X.R[1] * ay) + (X.R[2] * az + X.t[0]), (X.R[3] * ax + X.R[4] * ay) + (X.R[5] * az + X.t[1]),
                              (X.R[6] * ax + X.R[7] * ay) + (X.R[8] * az + X.t[2])};
        const double e[3] = {(double)b.x - ta[0], (double)b.y - ta[1], (double)b.z - ta[2]};
        const double* m = maha + (size_t)i * 6;
        const double M[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]};
        double Me[3];
        for (int r = 0; r < 3; r++) Me[r] = M[r * 3] * e[0] + M[r * 3 + 1] * e[1] + M[r * 3 + 2] * e[2];
        acc[27] = e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
        acc[28] = 1.0;
        if (DERIV) {
            // J (3 x 6) = [ skew(T a) | -I ]
            double J[18] = {0, -ta[2], ta[1], -1, 0, 0, ta[2], 0, -ta[0], 0, -1, 0, -ta[1], ta[0], 0, 0, 0, -1};
            double MJ[18];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 6; c++) MJ[r * 6 + c] = M[r * 3] * J[c] + M[r * 3 + 1] * J[6 + c] + M[r * 3 + 2] * J[12 + c];
            int t = 0;
            for (int r = 0; r < 6; r++)
                for (int c = r; c < 6; c++) acc[t++] = J[r] * MJ[c] + J[6 + r] * MJ[6 + c] + J[12 + r] * MJ[12 + c];
            for (int r = 0; r < 6; r++) acc[21 + r] = J[r] * Me[0] + J[6 + r] * Me[1] + J[12 + r] * Me[2];
        }
    }
    __shared__ double red[kGicpThreads / 64][kGicpAcc];
#pragma unroll
    for (int a = 0; a < kGicpAcc; a++) {
        double v = acc[a];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][a] = v;
    }
    __syncthreads();
    if (threadIdx.x < kGicpAcc) {
        double v = 0.0;
        for (int w2 = 0; w2 < kGicpThreads / 64; w2++) v += red[w2][threadIdx.x];
        partial[(size_t)blockIdx.x * kGicpAcc + threadIdx.x] = v;
    }
}


// ---- the voxelised variant: fast_gicp::FastVGICP (fast_vgicp_impl.hpp:72-204, fast_vgicp_voxel.hpp:125-182) ----------------------------
// Target = Gaussian voxels of `voxel_resolution` (key floor(x / res - 0.5) in f64): mean of the points' positions and mean of their
// (regularised, 20-NN) covariances, ADDITIVE mode; a source point corresponds to the voxel(s) its transformed position falls in (DIRECT1:
// that voxel; DIRECT7 / 27: its neighbours too), weight sqrt(points in the voxel), Mahalanobis matrix (C_voxel + R C_A R^T)^-1.
struct __attribute__((aligned(16))) VgicpVoxel {
    double mean[3];
    double cov[6];
    double n;
};
struct VgicpOffsets {
    int n;
    int off[27][3];
};

// stamp: the voxel map's copy of the target carries the point's index in the k-NN grid's pool order (where its covariance lies)
__global__ void __launch_bounds__(256) vgicp_stamp_kernel(const float4* __restrict__ in, float4* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    out[i] = make_float4(p.x, p.y, p.z, __uint_as_float(i));
}

// AdditiveGaussianVoxel::append / finalize: one lane per occupied voxel, its points in pool order
__global__ void __launch_bounds__(256) vgicp_fold_kernel(const Slot* __restrict__ table, uint32_t table_cap, const float4* __restrict__ pool,
                                                         const double* __restrict__ tcov, VgicpVoxel* __restrict__ vox) {
    const uint32_t h = blockIdx.x * 256u + threadIdx.x;
    if (h >= table_cap) return;
    const Slot s = table[h];
    if (s.key == kEmptyKey || s.cnt == 0) return;
    double m[3] = {0, 0, 0}, c[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j = 0; j < s.cnt; j++) {
        const float4 p = pool[s.ptr + j];
        m[0] += (double)p.x; m[1] += (double)p.y; m[2] += (double)p.z;
        const double* pc = tcov + (size_t)__float_as_uint(p.w) * 6;
#pragma unroll
        for (int a = 0; a < 6; a++) c[a] += pc[a];
    }
    const double n = (double)s.cnt;
    VgicpVoxel v;
    for (int a = 0; a < 3; a++) v.mean[a] = m[a] / n;
    for (int a = 0; a < 6; a++) v.cov[a] = c[a] / n;
    v.n = n;
    vox[h] = v;
}

__device__ inline void gicp_transform_d(const GicpXform& X, const float4 a, double ta[3]) {
    const double ax = (double)a.x, ay = (double)a.y, az = (double)a.z;
    ta[0] = (X.R[0] * ax + X.R[1] * ay) + (X.R[2] * az + X.t[0]);
    ta[1] = (X.R[3] * ax + X.R[4] * ay) + (X.R[5] * az + X.t[1]);
    ta[2] = (X.R[6] * ax + X.R[7] * ay) + (X.R[8] * az + X.t[2]);
}
__device__ inline void gicp_mahalanobis(const double* __restrict__ ca, const double* __restrict__ cb, const GicpXform& X, double* __restrict__ o) {
    const double A[9] = {ca[0], ca[1], ca[2], ca[1], ca[3], ca[4], ca[2], ca[4], ca[5]};
    double RA[9], M[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { double s = 0; for (int k2 = 0; k2 < 3; k2++) s += X.R[r * 3 + k2] * A[k2 * 3 + c]; RA[r * 3 + c] = s; }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { double s = 0; for (int k2 = 0; k2 < 3; k2++) s += RA[r * 3 + k2] * X.R[c * 3 + k2]; M[r * 3 + c] = s; }
    M[0] += cb[0]; M[1] += cb[1]; M[2] += cb[2]; M[3] += cb[1]; M[4] += cb[3]; M[5] += cb[4]; M[6] += cb[2]; M[7] += cb[4]; M[8] += cb[5];
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id;
    o[1] = (M[2] * M[7] - M[1] * M[8]) * id;
    o[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    o[3] = (M[0] * M[8] - M[2] * M[6]) * id;
    o[4] = (M[2] * M[3] - M[0] * M[5]) * id;
    o[5] = (M[0] * M[4] - M[1] * M[3]) * id;
}

// update_correspondences (:72-116): one lane per (source point, neighbour offset)
__global__ void __launch_bounds__(kGicpThreads) vgicp_corr_kernel(const Slot* __restrict__ vtable, uint32_t vmask, const VgicpVoxel* __restrict__ vox, double vres,
                                                                  const float4* __restrict__ spool, const double* __restrict__ scov, uint32_t n_src,
                                                                  VgicpOffsets offs, GicpXform X, int32_t* __restrict__ corr, double* __restrict__ maha) {
    const uint32_t e = blockIdx.x * kGicpThreads + threadIdx.x;
    if (e >= n_src * (uint32_t)offs.n) return;
    const uint32_t i = e / (uint32_t)offs.n, o = e % (uint32_t)offs.n;
    double ta[3];
    gicp_transform_d(X, spool[i], ta);
    const int cx = (int)floor(ta[0] / vres - 0.5) + offs.off[o][0], cy = (int)floor(ta[1] / vres - 0.5) + offs.off[o][1],
              cz = (int)floor(ta[2] / vres - 0.5) + offs.off[o][2];
    uint32_t ptr, cnt, slot;
    if (!grid_find_slot(vtable, vmask, cx, cy, cz, ptr, cnt, slot)) { corr[e] = -1; return; }
    corr[e] = (int32_t)slot;
    gicp_mahalanobis(scov + (size_t)i * 6, vox[slot].cov, X, maha + (size_t)e * 6);
}

// linearize (:118-180) / compute_error (:182-204) on the cached voxel correspondences
template <bool DERIV>
__global__ void __launch_bounds__(kGicpThreads) vgicp_cost_kernel(const VgicpVoxel* __restrict__ vox, const float4* __restrict__ spool, uint32_t n_src, int n_off,
                                                                  const int32_t* __restrict__ corr, const double* __restrict__ maha, GicpXform X,
                                                                  double* __restrict__ partial) {
    const uint32_t e = blockIdx.x * kGicpThreads + threadIdx.x;
    double acc[kGicpAcc];
#pragma unroll
    for (int a = 0; a < kGicpAcc; a++) acc[a] = 0.0;
    if (e < n_src * (uint32_t)n_off && corr[e] >= 0) {
        const VgicpVoxel v = vox[corr[e]];
        double ta[3];
        gicp_transform_d(X, spool[e / (uint32_t)n_off], ta);
        const double er[3] = {v.mean[0] - ta[0], v.mean[1] - ta[1], v.mean[2] - ta[2]};
        const double* m = maha + (size_t)e * 6;
        const double M[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]};
        const double w = sqrt(v.n);
        double Me[3];
        for (int r = 0; r < 3; r++) Me[r] = M[r * 3] * er[0] + M[r * 3 + 1] * er[1] + M[r * 3 + 2] * er[2];
        acc[27] = w * (er[0] * Me[0] + er[1] * Me[1] + er[2] * Me[2]);
        acc[28] = 1.0;
        if (DERIV) {
            double J[18] = {0, -ta[2], ta[1], -1, 0, 0, ta[2], 0, -ta[0], 0, -1, 0, -ta[1], ta[0], 0, 0, 0, -1};
            double MJ[18];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 6; c++) MJ[r * 6 + c] = M[r * 3] * J[c] + M[r * 3 + 1] * J[6 + c] + M[r * 3 + 2] * J[12 + c];
            int t = 0;
            for (int r = 0; r < 6; r++)
                for (int c = r; c < 6; c++) acc[t++] = w * (J[r] * MJ[c] + J[6 + r] * MJ[6 + c] + J[12 + r] * MJ[12 + c]);
            for (int r = 0; r < 6; r++) acc[21 + r] = w * (J[r] * Me[0] + J[6 + r] * Me[1] + J[12 + r] * Me[2]);
        }
    }
    __shared__ double red[kGicpThreads / 64][kGicpAcc];
#pragma unroll
    for (int a = 0; a < kGicpAcc; a++) {
        double v = acc[a];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][a] = v;
    }
    __syncthreads();
    if (threadIdx.x < kGicpAcc) {
        double v = 0.0;
        for (int w2 = 0; w2 < kGicpThreads / 64; w2++) v += red[w2][threadIdx.x];
        partial[(size_t)blockIdx.x * kGicpAcc + threadIdx.x] = v;
    }
}

struct GicpReport {
    double acc[kGicpAcc];
    uint32_t seq, pad;
};

__global__ void __launch_bounds__(1024) gicp_report_kernel(const double* __restrict__ partial, uint32_t nb, GicpReport* __restrict__ out, uint32_t seq) {
    __shared__ double acc[kGicpAcc];
    const int tid = threadIdx.x, c = tid >> 5, l = tid & 31;
    double s = 0.0;
    if (c < kGicpAcc)
        for (uint32_t b = l; b < nb; b += 32) s += partial[(size_t)b * kGicpAcc + c];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (c < kGicpAcc && l == 0) acc[c] = s;
    __syncthreads();
    if (tid < kGicpAcc) out->acc[tid] = acc[tid];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) *reinterpret_cast<volatile uint32_t*>(&out->seq) = seq;
}

}  // namespace lio

using namespace lio;

struct lio_gicp {
    int device = 0;
    float res = 1.0f;
    int k = 20;
    uint32_t max_points = 0;
    lio_map* grid[2] = {nullptr, nullptr};  // 0 target, 1 source: hash grids holding the clouds
    uint32_t n[2] = {0, 0};
    double* cov[2] = {nullptr, nullptr};
    int32_t* corr = nullptr;
    double* maha = nullptr;
    double* partial = nullptr;
    GicpReport* report = nullptr;
    GicpReport* report_dev = nullptr;
    uint32_t seq = 0;
    float4* stage = nullptr;
    // voxelised variant (FastVGICP): off while voxel_res == 0
    double voxel_res = 0.0;
    VgicpOffsets offs;
    lio_map* vmap = nullptr;       // the target's Gaussian voxels: hash grid keyed as fast_vgicp_voxel.hpp does
    VgicpVoxel* vvox = nullptr;    // one record per table slot
    bool vmap_valid = false;
    int32_t* vcorr = nullptr;      // [n_src x offsets]
    double* vmaha = nullptr;
    double* vpartial = nullptr;
    uint32_t vblocks = 0;
};

namespace {

GicpXform to_gx(const double T[16]) {
    GicpXform x;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) { x.R[i * 3 + j] = T[i * 4 + j]; x.Rf[i * 3 + j] = (float)T[i * 4 + j]; }
        x.t[i] = T[i * 4 + 3];
        x.tf[i] = (float)T[i * 4 + 3];
    }
    return x;
}

int gicp_set_cloud(lio_gicp* g, int which, const float* xyzi, uint32_t n) {
    if (!g || (!xyzi && n)) return LIO_E_INVALID;
    if (n > g->max_points) { set_error("lio_gicp: cloud of %u points exceeds max_points %u", n, g->max_points); return LIO_E_CAPACITY; }
    if ((int)n < g->k) { set_error("lio_gicp: a cloud needs at least k = %d points", g->k); return LIO_E_INVALID; }
    hipSetDevice(g->device);
    // an EMPTY grid per cloud: the first batch into an empty map is laid out exactly (cell by cell, contiguous from the start of the pool).
    // The grid object is made once and emptied in place afterwards (map_clear: memsets; a fresh lio_map_create per call was 5-65 ms of
    // hipMalloc / hipFree, more than the covariances)
    if (!g->grid[which]) {
        g->grid[which] = map_create_mode(g->device, g->res, g->max_points, g->max_points, 1, 1);
        if (!g->grid[which]) return LIO_E_DEVICE;
    } else {
        const int rc0 = map_clear(g->grid[which]);
        if (rc0 != LIO_OK) return rc0;
    }
    lio_map* m = g->grid[which];
    hipStream_t st = m->stream;
    LIO_HIP_TRY(hipMemcpyAsync(g->stage, xyzi, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, st));
    int rc = lio_map_insert_device(m, g->stage, n, 0.0);
    if (rc != LIO_OK) return rc;
    g->n[which] = n;
    if (which == 0) g->vmap_valid = false;
    hipLaunchKernelGGL(gicp_cov_kernel, (uint32_t)(((uint64_t)n * kGrp + kGrpThreads - 1) / kGrpThreads), kGrpThreads, 0, st, m->table, m->table_mask, m->pool, n, g->res, g->k, g->cov[which]);
    LIO_HIP_TRY(hipGetLastError());
    LIO_HIP_TRY(hipStreamSynchronize(st));
    return LIO_OK;
}

// create_voxelmap (fast_vgicp_voxel.hpp:129-162) of the current target: built on first use after the target or the mode changed
int vgicp_build(lio_gicp* g) {
    lio_map* mt = g->grid[0];
    hipStream_t st = mt->stream;
    const uint32_t n = g->n[0];
    if (g->vmap && g->vmap->res != (float)g->voxel_res) {  // another voxel size: another grid
        lio_map_destroy(g->vmap); g->vmap = nullptr;
        hipFree(g->vvox); g->vvox = nullptr;
    }
    if (!g->vmap) {
        g->vmap = map_create_mode(g->device, (float)g->voxel_res, g->max_points, g->max_points, 1, 2);
        if (!g->vmap) return LIO_E_DEVICE;
        LIO_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->vvox), (size_t)g->vmap->table_cap * sizeof(VgicpVoxel)));
    } else {
        const int rc0 = map_clear(g->vmap);
        if (rc0 != LIO_OK) return rc0;
    }
    hipLaunchKernelGGL(vgicp_stamp_kernel, (n + 255) / 256, 256, 0, st, mt->pool, g->stage, n);
    LIO_HIP_TRY(hipStreamSynchronize(st));
    const int rc = lio_map_insert_device(g->vmap, g->stage, n, 0.0);
    if (rc != LIO_OK) return rc;
    hipLaunchKernelGGL(vgicp_fold_kernel, (g->vmap->table_cap + 255) / 256, 256, 0, g->vmap->stream, g->vmap->table, g->vmap->table_cap, g->vmap->pool, g->cov[0],
                       g->vvox);
    LIO_HIP_TRY(hipGetLastError());
    LIO_HIP_TRY(hipStreamSynchronize(g->vmap->stream));
    g->vmap_valid = true;
    return LIO_OK;
}

int gicp_eval(lio_gicp* g, const double T[16], double max_corr_dist, bool update, bool deriv, double* H, double* b, double* err, uint32_t* n_corr) {
    if (!g->grid[0] || !g->grid[1]) { set_error("lio_gicp: set target and source first"); return LIO_E_STATE; }
    hipStream_t st = g->grid[0]->stream;
    const GicpXform X = to_gx(T);
    const uint32_t ns = g->n[1];
    uint32_t blocks = (ns + kGicpThreads - 1) / kGicpThreads;
    lio_map* mt = g->grid[0];
    const double* partial = g->partial;
    if (g->voxel_res > 0.0) {  // FastVGICP::linearize / compute_error
        if (!g->vmap_valid) {
            const int rc = vgicp_build(g);
            if (rc != LIO_OK) return rc;
        }
        const uint32_t ne = ns * (uint32_t)g->offs.n;
        blocks = (ne + kGicpThreads - 1) / kGicpThreads;
        const double vres = (double)(float)g->voxel_res;
        if (update)
            hipLaunchKernelGGL(vgicp_corr_kernel, blocks, kGicpThreads, 0, st, g->vmap->table, g->vmap->table_mask, g->vvox, vres, g->grid[1]->pool, g->cov[1], ns, g->offs, X,
                               g->vcorr, g->vmaha);
        if (deriv) hipLaunchKernelGGL(vgicp_cost_kernel<true>, blocks, kGicpThreads, 0, st, g->vvox, g->grid[1]->pool, ns, g->offs.n, g->vcorr, g->vmaha, X, g->vpartial);
        else hipLaunchKernelGGL(vgicp_cost_kernel<false>, blocks, kGicpThreads, 0, st, g->vvox, g->grid[1]->pool, ns, g->offs.n, g->vcorr, g->vmaha, X, g->vpartial);
        partial = g->vpartial;
    } else {
    if (update) {
        const double d2 = max_corr_dist * max_corr_dist;
        hipLaunchKernelGGL(gicp_corr_kernel, (uint32_t)(((uint64_t)ns * kGrp + kGrpThreads - 1) / kGrpThreads), kGrpThreads, 0, st, mt->table, mt->table_mask, mt->pool, g->res, g->cov[0], g->grid[1]->pool, g->cov[1], ns, X,
                           d2 > 3.0e38 ? 3.0e38f : (float)d2, g->corr, g->maha);
    }
    if (deriv) hipLaunchKernelGGL(gicp_cost_kernel<true>, blocks, kGicpThreads, 0, st, mt->pool, g->grid[1]->pool, ns, g->corr, g->maha, X, g->partial);
    else hipLaunchKernelGGL(gicp_cost_kernel<false>, blocks, kGicpThreads, 0, st, mt->pool, g->grid[1]->pool, ns, g->corr, g->maha, X, g->partial);
    }
    const uint32_t seq = ++g->seq;
    hipLaunchKernelGGL(gicp_report_kernel, 1, 1024, 0, st, partial, blocks, g->report_dev, seq);
    LIO_HIP_TRY(hipGetLastError());
    volatile uint32_t* ps = &g->report->seq;
    for (uint64_t spin = 0; *ps != seq; spin++) {
        __builtin_ia32_pause();
        if (spin > 4000 && (spin & 63) == 0) sched_yield();
        if (spin > 200000000ull) {
            LIO_HIP_TRY(hipStreamSynchronize(st));
            if (*ps != seq) { set_error("gicp: the cost kernel did not report"); return LIO_E_DEVICE; }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const double* a = g->report->acc;
    if (H) {
        int t = 0;
        for (int r = 0; r < 6; r++)
            for (int c = r; c < 6; c++) { H[r * 6 + c] = a[t]; H[c * 6 + r] = a[t]; t++; }
    }
    if (b) for (int r = 0; r < 6; r++) b[r] = a[21 + r];
    if (err) *err = a[27];
    if (n_corr) *n_corr = (uint32_t)(a[28] + 0.5);
    return LIO_OK;
}

}  // namespace

extern "C" {

lio_gicp* lio_gicp_create(int device, float grid_resolution, uint32_t max_points, int k_correspondences) {
    if (!(grid_resolution > 0.f) || max_points == 0 || k_correspondences < 3 || k_correspondences > kGicpMaxK) { set_error("lio_gicp_create: bad argument"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_error("lio_gicp_create: no HIP device %d (this library has no CPU fallback)", device); return nullptr; }
    lio_gicp* g = new lio_gicp();
    g->device = device;
    g->res = grid_resolution;
    g->k = k_correspondences;
    g->max_points = max_points;
    const uint32_t blocks = (max_points + kGicpThreads - 1) / kGicpThreads;
    bool ok = hipMalloc(reinterpret_cast<void**>(&g->cov[0]), (size_t)max_points * 6 * sizeof(double)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&g->cov[1]), (size_t)max_points * 6 * sizeof(double)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&g->corr), (size_t)max_points * sizeof(int32_t)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&g->maha), (size_t)max_points * 6 * sizeof(double)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&g->partial), (size_t)blocks * kGicpAcc * sizeof(double)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&g->stage), (size_t)max_points * sizeof(float4)) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&g->report), sizeof(GicpReport), hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer(reinterpret_cast<void**>(&g->report_dev), g->report, 0) == hipSuccess;
    if (!ok) { set_error("lio_gicp_create: allocation failed"); lio_gicp_destroy(g); return nullptr; }
    memset(g->report, 0, sizeof(GicpReport));
    return g;
}

void lio_gicp_destroy(lio_gicp* g) {
    if (!g) return;
    hipSetDevice(g->device);
    for (int w = 0; w < 2; w++) {
        if (g->grid[w]) lio_map_destroy(g->grid[w]);
        if (g->cov[w]) hipFree(g->cov[w]);
    }
    if (g->vmap) lio_map_destroy(g->vmap);
    if (g->vvox) hipFree(g->vvox);
    if (g->vcorr) hipFree(g->vcorr);
    if (g->vmaha) hipFree(g->vmaha);
    if (g->vpartial) hipFree(g->vpartial);
    if (g->corr) hipFree(g->corr);
    if (g->maha) hipFree(g->maha);
    if (g->partial) hipFree(g->partial);
    if (g->stage) hipFree(g->stage);
    if (g->report) hipHostFree(g->report);
    delete g;
}

int lio_gicp_set_target(lio_gicp* g, const float* xyzi, uint32_t n) { return gicp_set_cloud(g, 0, xyzi, n); }
int lio_gicp_set_source(lio_gicp* g, const float* xyzi, uint32_t n) { return gicp_set_cloud(g, 1, xyzi, n); }

int lio_gicp_set_voxel_mode(lio_gicp* g, double voxel_resolution, int search_method) {
    if (!g || voxel_resolution < 0.0 || (search_method != 1 && search_method != 7 && search_method != 27)) { set_error("lio_gicp_set_voxel_mode: bad argument"); return LIO_E_INVALID; }
    hipSetDevice(g->device);
    if (g->vcorr) { hipFree(g->vcorr); g->vcorr = nullptr; }
    if (g->vmaha) { hipFree(g->vmaha); g->vmaha = nullptr; }
    if (g->vpartial) { hipFree(g->vpartial); g->vpartial = nullptr; }
    g->voxel_res = voxel_resolution;
    g->vmap_valid = false;
    if (voxel_resolution == 0.0) return LIO_OK;
    VgicpOffsets& o = g->offs;  // neighbor_offsets(search_method), fast_vgicp_voxel.hpp:10-43, same order
    o.n = 0;
    auto push = [&](int x, int y, int z) { o.off[o.n][0] = x; o.off[o.n][1] = y; o.off[o.n][2] = z; o.n++; };
    if (search_method == 1) push(0, 0, 0);
    else if (search_method == 7) { push(0, 0, 0); push(1, 0, 0); push(-1, 0, 0); push(0, 1, 0); push(0, -1, 0); push(0, 0, 1); push(0, 0, -1); }
    else
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) push(i - 1, j - 1, k - 1);
    const size_t ne = (size_t)g->max_points * o.n;
    g->vblocks = (uint32_t)((ne + kGicpThreads - 1) / kGicpThreads);
    if (hipMalloc(reinterpret_cast<void**>(&g->vcorr), ne * sizeof(int32_t)) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&g->vmaha), ne * 6 * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&g->vpartial), (size_t)g->vblocks * kGicpAcc * sizeof(double)) != hipSuccess) {
        set_error("lio_gicp_set_voxel_mode: allocation failed");
        return LIO_E_DEVICE;
    }
    return LIO_OK;
}

// the Gaussian voxel of the target that holds point p: (points, mean, regularised covariance); 0 = no such voxel
int lio_gicp_voxel_at(lio_gicp* g, const float p[3], double mean[3], double cov6[6]) {
    if (!g || !p || g->voxel_res <= 0.0 || !g->grid[0]) return LIO_E_INVALID;
    hipSetDevice(g->device);
    if (!g->vmap_valid) { const int rc = vgicp_build(g); if (rc != LIO_OK) return rc; }
    const double res = (double)(float)g->voxel_res;
    const int cx = (int)floor((double)p[0] / res - 0.5), cy = (int)floor((double)p[1] / res - 0.5), cz = (int)floor((double)p[2] / res - 0.5);
    // walk the probe sequence on the host over a copy of the (small) table
    std::vector<Slot> tab(g->vmap->table_cap);
    LIO_HIP_TRY(hipMemcpy(tab.data(), g->vmap->table, tab.size() * sizeof(Slot), hipMemcpyDeviceToHost));
    const unsigned long long want = pack_key(cx, cy, cz);
    for (uint32_t h = 0; h < g->vmap->table_cap; h++) {
        if (tab[h].key != want || tab[h].cnt == 0) continue;
        VgicpVoxel v;
        LIO_HIP_TRY(hipMemcpy(&v, g->vvox + h, sizeof(v), hipMemcpyDeviceToHost));
        if (mean) for (int a = 0; a < 3; a++) mean[a] = v.mean[a];
        if (cov6) for (int a = 0; a < 6; a++) cov6[a] = v.cov[a];
        return (int)v.n;
    }
    return 0;
}

int lio_gicp_download(lio_gicp* g, int which, float* xyzi, double* cov6, uint32_t cap) {
    if (!g || which < 0 || which > 1 || !g->grid[which]) return LIO_E_INVALID;
    const uint32_t n = g->n[which];
    if (n > cap) return LIO_E_CAPACITY;
    hipSetDevice(g->device);
    if (xyzi) LIO_HIP_TRY(hipMemcpy(xyzi, g->grid[which]->pool, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    if (cov6) LIO_HIP_TRY(hipMemcpy(cov6, g->cov[which], (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost));
    return (int)n;
}

int lio_gicp_correspondences(lio_gicp* g, int32_t* corr, uint32_t cap) {
    if (!g || !corr || !g->grid[1]) return LIO_E_INVALID;
    if (g->n[1] > cap) return LIO_E_CAPACITY;
    hipSetDevice(g->device);
    LIO_HIP_TRY(hipMemcpy(corr, g->corr, (size_t)g->n[1] * sizeof(int32_t), hipMemcpyDeviceToHost));
    return (int)g->n[1];
}

int lio_gicp_linearize(lio_gicp* g, const double T[16], double max_corr_dist, int update_corr, int with_derivatives, double H[36], double b[6], double* err,
                       uint32_t* n_corr) {
    if (!g || !T) return LIO_E_INVALID;
    hipSetDevice(g->device);
    return gicp_eval(g, T, max_corr_dist, update_corr != 0, with_derivatives != 0, H, b, err, n_corr);
}

int lio_gicp_align(lio_gicp* g, const double guess[16], const lio_ndt_params* prm, double max_corr_dist, double out[16], int* iterations, int* converged) {
    if (!g || !guess || !out) return LIO_E_INVALID;
    hipSetDevice(g->device);
    lio_ndt_params p;
    if (prm) p = *prm;
    else {  // select_registration_method("FAST_GICP") (registrations.cpp:33-42) over LsqRegistration's defaults (lsq_registration_impl.hpp:20-35)
        lio_ndt_default_params(&p);
        p.rotation_epsilon_deg = 1e-2;
        p.transformation_epsilon = 0.01;
        p.max_iterations = 64;
        p.max_process_time_ms = -1;
    }
    auto lin = [&](const double x[16], double H[36], double b[6], double* y) { return gicp_eval(g, x, max_corr_dist, true, true, H, b, y, nullptr); };
    auto er = [&](const double*, const double x[16], double* y) { return gicp_eval(g, x, max_corr_dist, false, false, nullptr, nullptr, y, nullptr); };
    return lsq_align(p, guess, lin, er, out, iterations, converged);
}

}  // extern "C"
