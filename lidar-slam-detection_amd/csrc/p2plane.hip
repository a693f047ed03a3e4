// p2plane.hip -- point-to-plane measurement model of the FastLIO frontend on gfx950.
//
// Replaces the body of h_share_model_geometric
// (/root/reference/slam/mapping/fastlio/src/laserMapping.cpp:813-982) and map_incremental's per-point
// decision (:523-563):
//   linearize_kernel  one lane per downsampled point: body->world (f64), esti_plane on the five cached
//                     neighbours (include/common_lib.h:236-268: 5x3 column-pivoted Householder QR in f32),
//                     residual gate (:860-871), Jacobian block [n, (R_il p + t_il) x (R_wi^T n)] (:909-931)
//                     and a fixed-order f64 block reduction of J^T J (21), J^T h (6), sum|r|, count.
//                     The N_eff x 15 matrix h_x of the reference never materialises.
//   finalize_kernel   one workgroup folds the partials in a fixed order and writes the 29-number record straight into
//                     mapped host memory; the host spins on its sequence word (the 3x3 eigen-decomposition of
//                     sum n n^T is done there).
//   degeneracy_kernel the six degeneracy sums of :946-964; launched only when the eigenvalue bound
//                     lambda_i - 0.1736^2 N_eff >= 250 does not already decide the test (see capi.hip).
//   classify_kernel   map_incremental's need_add test + ballot/prefix-sum compaction of the points to insert.
// Per-point f32 arithmetic is written as explicit sequential IEEE operations (compiled with
// -ffp-contract=off) in the operation order of Eigen's ColPivHouseholderQR (scalar build: unrolled redux trees for the
// fixed-size norms, column-oriented triangular solve) so that gates flip exactly where the reference's own
// esti_plane does (pinned through oracle/_ref, see tests/test_oracle_vs_ref.py).
#include "knn_dev.h"
#include "lio_common.h"

namespace lio {


__device__ inline void body_to_world_d(const PoseArgs& P, const float4 pb, double pi[3], float4& pw) {
    const double vx = (double)pb.x, vy = (double)pb.y, vz = (double)pb.z;
    double ux = P.ql[1] * vz - P.ql[2] * vy, uy = P.ql[2] * vx - P.ql[0] * vz, uz = P.ql[0] * vy - P.ql[1] * vx;
    ux += ux; uy += uy; uz += uz;
    double cx = P.ql[1] * uz - P.ql[2] * uy, cy = P.ql[2] * ux - P.ql[0] * uz, cz = P.ql[0] * uy - P.ql[1] * ux;
    pi[0] = ((vx + P.ql[3] * ux) + cx) + P.tl[0];
    pi[1] = ((vy + P.ql[3] * uy) + cy) + P.tl[1];
    pi[2] = ((vz + P.ql[3] * uz) + cz) + P.tl[2];
    ux = P.qw[1] * pi[2] - P.qw[2] * pi[1]; uy = P.qw[2] * pi[0] - P.qw[0] * pi[2]; uz = P.qw[0] * pi[1] - P.qw[1] * pi[0];
    ux += ux; uy += uy; uz += uz;
    cx = P.qw[1] * uz - P.qw[2] * uy; cy = P.qw[2] * ux - P.qw[0] * uz; cz = P.qw[0] * uy - P.qw[1] * ux;
    pw.x = (float)(((pi[0] + P.qw[3] * ux) + cx) + P.tw[0]);
    pw.y = (float)(((pi[1] + P.qw[3] * uy) + cy) + P.tw[1]);
    pw.z = (float)(((pi[2] + P.qw[3] * uz) + cz) + P.tw[2]);
    pw.w = pb.w;
}

// common_lib.h:236-268.  A n = -1 over the five neighbours, n by ColPivHouseholderQR::solve, then
// pabcd = (n / |n|, 1 / |n|); false if any |n.q + d| > threshold.  All indices are compile-time after
// unrolling so the 5x3 system lives in registers.
__device__ inline bool esti_plane_dev(const float4 pt[5], float threshold, float pabcd[4]) {
    float A[5][3], b[5];
#pragma unroll
    for (int j = 0; j < 5; j++) { A[j][0] = pt[j].x; A[j][1] = pt[j].y; A[j][2] = pt[j].z; b[j] = -1.0f; }
    const float eps = 1.1920929e-07f;
    float normUpd[3], normDir[3], hcoef[3];
    int perm[3] = {0, 1, 2};
    float maxnorm = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        // m_qr.col(k).norm(): fixed size 5 -> Eigen's completely unrolled redux is the tree (x0 + x1) + (x2 + (x3 + x4))
        const float s = (A[0][k] * A[0][k] + A[1][k] * A[1][k]) + (A[2][k] * A[2][k] + (A[3][k] * A[3][k] + A[4][k] * A[4][k]));
        normDir[k] = sqrtf(s);
        normUpd[k] = normDir[k];
        maxnorm = fmaxf(maxnorm, normDir[k]);
    }
    const float thr_helper = ((maxnorm * eps) * (maxnorm * eps)) / 5.0f;
    const float downdate_thr = sqrtf(eps);
    int nonzero = 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int big = k;
        float bigv = normUpd[k];
#pragma unroll
        for (int j = k + 1; j < 3; j++)
            if (normUpd[j] > bigv) { bigv = normUpd[j]; big = j; }
        const float big_sq = bigv * bigv;
        if (nonzero == 3 && big_sq < thr_helper * (float)(5 - k)) nonzero = k;
#pragma unroll
        for (int j = k + 1; j < 3; j++) {
            if (big == j) {
#pragma unroll
                for (int r = 0; r < 5; r++) { const float t = A[r][k]; A[r][k] = A[r][j]; A[r][j] = t; }
                { const float t = normUpd[k]; normUpd[k] = normUpd[j]; normUpd[j] = t; }
                { const float t = normDir[k]; normDir[k] = normDir[j]; normDir[j] = t; }
                { const int t = perm[k]; perm[k] = perm[j]; perm[j] = t; }
            }
        }
        const float c0 = A[k][k];
        float tail = 0.f;
#pragma unroll
        for (int r = k + 1; r < 5; r++) tail = tail + A[r][k] * A[r][k];
        float tau, beta;
        if (tail <= 1.17549435e-38f) {
            tau = 0.f;
            beta = c0;
#pragma unroll
            for (int r = k + 1; r < 5; r++) A[r][k] = 0.f;
        } else {
            beta = sqrtf(c0 * c0 + tail);
            if (c0 >= 0.f) beta = -beta;
            const float den = c0 - beta;
#pragma unroll
            for (int r = k + 1; r < 5; r++) A[r][k] = A[r][k] / den;
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
        hcoef[k] = tau;
        if (tau != 0.f) {
#pragma unroll
            for (int j = k + 1; j < 3; j++) {
                float t = 0.f;
#pragma unroll
                for (int r = k + 1; r < 5; r++) t = t + A[r][k] * A[r][j];
                t = t + A[k][j];
                A[k][j] = A[k][j] - tau * t;
#pragma unroll
                for (int r = k + 1; r < 5; r++) A[r][j] = A[r][j] - (tau * A[r][k]) * t;
            }
        }
#pragma unroll
        for (int j = k + 1; j < 3; j++) {
            if (normUpd[j] != 0.f) {
                float t = fabsf(A[k][j]) / normUpd[j];
                t = (1.f + t) * (1.f - t);
                if (t < 0.f) t = 0.f;
                const float ratio = normUpd[j] / normDir[j];
                const float t2 = t * (ratio * ratio);
                if (t2 <= downdate_thr) {
                    float s = 0.f;
#pragma unroll
                    for (int r = k + 1; r < 5; r++) s = s + A[r][j] * A[r][j];
                    normDir[j] = sqrtf(s);
                    normUpd[j] = normDir[j];
                } else {
                    normUpd[j] = normUpd[j] * sqrtf(t);
                }
            }
        }
    }
    // c = Q^T b
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k < nonzero) {
            const float tau = hcoef[k];
            if (tau != 0.f) {
                float t = 0.f;
#pragma unroll
                for (int r = k + 1; r < 5; r++) t = t + A[r][k] * b[r];
                t = t + b[k];
                b[k] = b[k] - tau * t;
#pragma unroll
                for (int r = k + 1; r < 5; r++) b[r] = b[r] - (tau * A[r][k]) * t;
            }
        }
    }
    // triangularView<Upper>().solveInPlace(): Eigen's column-major vector solve is column oriented
    // (x_i = b_i / a_ii, then b_r -= x_i * a_ri for the rows above)
    float xs[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 2; i >= 0; i--) {
        if (i < nonzero) {
            xs[i] = b[i] / A[i][i];
#pragma unroll
            for (int r = 0; r < 3; r++)
                if (r < i) b[r] = b[r] - xs[i] * A[r][i];
        }
    }
    float nv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (i < nonzero) {
#pragma unroll
            for (int a = 0; a < 3; a++)
                if (perm[i] == a) nv[a] = xs[i];
        }
    }
    const float n = sqrtf(nv[0] * nv[0] + (nv[1] * nv[1] + nv[2] * nv[2]));  // normvec.norm(): fixed size 3 -> tree x0 + (x1 + x2)
    pabcd[0] = nv[0] / n;
    pabcd[1] = nv[1] / n;
    pabcd[2] = nv[2] / n;
    pabcd[3] = 1.0f / n;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const float v = ((pabcd[0] * pt[j].x + pabcd[1] * pt[j].y) + pabcd[2] * pt[j].z) + pabcd[3];
        if (fabsf(v) > threshold) ok = false;
    }
    return ok;
}

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ void linearize_body(const PoseArgs& pose, int redo_knn, const ScanDev* __restrict__ sd,
                                               const float4* __restrict__ ds_body, float4* __restrict__ ds_world,
                                               const float4* __restrict__ nn_pts, uint32_t nn_stride,
                                               const int32_t* __restrict__ nn_cnt, uint8_t* __restrict__ selected,
                                               float4* __restrict__ normvec, double* __restrict__ partial, uint32_t n_known = 0xFFFFFFFFu,
                                               uint32_t vb = 0xFFFFFFFFu) {
    // vb: the block of kLinThreads points this call works on (the batched kernel strides over a slot's blocks: its grid is sized for the largest cloud
    // a slot may hold -- 1 563 blocks at max_ds 100 000 -- and a 12 000-point scan left 1 380 of them per slot and launch to start, wait for the
    // slot's filter words and exit: 100 000 one-wave workgroups per launch of 64 scans, most of the kernel's 39 us); default: the launch's own block
    if (vb == 0xFFFFFFFFu) vb = blockIdx.x;
    const uint32_t n = n_known != 0xFFFFFFFFu ? n_known : sd->n_ds;  // (the batched kernels have read it with the slot's filter words)
    if (vb * kLinThreads >= n) return;  // finalize only reads the first ceil(n / kLinThreads) partials
    const uint32_t i = vb * kLinThreads + threadIdx.x;
    // the point's addends are formed at reduction time from its Jacobian row, residual and |residual| (8 doubles live instead of 29)
    double row[6] = {0, 0, 0, 0, 0, 0};
    double h = 0.0, ares = 0.0, one = 0.0;
    // Everything this point needs from memory has an address that depends on i alone: requested up front, together (clamped, unconditional; which
    // set is wanted depends on redo_knn, uniform over the workgroup).  Written as it reads -- point, then its neighbour count inside `if (i < n)`,
    // then the cached plane or the five neighbours inside `if (sel)` -- the kernel made three to four memory round trips one after the other.
    const uint32_t ic = i < n ? i : n - 1u;
    float4 pb = ds_body[ic];
    float* plane_d = reinterpret_cast<float*>(normvec + nn_stride);
    int32_t cnt_i = 0;
    uint32_t sel_i = 0;
    float4 nv_c = make_float4(0.f, 0.f, 0.f, 0.f);
    float pd_c = 0.f;
    float4 near[5];
    if (redo_knn) {
        cnt_i = nn_cnt[ic];
#pragma unroll
        for (int k = 0; k < 5; k++) near[k] = nn_pts[(size_t)k * nn_stride + ic];
#pragma unroll
        for (int k = 0; k < 5; k++) pin_loaded(near[k]);
    } else {
        sel_i = selected[ic];
        nv_c = normvec[ic];
        pd_c = plane_d[ic];
        pin_loaded(nv_c);
        pin_loaded(pd_c);
    }
    pin_loaded(pb);
    if (i < n) {
        double pi[3];
        float4 pw;
        body_to_world_d(pose, pb, pi, pw);
        ds_world[i] = pw;
        // point_selected_surf is re-armed only by a neighbour search (laserMapping.cpp:842-854)
        bool sel = redo_knn ? (cnt_i >= 5) : (sel_i != 0);
        if (sel) {
            // The plane is a function of the five neighbours alone: a pass that does not search (ES:1646-1650 leaves `converge` false) would
            // fit the plane it fitted last pass -- same inputs, same bits.  It is kept from the pass that searched: (a, b, c) in normvec,
            // d behind it (plane_d); a point is only evaluated again while it stayed selected, and it stays selected only through passes
            // that wrote both.  Skips the 80-byte neighbour load and the QR (most of this kernel's instructions) in 2.6 of 4.6 passes.
            // (An all-zero normal = nothing cached: a bare lio_p2plane_linearize(redo_knn = 0) on a fresh scan takes the long way.)
            float pabcd[4];
            bool have = false;
            if (!redo_knn) {
                if (nv_c.x != 0.f || nv_c.y != 0.f || nv_c.z != 0.f) {
                    pabcd[0] = nv_c.x; pabcd[1] = nv_c.y; pabcd[2] = nv_c.z; pabcd[3] = pd_c;
                    have = true;
                }
            }
            if (!have) {
                if (!redo_knn) {  // (rare: nothing cached)
#pragma unroll
                    for (int k = 0; k < 5; k++) near[k] = nn_pts[(size_t)k * nn_stride + i];
                }
                have = esti_plane_dev(near, 0.1f, pabcd);
                if (have) plane_d[i] = pabcd[3];
            }
            sel = false;
            if (have) {
                const float pd2 = ((pabcd[0] * pw.x + pabcd[1] * pw.y) + pabcd[2] * pw.z) + pabcd[3];
                const double pbn = sqrt((double)pb.x * pb.x + ((double)pb.y * pb.y + (double)pb.z * pb.z));  // V3D::norm(): Eigen tree x0 + (x1 + x2)
                // float s = 1 - 0.9 * fabs(pd2) / sqrt(p_body.norm()); if (s > 0.9)   (laserMapping.cpp:861-863)
                const float sc = (float)(1 - 0.9 * fabs((double)pd2) / sqrt(pbn));
                if ((double)sc > 0.9) {
                    sel = true;
                    normvec[i] = make_float4(pabcd[0], pabcd[1], pabcd[2], pd2);
                    // Jacobian block (laserMapping.cpp:909-931), f64
                    const double nx = (double)pabcd[0], ny = (double)pabcd[1], nz = (double)pabcd[2];
                    // C = rot.conjugate() * norm_vec
                    const double qx = -pose.qw[0], qy = -pose.qw[1], qz = -pose.qw[2], qw = pose.qw[3];
                    double ux = qy * nz - qz * ny, uy = qz * nx - qx * nz, uz = qx * ny - qy * nx;
                    ux += ux; uy += uy; uz += uz;
                    const double cx = (nx + qw * ux) + (qy * uz - qz * uy);
                    const double cy = (ny + qw * uy) + (qz * ux - qx * uz);
                    const double cz = (nz + qw * uz) + (qx * uy - qy * ux);
                    row[0] = nx; row[1] = ny; row[2] = nz;
                    row[3] = pi[1] * cz - pi[2] * cy;  // A = point_crossmat * C = p_imu x C
                    row[4] = pi[2] * cx - pi[0] * cz;
                    row[5] = pi[0] * cy - pi[1] * cx;
                    h = -(double)pd2;
                    ares = (double)fabsf(pd2);
                    one = 1.0;
                }
            }
        }
        selected[i] = sel ? 1 : 0;
    }
    // Workgroup reduction, fixed order (run-to-run identical):
    //   1. quads: two DPP quad-permute butterflies, (v0 + v1) + (v2 + v3) in every lane of the quad -- no LDS;
    //   2. one lane per quad parks the 29 quad sums in LDS, transposed [component][quad]: 29 x 16 doubles = 3.6 KB per (one-wave) workgroup
    //      (the full [29][64] form was 14.5 KB: ten waves per CU by LDS, half of what the registers allow);
    //   3. 29 lanes each add the quad sums of one component in quad order.
    constexpr int kQuads = kLinThreads / 4;
    static_assert(kLinThreads >= kAcc && kQuads % 2 == 0, "one summing lane per component");
    __shared__ double red[kAcc][kQuads];
    {
        int t = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = a; c < 6; c++) {
                double v = row[a] * row[c];
                v += dpp_f64<0xB1>(v);  // quad_perm [1, 0, 3, 2]
                v += dpp_f64<0x4E>(v);  // quad_perm [2, 3, 0, 1]
                if ((threadIdx.x & 3) == 0) red[t][threadIdx.x >> 2] = v;
                t++;
            }
#pragma unroll
        for (int a = 0; a < 8; a++) {
            double v = a < 6 ? row[a] * h : (a == 6 ? ares : one);
            v += dpp_f64<0xB1>(v);
            v += dpp_f64<0x4E>(v);
            if ((threadIdx.x & 3) == 0) red[21 + a][threadIdx.x >> 2] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < kAcc) {
        const double2* row = reinterpret_cast<const double2*>(&red[threadIdx.x][0]);
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < kQuads / 2; k++) {
            const double2 v = row[k];
            s += v.x;
            s += v.y;
        }
        partial[(size_t)vb * kAcc + threadIdx.x] = s;
    }
}

__global__ void __launch_bounds__(kLinThreads) linearize_kernel(PoseArgs pose, int redo_knn, const ScanDev* __restrict__ sd,
                                                                const float4* __restrict__ ds_body, float4* __restrict__ ds_world,
                                                                const float4* __restrict__ nn_pts, uint32_t nn_stride,
                                                                const int32_t* __restrict__ nn_cnt, uint8_t* __restrict__ selected,
                                                                float4* __restrict__ normvec, double* __restrict__ partial) {
    linearize_body(pose, redo_knn, sd, ds_body, ds_world, nn_pts, nn_stride, nn_cnt, selected, normvec, partial);
}
// the scans of a batch: pose and the "redo the neighbour search" flag come from the slot's device-resident filter
// workgroups per slot of linearize_batch: 256 x 64 = 16 384 points per sweep of the grid (a 0.5 m-leaf scan is 7 000 - 18 000 points)
#ifndef LIO_LIN_GRID
#define LIO_LIN_GRID 256
#endif
constexpr uint32_t kLinGridCap = LIO_LIN_GRID;
__global__ void __launch_bounds__(kLinThreads) linearize_batch(const SlotDesc* __restrict__ slots) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;
    const SlotGate sg = slot_gate(d);
    if ((sg.status != EK_RUNNING) | (sg.n_ds < d.min_ds)) return;
    const PoseArgs& pose = sg.pose;
    for (uint32_t vb = blockIdx.x; vb * kLinThreads < sg.n_ds; vb += gridDim.x) {
        linearize_body(pose, sg.converge, d.sd, d.ds_body, d.ds_world, d.nn_pts, d.max_ds, d.nn_cnt, d.selected, d.normvec, d.partial, sg.n_ds, vb);
        __syncthreads();  // (the reduction's LDS staging is reused by this workgroup's next block)
    }
}

// component `comp` of the workgroups' partial records b = l, l + 32, l + 64, ... < nb, added in that order -- with the loads of eight steps in flight
// (the plain loop `s += partial[b * kAcc + comp]` is one memory round trip per step: six of them for a scan of 12 000 points, the larger
// part of the filter-pass kernel's "copy-in + fold" phase).  Same additions, same order: the same bits.
__device__ __forceinline__ double fold_partials(const double* __restrict__ partial, uint32_t l, uint32_t nb, int comp) {
    double s = 0.0;
    uint32_t b = l;
    for (; b + 7u * 32u < nb; b += 8u * 32u) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = partial[(size_t)(b + 32u * k) * kAcc + comp];
#pragma unroll
        for (int k = 0; k < 8; k++) s += v[k];
    }
    if (b < nb) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t bb = b + 32u * k;
            v[k] = partial[(size_t)(bb < nb ? bb : b) * kAcc + comp];
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (b + 32u * k < nb) s += v[k];
    }
    return s;
}

// One workgroup folds the per-workgroup partials in a fixed order and writes the 29-number record straight into
// mapped pinned host memory; the host spins on the record's sequence word (no copy launch, no stream-sync call).
// (Folding inside linearize_kernel by the last workgroup to arrive -- agent-scope ticket, write-through payload --
// was measured: 23 us against 6.8 + 6.1 us for the two launches, because one wave then does the whole fold and
// the PCIe stores; kept as two launches.)
constexpr int kFinThreads = 1024;

__global__ void __launch_bounds__(kFinThreads) finalize_kernel(ScanDev* __restrict__ sd, const double* __restrict__ partial,
                                                               const MapDev* __restrict__ md, lio_normal_eq* __restrict__ out) {
    __shared__ double acc[kAcc];
    const int tid = threadIdx.x;
    const uint32_t n = sd->n_ds;
    const uint32_t nb = (n + kLinThreads - 1) / kLinThreads;
    // fixed-order reduction of the block partials: component c is owned by 32 lanes (stride-32 chunks,
    // then a fixed xor tree) -> run-to-run identical
    {
        const int c = tid >> 5, l = tid & 31;
        double s = 0.0;
        if (c < kAcc) s = fold_partials(partial, (uint32_t)l, nb, c);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (c < kAcc && l == 0) acc[c] = s;
    }
    __syncthreads();
    // the record goes out through 64 lanes at once (a handful of PCIe writes instead of ~60 dependent-looking single stores), then
    // one system-scope fence and the sequence word
    if (tid < 64) {
        // acc[] holds the upper triangle row by row: index of (a, c), a <= c
        if (tid < 36) {
            int a = tid / 6, c = tid % 6;
            if (a > c) { const int u = a; a = c; c = u; }
            const int t = a * 6 - a * (a - 1) / 2 + (c - a);
            out->JtJ[tid] = acc[t];
        } else if (tid < 42) {
            out->Jtr[tid - 36] = acc[21 + (tid - 36)];
        } else if (tid < 51) {
            const int k = tid - 42;
            int a = k / 3, c = k % 3;
            if (a > c) { const int u = a; a = c; c = u; }
            out->nnT[k] = acc[a * 6 - a * (a - 1) / 2 + (c - a)];
        } else if (tid < 54) {
            out->contri[tid - 51] = 0.0;
        } else if (tid < 57) {
            out->strong[tid - 54] = 0.0;
        } else if (tid == 57) {
            out->sum_abs_res = acc[27];
            out->n_eff = (uint32_t)(acc[28] + 0.5);
            out->n_ds = n;
            out->n_tie = sd->n_tie;  // hand the tie queue to the host and re-arm it
            sd->n_tie_done = sd->n_tie;
            sd->n_tie = 0;
        } else if (tid == 58) {
            unsigned long long kc = 0ull;
            if (md)
                for (int k = 0; k < 64; k++) kc += md->knn_cand[k * 16];
            out->n_knn_candidates_lo = (uint32_t)kc;
            out->n_knn_candidates_hi = (uint32_t)(kc >> 32);
        }
        __threadfence_system();
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t seq = sd->seq + 1u;
        sd->seq = seq;
        *reinterpret_cast<volatile uint32_t*>(&out->seq) = seq;
    }
}

// degeneracy sums (laserMapping.cpp:946-964): rows re-normalised, |cos| against each eigenvector of sum n n^T.
// Every addend is a float in (0.1736, 1] widened to double, so sums of up to 2^17 of them are exact in f64
// and the f64 atomics below are order-independent: the result is run-to-run identical.
__global__ void __launch_bounds__(256) degeneracy_kernel(const ScanDev* __restrict__ sd, const uint8_t* __restrict__ selected,
                                                         const float4* __restrict__ normvec, lio_normal_eq* __restrict__ out) {
    const uint32_t n = sd->n_ds;
    if (blockIdx.x * 256u >= n) return;
    __shared__ double red[4][6];
    __shared__ double V[9];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < 9) V[tid] = out->eigvec[tid];
    __syncthreads();
    double s[6] = {0, 0, 0, 0, 0, 0};
    const uint32_t i = blockIdx.x * 256u + tid;
    if (i < n && selected[i]) {
        const float4 nv = normvec[i];
        double f0 = (double)nv.x, f1 = (double)nv.y, f2 = (double)nv.z;
        const double nn = sqrt(f0 * f0 + f1 * f1 + f2 * f2);
        if (nn > 0) { f0 /= nn; f1 /= nn; f2 /= nn; }
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const float dotp = (float)fabs(f0 * V[0 * 3 + e] + f1 * V[1 * 3 + e] + f2 * V[2 * 3 + e]);
            if (dotp > 0.1736f) s[e] += (double)dotp;
            if (dotp > 0.7070f) s[3 + e] += (double)dotp;
        }
    }
#pragma unroll
    for (int e = 0; e < 6; e++) {
        const double v = wave_sum(s[e]);
        if (lane == 0) red[wave][e] = v;
    }
    __syncthreads();
    if (tid < 6) {
        const double v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        if (v != 0.0) atomicAdd(tid < 3 ? &out->contri[tid] : &out->strong[tid - 3], v);
    }
}

// ---- the device-resident iterate loop: one pass of esekf::update_iterated_dyn_share_modified per launch, one workgroup per scan ----
// Folds the partial sums of the slot's linearisation exactly as finalize_kernel does (same owner lanes, same order: bit-identical
// sums), then runs what the host did between two device passes: eigen-decomposition + degeneracy logic (the six sums over the
// points are evaluated by this workgroup when the eigenvalue bound does not decide), stale-measurement rule, 23-DoF filter step,
// convergence flags (eskf_dev.h).  A slot that finishes publishes its result record in mapped host memory.
constexpr int kStepThreads = 1024;

// JOINT (lio_batch's joint mode): the 29 sums of this pass are not the slot's own -- every rank's record of this slot (its sub-maps' sums,
// folded by joint_fold_batch, all-gathered) is added in RANK ORDER, the same order on every rank: identical bits, identical filter steps.
template <bool JOINT>
__global__ void __launch_bounds__(kStepThreads) step_batch(const SlotDesc* __restrict__ slots, const double* __restrict__ gathered, int world,
                                                           uint32_t n_slots) {
    const SlotDesc& d = slots[blockIdx.x];
    if (!d.active) return;
    EskfDev& cg = *d.ctrl;
    if (cg.status != EK_RUNNING) return;  // finished (and published) in an earlier pass
    // the filter (state, covariances, loop state -- everything before the logs) is copied into LDS, worked on by the first wave
    // with wave-level fences only, and written back; the other fifteen waves fold the partial sums and, in the rare pass that needs
    // them, the degeneracy sums
    __shared__ __attribute__((aligned(16))) EskfDev c;
    __shared__ EkWork w;
    __shared__ double acc[kAcc];
    __shared__ double red6[kStepThreads / 64][6];
    const int tid = threadIdx.x;
    constexpr int kCoreWords = (int)(offsetof(EskfDev, log) / 8);
    EK_STAMP(0);
    // the scan's size first: the loads of the partial sums depend on it, the copy of the filter does not -- both are then in flight together
    const uint32_t n = d.sd->n_ds;
    const bool skip = (d.sd->err & 1u) || n < d.min_ds;  // more voxels than max_ds / too few points (laserMapping.cpp:1246): nothing is registered
    const uint32_t nb = (n + kLinThreads - 1) / kLinThreads;
    {
        const double* src = reinterpret_cast<const double*>(&cg);
        double* dst = reinterpret_cast<double*>(&c);
        double v0 = 0, v1 = 0;
        const int k0 = tid, k1 = tid + kStepThreads;
        if (k0 < kCoreWords) v0 = src[k0];
        if (k1 < kCoreWords) v1 = src[k1];
        static_assert(kCoreWords <= 2 * kStepThreads, "filter core: two words per thread");
        double s = 0.0;
        const int comp = tid >> 5, l = tid & 31;
        if constexpr (JOINT) {
            if (!skip && tid < kAcc) {  // sum_ranks_kernel's order (comm.hip): rank 0's record, then + rank 1's, ...
                s = gathered[(size_t)blockIdx.x * 32 + tid];
                for (int r = 1; r < world; r++) s += gathered[((size_t)r * n_slots + blockIdx.x) * 32 + tid];
            }
        } else {
            if (!skip && comp < kAcc) s = fold_partials(d.partial, (uint32_t)l, nb, comp);
        }
        if (k0 < kCoreWords) dst[k0] = v0;
        if (k1 < kCoreWords) dst[k1] = v1;
        if constexpr (JOINT) {
            if (!skip && tid < kAcc) acc[tid] = s;
        } else if (!skip) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (comp < kAcc && l == 0) acc[comp] = s;
        }
    }
    __syncthreads();
    EK_STAMP(1);
    if (tid == 0) d.sd->n_tie = 0;  // the tie queue of this pass's neighbour search has been served (knn_exact_batch_kernel)
    const int log0 = c.n_log;
    // Wave 0 decides about the measurement (validity, degeneracy) while wave 1 prepares everything of the filter step that does not
    // depend on it (boxminus, manifold Jacobians, P <- J P J^T: the transcendental-heavy half of the pass).
    if (skip) {
        if (tid == 0) c.status = EK_SKIPPED;
    } else if (tid < 64) {
        ek_measure_head(c, w, acc, c.converge);
        EK_STAMP(20);
        if constexpr (JOINT) {
            // the six degeneracy sums of a joint registration live on several sub-maps and ranks: such a pass (rare: the eigenvalue bound of the
            // GLOBAL sum n n^T did not decide) is left to the host-driven joint path -- every rank reaches this decision from the same bits
            if (w.flag[1]) {
                if (tid == 0) { c.status = EK_NEEDS_HOST; c.n_pass--; if (c.converge) c.n_knn--; w.flag[0] = 0; w.flag[5] = 1; }
                EK_SYNC();
            } else {
                ek_measure_tail(c, w);
            }
        } else {
            if (!w.flag[1]) ek_measure_tail(c, w);  // the usual case: no degeneracy sums needed, everything about the measurement is settled here
            EK_STAMP(21);
        }
    } else if (tid < 128) {
        ek_step_prep(c, w);
    }
    __syncthreads();
    EK_STAMP(2);
    if (!JOINT && !skip && w.flag[1]) {  // the six degeneracy sums (laserMapping.cpp:946-964): every addend is a float in (0.1736, 1] widened to double,
                               // sums of < 2^17 of them are exact in f64 -> any reduction order gives the same bits
        double s6[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t i = tid; i < n; i += kStepThreads) {
            if (!d.selected[i]) continue;
            const float4 nv = d.normvec[i];
            double f0 = (double)nv.x, f1 = (double)nv.y, f2 = (double)nv.z;
            const double nn = sqrt(f0 * f0 + f1 * f1 + f2 * f2);
            if (nn > 0) { f0 /= nn; f1 /= nn; f2 /= nn; }
#pragma unroll
            for (int e = 0; e < 3; e++) {
                const float dotp = (float)fabs(f0 * w.eigvec[0 * 3 + e] + f1 * w.eigvec[1 * 3 + e] + f2 * w.eigvec[2 * 3 + e]);
                if (dotp > 0.1736f) s6[e] += (double)dotp;
                if (dotp > 0.7070f) s6[3 + e] += (double)dotp;
            }
        }
#pragma unroll
        for (int e = 0; e < 6; e++) {
            const double v = wave_sum(s6[e]);
            if ((tid & 63) == 0) red6[tid >> 6][e] = v;
        }
        __syncthreads();
        if (tid < 6) {
            double v = 0.0;
            for (int k = 0; k < kStepThreads / 64; k++) v += red6[k][tid];
            w.cs[tid] = v;
        }
        __syncthreads();
        if (tid < 64) ek_measure_tail(c, w);
    }
    if (tid >= 64) return;
    EK_STAMP(3);
    // ---- first wave only from here on ----
    if (!skip && c.status == EK_RUNNING && w.flag[0]) ek_step_solve(c, w);
    EK_SYNC();
    EK_STAMP(5);
    {   // write back: the filter, the log entries of this pass
        const double* src = reinterpret_cast<const double*>(&c);
        double* dst = reinterpret_cast<double*>(&cg);
        for (int k = tid; k < kCoreWords; k += 64) dst[k] = src[k];
        constexpr int kLogWords = (int)(sizeof(EkPassLog) / 8);
        const int log1 = c.n_log < kEkMaxPass ? c.n_log : kEkMaxPass;
        // (an aborted pass -- N_eff < 23 hand-over -- leaves its half-written entry behind: the host does not read past n_log)
        for (int e = log0; e < log1 + (c.status == EK_NEEDS_HOST ? 1 : 0) && e < kEkMaxPass; e++) {
            const double* ls = reinterpret_cast<const double*>(&c.log[e]);
            double* ld = reinterpret_cast<double*>(&cg.log[e]);
            for (int k = tid; k < kLogWords; k += 64) ld[k] = ls[k];
        }
    }
    EK_STAMP(6);
    if (c.status != EK_RUNNING) {
        lio_batch_result* r = d.result;
        if (tid < 26) r->state[tid] = c.x[tid];
        if (tid == 32) {
            r->status = c.status; r->n_pass = c.n_pass; r->n_knn_pass = c.n_knn; r->n_ds = (int32_t)n;
            r->n_eff = c.n_eff_last; r->degenerate = c.is_degenerate; r->radix_passes = (int32_t)((d.sd->nbits + 7u) >> 3); r->err = (int32_t)d.sd->err;
            r->loop_i = c.i; r->loop_t = c.t; r->loop_converge = c.converge; r->pad = 0;
        }
        __threadfence_system();
        EK_SYNC();
        if (tid == 0) *reinterpret_cast<volatile uint32_t*>(&r->seq) = d.seq;
    }
    EK_STAMP(7);
}

#ifdef LIO_STEP_TRACE
extern "C" int lio_debug_step_trace(unsigned long long* out64, int reset) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ek_trace), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[64] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_ek_trace), z, sizeof(z)); }
    return 0;
}
#endif

// ---- joint mode of the batched engine: a scan is registered against SEVERAL sub-maps (M on this GPU, more on other ranks) at once ------------
// Descriptors are laid out [sub-map][slot]; row 0 is the slot's own scan buffers (the cloud is downsampled there), rows 1 .. M-1 are further
// scan buffer sets (own neighbour cache, gates, partial sums) that share the slot's filter block (pose, loop state).

// the slot's downsampled cloud and size to the other sub-maps' scan buffers (blockIdx.y = (m - 1) * B + slot), device to device
__global__ void __launch_bounds__(256) share_ds_batch(const SlotDesc* __restrict__ descs, uint32_t n_slots) {
    const uint32_t slot = blockIdx.y % n_slots, m = 1u + blockIdx.y / n_slots;
    const SlotDesc& src = descs[slot];
    const SlotDesc& dst = descs[(size_t)m * n_slots + slot];
    if (!src.active) return;
    const uint32_t n = src.sd->n_ds;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst.ds_body[i] = src.ds_body[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        dst.sd->n_ds_prev = dst.sd->cache_n;  // what vg_keys does for the slot's own buffers: the neighbour cache's size before this scan
        dst.sd->n_ds = n;
        dst.sd->err = src.sd->err;
        dst.sd->n_tie = 0;
    }
}

int lio_allgather_records_internal(::lio_comm* c, const double* d_local, double* d_gathered, uint32_t n_records, hipStream_t st);  // comm.hip
// ---- the downsample of a joint round divided among the ranks ------------------------------------------------------------------------------
// Every rank registers every scan of a round (against its own sub-maps), so every rank needs every scan's downsampled cloud -- but not its own
// run of the voxel-grid chain over every scan: that part of a round did not shrink with the number of GPUs (12 of 99 us per scan at N = 1:
// the ceiling of config 5's scaling).  Rank r runs the chain for the slots [r * per, (r + 1) * per) only and the clouds travel: ONE all-gather
// per round of fixed-size slot chunks {64 header words, `cap` points}, rank-major = slot-major, on the round's stream.  The chain is
// deterministic, so the foreign clouds are bit for bit what the rank would have computed (tests/test_dist.py compares the two forms).
// `cap` is NOT max_ds (sized for the worst scan: 1.6 MB per slot in the bench, eight times a real cloud -- the all-gather would cost more than
// the chain it saves): the host keeps it at 1.25 x the largest cloud the batch has seen (batch.hip), the same on every rank because every rank
// sees every result.  A cloud that does not fit is cut, flagged in its header and in ScanDev::err (bit 2) on EVERY rank, and the job runs again
// with cap = the buffers' full size -- the path an under-launched radix sort already takes.
constexpr uint32_t kDsHdrBytes = 256;

// blockIdx.y = my slot i (global slot s0 + i): header + cloud into the rank's send buffer
__global__ void __launch_bounds__(256) pack_ds_batch(const SlotDesc* __restrict__ descs, uint32_t s0, char* __restrict__ send, size_t slot_bytes, uint32_t cap) {
    const SlotDesc& d = descs[s0 + blockIdx.y];
    char* chunk = send + (size_t)blockIdx.y * slot_bytes;
    uint32_t* hdr = reinterpret_cast<uint32_t*>(chunk);
    float4* pts = reinterpret_cast<float4*>(chunk + kDsHdrBytes);
    uint32_t n = d.active ? d.sd->n_ds : 0u;
    const bool cut = n > cap;
    if (cut) n = cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr[0] = n;
        hdr[1] = d.active ? (d.sd->err | (cut ? 4u : 0u)) : 0u;
        hdr[2] = d.active ? d.sd->nbits : 0u;
        hdr[3] = d.active;
        if (cut) d.sd->err |= 4u;  // (the owner's own copy is whole, but the round is void for everybody: the job runs again)
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) pts[i] = d.ds_body[i];
}
// blockIdx.y = slot: a slot another rank downsampled receives its cloud and the words the chain leaves behind (what share_ds_batch does for the
// further local sub-maps' rows; nbits so that every rank takes the same "sort was launched with too few passes: again" decision)
__global__ void __launch_bounds__(256) unpack_ds_batch(const SlotDesc* __restrict__ descs, uint32_t s0, uint32_t s1, const char* __restrict__ all, size_t slot_bytes) {
    const uint32_t slot = blockIdx.y;
    if (slot >= s0 && slot < s1) return;
    const SlotDesc& d = descs[slot];
    if (!d.active) return;
    const char* chunk = all + (size_t)slot * slot_bytes;
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(chunk);
    const float4* pts = reinterpret_cast<const float4*>(chunk + kDsHdrBytes);
    const uint32_t n = hdr[0];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d.ds_body[i] = pts[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.sd->n_ds_prev = d.sd->cache_n;
        d.sd->n_ds = n;
        d.sd->err = hdr[1];
        d.sd->nbits = hdr[2];
        d.sd->n_tie = 0;
    }
}

size_t ds_exchange_slot_bytes(uint32_t cap) { return ((size_t)kDsHdrBytes + (size_t)cap * sizeof(float4) + 255u) & ~(size_t)255u; }

// the chain's outputs of the slots [s0, s1) to every rank, everybody else's to this one; `per` = slots per rank (the last ranks may own fewer, or none)
int p2plane_batch_exchange_ds(::lio_comm* comm, int world, int (*gather_hook)(void*, const double*, double*, uint32_t, void*), void* gather_ctx, hipStream_t st,
                              const SlotDesc* d_descs, int n_slots, int per, int s0, int s1, uint32_t cap, char* d_send, char* d_all) {
    const size_t slot_bytes = ds_exchange_slot_bytes(cap);
    uint32_t bx = (cap + 255u) / 256u;
    if (bx > 64u) bx = 64u;
    if (bx == 0) bx = 1;
    if (s1 > s0) hipLaunchKernelGGL(pack_ds_batch, dim3(bx, (uint32_t)(s1 - s0)), 256, 0, st, d_descs, (uint32_t)s0, d_send, slot_bytes, cap);
    LIO_HIP_TRY(hipGetLastError());
    const size_t n_records = (size_t)per * slot_bytes / 256u;  // the transports count in records of 32 doubles
    if (n_records > 0xFFFFFFFFull) { set_error("joint batch: a rank's share of a round's downsampled clouds exceeds the transport's record count"); return LIO_E_CAPACITY; }
    if (gather_hook) {
        const int rc = gather_hook(gather_ctx, reinterpret_cast<const double*>(d_send), reinterpret_cast<double*>(d_all), (uint32_t)n_records, st);
        if (rc != LIO_OK) { set_error("the gather hook returned %d", rc); return rc < 0 ? rc : LIO_E_DEVICE; }
    } else {
        const int rc = lio_allgather_records_internal(comm, reinterpret_cast<const double*>(d_send), reinterpret_cast<double*>(d_all), (uint32_t)n_records, st);
        if (rc != LIO_OK) return rc;
    }
    hipLaunchKernelGGL(unpack_ds_batch, dim3(bx, (uint32_t)n_slots), 256, 0, st, d_descs, (uint32_t)s0, (uint32_t)s1, d_all, slot_bytes);
    LIO_HIP_TRY(hipGetLastError());
    (void)world;
    return LIO_OK;
}

// per slot: the partial sums of every local sub-map's linearisation folded exactly as finalize_kernel / step_batch fold them (component c is
// owned by 32 lanes: stride-32 chunks, fixed xor tree), then added in sub-map order -- what the host-driven joint path (engine.hip:
// joint_reduce) forms from the sub-maps' reports -- into the rank's 32-double record of the slot; zeros for a slot that is idle or done
__global__ void __launch_bounds__(1024) joint_fold_batch(const SlotDesc* __restrict__ descs, uint32_t n_slots, int n_maps, double* __restrict__ local32) {
    const uint32_t slot = blockIdx.x;
    const SlotDesc& d0 = descs[slot];
    const int tid = threadIdx.x, comp = tid >> 5, l = tid & 31;
    double total = 0.0;
    bool run = d0.active != 0;
    if (run) {
        const EskfDev* c = d0.ctrl;
        const uint32_t n = d0.sd->n_ds;
        run = c->status == EK_RUNNING && !((d0.sd->err & 1u) || n < d0.min_ds);
        if (run) {
            const uint32_t nb = (n + kLinThreads - 1) / kLinThreads;
            for (int m = 0; m < n_maps; m++) {
                const SlotDesc& d = descs[(size_t)m * n_slots + slot];
                double s = 0.0;
                if (comp < kAcc) s = fold_partials(d.partial, (uint32_t)l, nb, comp);
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off);
                total = m == 0 ? s : total + s;
                if (m > 0 && tid == 0) d.sd->n_tie = 0;  // this pass's tie queue of the sub-map has been served (step_batch re-arms row 0's)
            }
        }
    }
    if (comp < 32 && l == 0) local32[(size_t)slot * 32 + comp] = (run && comp < kAcc) ? total : 0.0;
}

int p2plane_batch_share(hipStream_t st, const SlotDesc* d_descs, int n_slots, int n_maps, uint32_t ds_bound) {
    if (n_maps <= 1) return LIO_OK;
    uint32_t bx = (ds_bound + 255u) / 256u;
    if (bx > 64u) bx = 64u;
    if (bx == 0) bx = 1;
    hipLaunchKernelGGL(share_ds_batch, dim3(bx, (uint32_t)(n_slots * (n_maps - 1))), 256, 0, st, d_descs, (uint32_t)n_slots);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

// one pass per loop turn: {neighbour search where the filter asks for it, linearisation} against every local sub-map, the rank's records, the
// all-gather across ranks (comm.hip; a world of one gathers nothing), the filter pass on the rank-ordered sums.  Enqueued blind, eagerly (no
// graph: a collective sits in the middle of every pass).
int p2plane_batch_update_joint(lio_map** maps, int n_maps, ::lio_comm* comm, int world, int (*gather_hook)(void*, const double*, double*, uint32_t, void*),
                               void* gather_ctx, hipStream_t st, const SlotDesc* d_descs, int n_slots, uint32_t ds_bound,
                               int n_passes, double* d_local32, double* d_gathered, BatchTimer* bt, const MapRef* d_rowmaps) {
    uint32_t lin_blocks = (ds_bound + kLinThreads - 1) / kLinThreads;
    if (lin_blocks > kLinGridCap) lin_blocks = kLinGridCap;  // (linearize_batch strides over a slot's blocks)
    if (lin_blocks == 0) lin_blocks = 1;
    for (int p = 0; p < n_passes; p++) {
        if (d_rowmaps) {
            // round 5: ONE neighbour search (+ its tie redo) and ONE linearisation over all [sub-map x slot] rows of the pass -- the rows are independent;
            // a launch per local sub-map was 3 M launches per pass (24 with eight sub-maps on one GPU), each filling a fraction of the chip.  The search
            // kernel is the sequence batch's (table / pool / counters from the row's MapRef instead of kernel arguments): the same body, the same bits
            const int rows = n_maps * n_slots;
            const StencilArgs& sa = maps[0]->stencil;
            const int sid = maps[0]->stencil_id;
            if (bt) bt->begin(1);
            const int rc = knn_seq_launch(st, d_rowmaps, d_descs, rows, (ds_bound + 15) / 16, &sa, &sid, 1);
            if (bt) bt->end(1);
            if (rc != LIO_OK) return rc;
            if (bt) bt->begin(2);
            hipLaunchKernelGGL(linearize_batch, dim3(lin_blocks, (uint32_t)rows), kLinThreads, 0, st, d_descs);
            if (bt) bt->end(2);
        }
        for (int m = 0; m < n_maps && !d_rowmaps; m++) {
            const SlotDesc* row = d_descs + (size_t)m * n_slots;
            if (bt) bt->begin(1);
            const int rc = knn_batch_launch(maps[m], st, row, n_slots, (ds_bound + 15) / 16, 0);
            if (bt) bt->end(1);
            if (rc != LIO_OK) return rc;
            if (bt) bt->begin(2);
            hipLaunchKernelGGL(linearize_batch, dim3(lin_blocks, (uint32_t)n_slots), kLinThreads, 0, st, row);
            if (bt) bt->end(2);
        }
        if (bt) bt->begin(3);
        hipLaunchKernelGGL(joint_fold_batch, dim3((uint32_t)n_slots), 1024, 0, st, d_descs, (uint32_t)n_slots, n_maps, d_local32);
        LIO_HIP_TRY(hipGetLastError());
        const double* sums = d_local32;
        if (gather_hook && world > 1) {  // lio_batch_set_gather_hook: the caller's transport instead of RCCL
            const int rc = gather_hook(gather_ctx, d_local32, d_gathered, (uint32_t)n_slots, st);
            if (rc != LIO_OK) { set_error("the gather hook returned %d", rc); return rc < 0 ? rc : LIO_E_DEVICE; }
            sums = d_gathered;
        } else if (comm && world > 1) {
            const int rc = lio_allgather_records_internal(comm, d_local32, d_gathered, (uint32_t)n_slots, st);
            if (rc != LIO_OK) return rc;
            sums = d_gathered;
        }
        hipLaunchKernelGGL((step_batch<true>), dim3((uint32_t)n_slots), kStepThreads, 0, st, d_descs, sums, world, (uint32_t)n_slots);
        if (bt) bt->end(3);
    }
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

// the whole iterated update of every slot, enqueued blind: (neighbour search if the filter asks for it, linearisation, filter pass) x
// (maximum_iter + 1); slots that converge early skip the rest of the launches
int p2plane_batch_update(lio_map* m, hipStream_t st, const SlotDesc* d_slots, int n_slots, uint32_t ds_bound, int n_passes, BatchTimer* bt, int count_touched) {
    uint32_t lin_blocks = (ds_bound + kLinThreads - 1) / kLinThreads;
    if (lin_blocks > kLinGridCap) lin_blocks = kLinGridCap;  // (linearize_batch strides over a slot's blocks)
    if (lin_blocks == 0) lin_blocks = 1;
    for (int p = 0; p < n_passes; p++) {
        if (bt) bt->begin(1);
        // sixteen lanes per query (knn.hip) + the exact redo of queued ties (usually an empty launch)
        const int rc = knn_batch_launch(m, st, d_slots, n_slots, (ds_bound + 15) / 16, count_touched);
        if (bt) bt->end(1);
        if (rc != LIO_OK) return rc;
        if (bt) bt->begin(2);
        hipLaunchKernelGGL(linearize_batch, dim3(lin_blocks, (uint32_t)n_slots), kLinThreads, 0, st, d_slots);
        if (bt) { bt->end(2); bt->begin(3); }
        hipLaunchKernelGGL((step_batch<false>), dim3((uint32_t)n_slots), kStepThreads, 0, st, d_slots, static_cast<const double*>(nullptr), 1, (uint32_t)n_slots);
        if (bt) bt->end(3);
    }
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

// sequence mode: the same blind loop, every slot against its own map
int p2plane_seq_update(hipStream_t st, const MapRef* d_maps, const SlotDesc* d_slots, int n_slots, uint32_t ds_bound, int n_passes, const StencilArgs* stencils,
                       const int* stencil_ids, int n_stencils, BatchTimer* bt) {
    uint32_t lin_blocks = (ds_bound + kLinThreads - 1) / kLinThreads;
    if (lin_blocks > kLinGridCap) lin_blocks = kLinGridCap;  // (linearize_batch strides over a slot's blocks)
    if (lin_blocks == 0) lin_blocks = 1;
    for (int p = 0; p < n_passes; p++) {
        if (bt) bt->begin(1);
        const int rc = knn_seq_launch(st, d_maps, d_slots, n_slots, (ds_bound + 15) / 16, stencils, stencil_ids, n_stencils);
        if (bt) bt->end(1);
        if (rc != LIO_OK) return rc;
        if (bt) bt->begin(2);
        hipLaunchKernelGGL(linearize_batch, dim3(lin_blocks, (uint32_t)n_slots), kLinThreads, 0, st, d_slots);
        if (bt) { bt->end(2); bt->begin(3); }
        hipLaunchKernelGGL((step_batch<false>), dim3((uint32_t)n_slots), kStepThreads, 0, st, d_slots, static_cast<const double*>(nullptr), 1, (uint32_t)n_slots);
        if (bt) bt->end(3);
    }
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

int p2plane_reduce(lio_map* m, lio_scan* s, const PoseArgs& pose, int redo_knn) {
    const uint32_t bound = s->have_ds > 0 ? (uint32_t)s->have_ds : (s->n_raw && s->n_raw < s->max_ds ? s->n_raw : s->max_ds);
    uint32_t blocks = (bound + kLinThreads - 1) / kLinThreads;
    if (blocks == 0) blocks = 1;
    if (blocks > s->partial_blocks) blocks = s->partial_blocks;
    kt_begin(s, 1);
    hipLaunchKernelGGL(linearize_kernel, blocks, kLinThreads, 0, s->stream, pose, redo_knn, s->dev, s->ds_body, s->ds_world, s->nn_pts,
                       s->max_ds, s->nn_cnt, s->selected, s->normvec, s->partial);
    hipLaunchKernelGGL(finalize_kernel, 1, kFinThreads, 0, s->stream, s->dev, s->partial, m ? m->dev : nullptr, s->h_result_dev);
    kt_end(s, 1);
    s->seq_expected++;
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

// map_incremental (laserMapping.cpp:523-563): which downsampled points enter the map, and in which order.  The reference fills two lists in
// point order -- PointToAdd, PointNoNeedDownsample -- and inserts them one after the other (:571-572); the order decides which voxel a scan
// touches LAST, i.e. the LRU positions, hence the eviction set when the quota cuts inside a scan.  Two launches: classify (flag per point,
// counts per workgroup), then an ordered scatter (exclusive scan over the workgroup counts, ballot + prefix sums inside) -- the staged
// batch is PointToAdd in point order followed by PointNoNeedDownsample in point order, independent of wave scheduling.
constexpr int kClsThreads = 256;

__device__ __forceinline__ void classify_body(const PoseArgs& pose, const ScanDev* __restrict__ sd, const float4* __restrict__ ds_body,
                                              float4* __restrict__ ds_world, const float4* __restrict__ nn_pts, uint32_t nn_stride,
                                              const int32_t* __restrict__ nn_cnt, float map_leaf, int ekf_inited, int seed_all,
                                              uint8_t* __restrict__ cls, uint32_t* __restrict__ blk_cnt /* [2][stride] */,
                                              uint32_t vb = 0xFFFFFFFFu, uint32_t stride = 0u) {
    // vb / stride: the block of kClsThreads points this call classifies and the row length of blk_cnt -- the launch's own block and grid by default
    // (one scan at a time: the host knows the cloud's size); the sequence batch strides over a slot's blocks with a grid sized for the typical cloud
    // (its capacity-sized grid -- 391 workgroups per slot for max_ds = 100 000 -- left six of seven workgroups to start, wait and leave)
    if (vb == 0xFFFFFFFFu) { vb = blockIdx.x; stride = gridDim.x; }
    const uint32_t n = sd->n_ds;
    const uint32_t i = vb * kClsThreads + threadIdx.x;
    int c = 0;  // 0 not added, 1 PointToAdd, 2 PointNoNeedDownsample
    if (i < n) {
        double pi[3];
        float4 pw;
        body_to_world_d(pose, ds_body[i], pi, pw);
        ds_world[i] = pw;
        const int cnt = nn_cnt[i];
        c = 1;
        if (!seed_all && cnt > 0 && ekf_inited) {
            const double fs = (double)map_leaf;
            float4 mid;
            mid.x = (float)(floor((double)pw.x / fs) * fs + 0.5 * fs);
            mid.y = (float)(floor((double)pw.y / fs) * fs + 0.5 * fs);
            mid.z = (float)(floor((double)pw.z / fs) * fs + 0.5 * fs);
            const float dist = ((pw.x - mid.x) * (pw.x - mid.x) + (pw.y - mid.y) * (pw.y - mid.y)) + (pw.z - mid.z) * (pw.z - mid.z);
            const double half = 0.5 * fs;
            const float4 n0 = nn_pts[i];
            if (fabs((double)(n0.x - mid.x)) > half && fabs((double)(n0.y - mid.y)) > half && fabs((double)(n0.z - mid.z)) > half) {
                c = 2;  // PointNoNeedDownsample
            } else if (cnt >= 5) {
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const float4 q = nn_pts[(size_t)k * nn_stride + i];
                    const float dq = ((q.x - mid.x) * (q.x - mid.x) + (q.y - mid.y) * (q.y - mid.y)) + (q.z - mid.z) * (q.z - mid.z);
                    if (dq < dist) c = 0;
                }
            }
        }
        cls[i] = (uint8_t)c;
    }
    __shared__ uint32_t wc[kClsThreads / 64][2];
    const unsigned long long m1 = __ballot(c == 1), m2 = __ballot(c == 2);
    if ((threadIdx.x & 63) == 0) { wc[threadIdx.x >> 6][0] = __popcll(m1); wc[threadIdx.x >> 6][1] = __popcll(m2); }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t t = 0;
        for (int w = 0; w < kClsThreads / 64; w++) t += wc[w][threadIdx.x];
        blk_cnt[threadIdx.x * stride + vb] = t;
    }
}

__device__ __forceinline__ void classify_scatter_body(const ScanDev* __restrict__ sd, const float4* __restrict__ ds_world,
                                                      const uint8_t* __restrict__ cls, const uint32_t* __restrict__ blk_cnt,
                                                      float4* __restrict__ stage, MapDev* md, uint32_t vb = 0xFFFFFFFFu, uint32_t nb = 0u,
                                                      uint32_t stride = 0u) {
    // vb / nb / stride: this call's block, the number of blocks that hold points, the row length of blk_cnt (defaults: the launch's block and grid)
    if (vb == 0xFFFFFFFFu) { vb = blockIdx.x; nb = gridDim.x; stride = gridDim.x; }
    const uint32_t n = sd->n_ds;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // exclusive prefix of this workgroup in each list, and the size of the first list (fixed order: deterministic slots)
    __shared__ uint32_t red[kClsThreads / 64][3];
    uint32_t preA = 0, preB = 0, totA = 0;
    for (uint32_t b = tid; b < nb; b += kClsThreads) {
        const uint32_t a = blk_cnt[b], bb = blk_cnt[stride + b];
        totA += a;
        if (b < vb) { preA += a; preB += bb; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { preA += __shfl_xor(preA, off); preB += __shfl_xor(preB, off); totA += __shfl_xor(totA, off); }
    if (lane == 0) { red[wave][0] = preA; red[wave][1] = preB; red[wave][2] = totA; }
    __syncthreads();
    preA = preB = totA = 0;
    for (int w = 0; w < kClsThreads / 64; w++) { preA += red[w][0]; preB += red[w][1]; totA += red[w][2]; }
    __syncthreads();
    const uint32_t i = vb * kClsThreads + tid;
    const int c = i < n ? (int)cls[i] : 0;
    const unsigned long long m1 = __ballot(c == 1), m2 = __ballot(c == 2);
    __shared__ uint32_t wc[kClsThreads / 64][2];
    if (lane == 0) { wc[wave][0] = __popcll(m1); wc[wave][1] = __popcll(m2); }
    __syncthreads();
    uint32_t offA = preA, offB = totA + preB, tb = 0;
    for (int w = 0; w < kClsThreads / 64; w++) {
        if (w < wave) { offA += wc[w][0]; offB += wc[w][1]; }
        tb += wc[w][1];
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (c == 1) stage[offA + __popcll(m1 & below)] = ds_world[i];
    else if (c == 2) stage[offB + __popcll(m2 & below)] = ds_world[i];
    if (vb == nb - 1 && tid == 0) md->n_add = totA + preB + tb;  // (the last block's exclusive prefix + its own count = the total)
}

__global__ void __launch_bounds__(kClsThreads) classify_kernel(PoseArgs pose, const ScanDev* __restrict__ sd, const float4* __restrict__ ds_body,
                                                               float4* __restrict__ ds_world, const float4* __restrict__ nn_pts, uint32_t nn_stride,
                                                               const int32_t* __restrict__ nn_cnt, float map_leaf, int ekf_inited, int seed_all,
                                                               uint8_t* __restrict__ cls, uint32_t* __restrict__ blk_cnt) {
    classify_body(pose, sd, ds_body, ds_world, nn_pts, nn_stride, nn_cnt, map_leaf, ekf_inited, seed_all, cls, blk_cnt);
}
__global__ void __launch_bounds__(kClsThreads) classify_scatter_kernel(const ScanDev* __restrict__ sd, const float4* __restrict__ ds_world,
                                                                       const uint8_t* __restrict__ cls, const uint32_t* __restrict__ blk_cnt,
                                                                       float4* __restrict__ stage, MapDev* md) {
    classify_scatter_body(sd, ds_world, cls, blk_cnt, stage, md);
}

// ---- sequence mode (lio_batch_create_sequences): map_incremental of every slot inside the round -----------------------------------------------
// One small workgroup per slot after the last filter pass: does this round's scan enter its map (the update finished on the device -- a scan
// handed to the host, skipped or over capacity is finished and inserted by the slot's engine afterwards), and the travel distance of the lidar
// origin after it (laserMapping.cpp:1288-1291: what AddPoints stamps new voxels with; the host repeats the same additions for its own copy).
__global__ void __launch_bounds__(64) seq_begin_insert_kernel(const MapRef* __restrict__ maps, const SlotDesc* __restrict__ slots, SeqDev* __restrict__ seq) {
    const SlotDesc& d = slots[blockIdx.x];
    if (threadIdx.x != 0) return;
    SeqDev& q = seq[blockIdx.x];
    q.go = 0;
    q.map_err = 0;
    q.n_add = 0;
    if (!d.active) return;
    const MapRef& r = maps[blockIdx.x];
    const EskfDev* c = d.ctrl;
    double travel = r.travel_prev;
    if (c->status == EK_DONE && !(d.sd->err & 1u)) {
        // pos_lid = pos + rot * offset_T_L_I, in the operation order of quat_rotate (eskf.cpp)
        const double* x = c->x;
        const double qx = x[3], qy = x[4], qz = x[5], qw = x[6];
        const double v[3] = {x[11], x[12], x[13]};
        const double t0 = 2.0 * (qy * v[2] - qz * v[1]), t1 = 2.0 * (qz * v[0] - qx * v[2]), t2 = 2.0 * (qx * v[1] - qy * v[0]);
        const double o0 = v[0] + qw * t0 + (qy * t2 - qz * t1), o1 = v[1] + qw * t1 + (qz * t0 - qx * t2), o2 = v[2] + qw * t2 + (qx * t1 - qy * t0);
        const double p[3] = {x[0] + o0, x[1] + o1, x[2] + o2};
        double d2 = 0;
        for (int i = 0; i < 3; i++) { const double dd = p[i] - r.last_pos_lid[i]; d2 += dd * dd; }
        travel = travel + sqrt(d2);
        q.go = r.do_insert ? 1u : 0u;
    }
    q.travel = travel;
}
__global__ void __launch_bounds__(kClsThreads) classify_seq(const MapRef* __restrict__ maps, const SlotDesc* __restrict__ slots, const SeqDev* __restrict__ seq) {
    if (!seq[blockIdx.y].go) return;
    const SlotDesc& d = slots[blockIdx.y];
    const MapRef& r = maps[blockIdx.y];
    const PoseArgs pose = pose_from_state(d.ctrl->x);
    const uint32_t n = d.sd->n_ds, stride = (d.max_ds + kClsThreads - 1) / kClsThreads;
    for (uint32_t vb = blockIdx.x; vb * kClsThreads < n; vb += gridDim.x) {
        classify_body(pose, d.sd, d.ds_body, d.ds_world, d.nn_pts, d.max_ds, d.nn_cnt, r.map_leaf, (int)r.ekf_inited, 0, reinterpret_cast<uint8_t*>(d.keys_a), d.hist, vb, stride);
        __syncthreads();  // (the workgroup's LDS staging is reused by its next block)
    }
}
__global__ void __launch_bounds__(kClsThreads) classify_scatter_seq(const MapRef* __restrict__ maps, const SlotDesc* __restrict__ slots,
                                                                    const SeqDev* __restrict__ seq) {
    if (!seq[blockIdx.y].go) return;
    const SlotDesc& d = slots[blockIdx.y];
    const MapRef& r = maps[blockIdx.y];
    const uint32_t n = d.sd->n_ds, stride = (d.max_ds + kClsThreads - 1) / kClsThreads, nb = (n + kClsThreads - 1) / kClsThreads;
    if (nb == 0) {  // an empty cloud: nothing is staged
        if (blockIdx.x == 0 && threadIdx.x == 0) r.md->n_add = 0;
        return;
    }
    for (uint32_t vb = blockIdx.x; vb < nb; vb += gridDim.x) {
        classify_scatter_body(d.sd, d.ds_world, reinterpret_cast<const uint8_t*>(d.keys_a), d.hist, r.stage, r.md, vb, nb, stride);
        __syncthreads();
    }
}
// the read-back record of every slot: the posterior covariance and what the insert left in the map's counters
__global__ void __launch_bounds__(256) seq_finish_kernel(const MapRef* __restrict__ maps, const SlotDesc* __restrict__ slots, SeqDev* __restrict__ seq) {
    const SlotDesc& d = slots[blockIdx.x];
    if (!d.active) return;
    SeqDev& q = seq[blockIdx.x];
    const EskfDev* c = d.ctrl;
    for (int k = threadIdx.x; k < kEkN * kEkN; k += 256) q.P[k] = c->P[k];
    if (threadIdx.x == 0) {
        const MapDev* md = maps[blockIdx.x].md;
        q.map_err = md->err;
        q.n_add = q.go ? md->n_add : 0u;
        q.n_voxels = md->n_voxels;
        q.n_points = md->n_points;
    }
}

int incremental_classify(lio_map* m, lio_scan* s, const PoseArgs& pose, float map_leaf, int ekf_inited, int seed_all) {
    const uint32_t bound = s->have_ds > 0 ? (uint32_t)s->have_ds : (s->n_raw && s->n_raw < s->max_ds ? s->n_raw : s->max_ds);
    if (bound > m->stage_cap) {
        set_error("map_incremental: %u points exceed the staging capacity %llu", bound, (unsigned long long)m->stage_cap);
        return LIO_E_CAPACITY;
    }
    if ((uint64_t)bound > 4ull * s->max_raw) { set_error("map_incremental: %u points exceed the flag scratch", bound); return LIO_E_CAPACITY; }
    uint32_t blocks = (bound + kClsThreads - 1) / kClsThreads;
    if (blocks == 0) blocks = 1;
    // scratch of the downsample that is free by now: the radix histograms (2 x blocks words), a key buffer (one flag byte per point)
    uint32_t* blk_cnt = s->hist;
    uint8_t* cls = reinterpret_cast<uint8_t*>(s->keys_a);
    hipLaunchKernelGGL(classify_kernel, blocks, kClsThreads, 0, s->stream, pose, s->dev, s->ds_body, s->ds_world, s->nn_pts, s->max_ds, s->nn_cnt,
                       map_leaf, ekf_inited, seed_all, cls, blk_cnt);
    hipLaunchKernelGGL(classify_scatter_kernel, blocks, kClsThreads, 0, s->stream, s->dev, s->ds_world, cls, blk_cnt, m->stage, m->dev);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

// map_incremental (laserMapping.cpp:523-576) of every slot whose update finished, against its own map, then the read-back records
int p2plane_seq_insert(hipStream_t st, const MapRef* d_maps, const SlotDesc* d_slots, SeqDev* d_seq, int n_slots, uint32_t ds_bound, int any_lru) {
    uint32_t blocks = (ds_bound + kClsThreads - 1) / kClsThreads;
    if (blocks > 64u) blocks = 64u;  // (classify_seq / classify_scatter_seq stride over a slot's blocks: 64 x 256 = 16 384 points per sweep of the grid)
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(seq_begin_insert_kernel, dim3((uint32_t)n_slots), 64, 0, st, d_maps, d_slots, d_seq);
    hipLaunchKernelGGL(classify_seq, dim3(blocks, (uint32_t)n_slots), kClsThreads, 0, st, d_maps, d_slots, d_seq);
    hipLaunchKernelGGL(classify_scatter_seq, dim3(blocks, (uint32_t)n_slots), kClsThreads, 0, st, d_maps, d_slots, d_seq);
    LIO_HIP_TRY(hipGetLastError());
    const int rc = map_insert_seq(st, d_maps, d_seq, n_slots, ds_bound, any_lru);
    if (rc != LIO_OK) return rc;
    hipLaunchKernelGGL(seq_finish_kernel, dim3((uint32_t)n_slots), 256, 0, st, d_maps, d_slots, d_seq);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

// the degeneracy sums against the eigenvectors the host wrote into d_result->eigvec
int p2plane_degeneracy(lio_scan* s) {
    const uint32_t bound = s->have_ds > 0 ? (uint32_t)s->have_ds : (s->n_raw && s->n_raw < s->max_ds ? s->n_raw : s->max_ds);
    hipLaunchKernelGGL(degeneracy_kernel, (bound + 255) / 256 ? (bound + 255) / 256 : 1, 256, 0, s->stream, s->dev, s->selected, s->normvec,
                       s->d_result);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

PoseArgs make_pose(const double pose_wi[7], const double ext_il[7]) {
    PoseArgs p;
    for (int i = 0; i < 3; i++) { p.tw[i] = pose_wi[i]; p.tl[i] = ext_il[i]; }
    for (int i = 0; i < 4; i++) { p.qw[i] = pose_wi[3 + i]; p.ql[i] = ext_il[3 + i]; }
    return p;
}

}  // namespace lio
