// knn_dev.h -- device inlines shared by the stencil-kNN kernel (knn.hip) and the kernels that read the pose from the device-resident
// filter state.
#pragma once
#include "eskf_dev.h"
#include "lio_common.h"

namespace lio {

__device__ inline void body_to_world(const PoseArgs& P, const float4 pb, float4& pw) {
    // laserMapping.cpp:831-836: p_global = rot * (offset_R_L_I * p_body + offset_T_L_I) + pos, in double, stored float.
    // Quaternion * vector as Eigen's _transformVector: uv = 2 (q.vec x v); v + w uv + q.vec x uv
    const double vx = (double)pb.x, vy = (double)pb.y, vz = (double)pb.z;
    double ux = P.ql[1] * vz - P.ql[2] * vy, uy = P.ql[2] * vx - P.ql[0] * vz, uz = P.ql[0] * vy - P.ql[1] * vx;
    ux += ux; uy += uy; uz += uz;
    double cx = P.ql[1] * uz - P.ql[2] * uy, cy = P.ql[2] * ux - P.ql[0] * uz, cz = P.ql[0] * uy - P.ql[1] * ux;
    const double ix = ((vx + P.ql[3] * ux) + cx) + P.tl[0];
    const double iy = ((vy + P.ql[3] * uy) + cy) + P.tl[1];
    const double iz = ((vz + P.ql[3] * uz) + cz) + P.tl[2];
    ux = P.qw[1] * iz - P.qw[2] * iy; uy = P.qw[2] * ix - P.qw[0] * iz; uz = P.qw[0] * iy - P.qw[1] * ix;
    ux += ux; uy += uy; uz += uz;
    cx = P.qw[1] * uz - P.qw[2] * uy; cy = P.qw[2] * ux - P.qw[0] * uz; cz = P.qw[0] * uy - P.qw[1] * ux;
    pw.x = (float)(((ix + P.qw[3] * ux) + cx) + P.tw[0]);
    pw.y = (float)(((iy + P.qw[3] * uy) + cy) + P.tw[1]);
    pw.z = (float)(((iz + P.qw[3] * uz) + cz) + P.tw[2]);
    pw.w = pb.w;
}


// A voxel with key c holds points with |p_a * inv_res - c_a| <= 0.5 evaluated in f32 (pos2grid): |p_a - c_a * res| <= res / 2 up to
// rounding of the product (6e-8 |p_a|).  The bound below shrinks the box side by 0.5 mm + 1e-6 |q_a| per axis (covers that rounding
// and the one of q_a - c_a * res up to |q| ~ 10 km) and the sum by 1e-5 (covers the f32 evaluation of the candidates' own d2), so it
// never exceeds the d2 the sweep would compute for a point of that voxel.
__device__ inline uint32_t cell_min_d2_bits(float qx, float qy, float qz, int cx, int cy, int cz, float res) {
    const float h = 0.5f * res;
    const float ax = fmaxf(fabsf(qx - (float)cx * res) - h - (5e-4f + 1e-6f * fabsf(qx)), 0.f);
    const float ay = fmaxf(fabsf(qy - (float)cy * res) - h - (5e-4f + 1e-6f * fabsf(qy)), 0.f);
    const float az = fmaxf(fabsf(qz - (float)cz * res) - h - (5e-4f + 1e-6f * fabsf(qz)), 0.f);
    return __float_as_uint((ax * ax + (ay * ay + az * az)) * 0.99999f);
}


// a double that is the same in every lane, moved to scalar registers (the compiler cannot prove a value loaded through a pointer
// uniform and would otherwise keep the 14 doubles of a pose in 28 vector registers per lane for the whole kernel: occupancy 4 instead
// of 6 for the kNN kernel)
__device__ inline double uniform_double(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// the rigid transforms of a 26-number filter state (lio_hip.h layout) as the kernels take them
__device__ inline PoseArgs pose_from_state(const double* __restrict__ x) {
    PoseArgs p;
#pragma unroll
    for (int i = 0; i < 3; i++) { p.tw[i] = uniform_double(x[i]); p.tl[i] = uniform_double(x[11 + i]); }
#pragma unroll
    for (int i = 0; i < 4; i++) { p.qw[i] = uniform_double(x[3 + i]); p.ql[i] = uniform_double(x[7 + i]); }
    return p;
}

// What a kernel of the batched loop decides on -- the slot's filter status, its "search again" flag, the scan's size -- and the pose it then works
// with, requested TOGETHER.  The short-circuit form `c->status != EK_RUNNING || !c->converge || d.sd->n_ds < d.min_ds` followed by
// pose_from_state(c->x) is four dependent memory round trips at the head of every workgroup (~4 us of a workgroup that lives ~15); a slot that
// turns out to be idle has asked for 14 doubles it does not use.
struct SlotGate {
    int status, converge;
    uint32_t n_ds;
    PoseArgs pose;
};
__device__ __forceinline__ SlotGate slot_gate(const SlotDesc& d) {
    const EskfDev* c = d.ctrl;
    SlotGate g;
    uint32_t st = (uint32_t)c->status, cv = (uint32_t)c->converge, n = d.sd->n_ds;
    double x[14];
#pragma unroll
    for (int i = 0; i < 14; i++) x[i] = c->x[i];
    pin_loaded(st);
    pin_loaded(cv);
    pin_loaded(n);
#pragma unroll
    for (int i = 0; i < 14; i++) asm volatile("" : "+v"(x[i]));
    g.status = (int)st;
    g.converge = (int)cv;
    g.n_ds = n;
    g.pose = pose_from_state(x);
    return g;
}

// the gate alone (kernels that keep the pose out of their scalar registers: the neighbour search reads it into LDS, see knn.hip)
struct SlotGateLite {
    int status, converge;
    uint32_t n_ds;
};
__device__ __forceinline__ SlotGateLite slot_gate_lite(const SlotDesc& d) {
    const EskfDev* c = d.ctrl;
    uint32_t st = (uint32_t)c->status, cv = (uint32_t)c->converge, n = d.sd->n_ds;
    pin_loaded(st);
    pin_loaded(cv);
    pin_loaded(n);
    SlotGateLite g;
    g.status = (int)st;
    g.converge = (int)cv;
    g.n_ds = n;
    return g;
}
// threads 0..13 of a workgroup copy the rigid transforms of a 26-number filter state (lio_hip.h layout: pos, rot, offset_R_L_I, offset_T_L_I) into a
// PoseArgs in LDS; the caller's next barrier publishes it
__device__ __forceinline__ void pose_fill_from_state(PoseArgs& P, const double* __restrict__ x) {
    const int t = threadIdx.x;
    if (t < 14) {
        const double v = x[t];
        double* dst = t < 3 ? &P.tw[t] : (t < 7 ? &P.qw[t - 3] : (t < 11 ? &P.ql[t - 7] : &P.tl[t - 11]));
        *dst = v;
    }
}

}  // namespace lio
