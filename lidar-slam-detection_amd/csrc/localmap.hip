// localmap.hip -- local-map assembly for localisation, resident on the device: Localization::runUpdateLocalMap
// (/root/reference/slam/localization/src/localization.cpp:303-373).  The reference keeps every key frame's world-frame cloud on
// the host, and every 10 m of travel concatenates the key frames within 30 m of the pose (nearest first, thinned by
// key_frame_distance, until 200 000 points), runs a VoxelGrid over the result and hands it to the matcher as its new target
// (a host -> device upload of up to 200 k points each time).  Here the key-frame clouds live in one HBM buffer; an update is
// a handful of device-to-device segment copies into a scan buffer, the VoxelGrid kernels (voxelgrid.hip) and the NDT target
// build (ndt.hip) -- nothing crosses PCIe.  Key-frame selection (tens of candidates) stays on the host.
#include <algorithm>
#include <array>
#include <cmath>
#include <vector>

#include "lio_common.h"

using namespace lio;

struct lio_localmap {
    int device;
    float4* pts;  // all key-frame clouds, back to back
    uint64_t cap, used;
    std::vector<uint64_t> off;
    std::vector<uint32_t> cnt;
    std::vector<std::array<float, 3>> pos;
    lio_scan* scan;  // assembly + VoxelGrid buffers
    uint32_t max_local;
    bool have_last;
    double last[3];
    bool have_map;
};

extern "C" {

lio_localmap* lio_localmap_create(int device, uint64_t max_total_points, uint32_t max_local_points, uint32_t max_keyframe_points) {
    if (max_total_points == 0 || max_local_points == 0 || max_keyframe_points == 0) { set_error("lio_localmap_create: bad argument"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_error("lio_localmap_create: no HIP device %d (this library has no CPU fallback)", device); return nullptr; }
    lio_localmap* lm = new lio_localmap();
    lm->device = device;
    lm->cap = max_total_points;
    lm->used = 0;
    lm->max_local = max_local_points;
    lm->have_last = lm->have_map = false;
    lm->pts = nullptr;
    // the concatenation stops AFTER the key frame that crosses max_local_points (localization.cpp:345-347)
    lm->scan = lio_scan_create(device, max_local_points + max_keyframe_points, max_local_points + max_keyframe_points);
    if (!lm->scan || hipMalloc(reinterpret_cast<void**>(&lm->pts), max_total_points * sizeof(float4)) != hipSuccess) {
        if (lm->scan) lio_scan_destroy(lm->scan);
        set_error("lio_localmap_create: allocation failed");
        delete lm;
        return nullptr;
    }
    return lm;
}

void lio_localmap_destroy(lio_localmap* lm) {
    if (!lm) return;
    hipSetDevice(lm->device);
    lio_scan_destroy(lm->scan);
    hipFree(lm->pts);
    delete lm;
}

// one key frame: its cloud already transformed to the map frame (KeyFrame::mTransfromPoints) and its position (the graph kd-tree's point)
int lio_localmap_add_keyframe(lio_localmap* lm, const float* world_xyzi, uint32_t n, const float position[3]) {
    if (!lm || (!world_xyzi && n) || !position) return LIO_E_INVALID;
    if (lm->used + n > lm->cap) { set_error("key-frame store full (%llu + %u > %llu points)", (unsigned long long)lm->used, n, (unsigned long long)lm->cap); return LIO_E_CAPACITY; }
    if (n > lm->scan->max_raw - lm->max_local) { set_error("key frame of %u points exceeds max_keyframe_points", n); return LIO_E_CAPACITY; }
    hipSetDevice(lm->device);
    if (n) LIO_HIP_TRY(hipMemcpy(lm->pts + lm->used, world_xyzi, (size_t)n * sizeof(float4), hipMemcpyHostToDevice));
    lm->off.push_back(lm->used);
    lm->cnt.push_back(n);
    lm->pos.push_back({position[0], position[1], position[2]});
    lm->used += n;
    return (int)lm->off.size() - 1;
}

int lio_localmap_num_keyframes(lio_localmap* lm) { return lm ? (int)lm->off.size() : LIO_E_INVALID; }

// One pass of runUpdateLocalMap's loop body.  Returns 0: pose within update_distance of the last update, nothing done;
// 1: target replaced; 2: no key frame within `radius` (target dropped, "out of map"); 3: nearest key frame >= 20 m away (target
// dropped).  n_keyframes / n_points report what went into the target (after the VoxelGrid).
int lio_localmap_update(lio_localmap* lm, lio_ndt* ndt, const double pose_xyz[3], double update_distance, double radius, double key_frame_distance,
                        float leaf, int* n_keyframes, uint32_t* n_points) {
    if (!lm || !ndt || !pose_xyz || !(leaf > 0.f)) return LIO_E_INVALID;
    if (n_keyframes) *n_keyframes = 0;
    if (n_points) *n_points = 0;
    if (lm->have_map && lm->have_last) {
        const double dx = pose_xyz[0] - lm->last[0], dy = pose_xyz[1] - lm->last[1], dz = pose_xyz[2] - lm->last[2];
        if (!(std::sqrt(dx * dx + dy * dy + dz * dz) > update_distance)) return 0;
    }
    hipSetDevice(lm->device);
    // mGraphKDTree->radiusSearch(searchPoint, radius): key frames within the radius, nearest first (f32 like pcl::PointXYZ)
    const float sx = (float)pose_xyz[0], sy = (float)pose_xyz[1], sz = (float)pose_xyz[2];
    std::vector<std::pair<float, int>> hits;
    for (size_t i = 0; i < lm->pos.size(); i++) {
        const float dx = lm->pos[i][0] - sx, dy = lm->pos[i][1] - sy, dz = lm->pos[i][2] - sz;
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 <= (float)(radius * radius)) hits.push_back({d2, (int)i});
    }
    if (hits.empty()) {
        lm->have_map = false;
        lio_ndt_set_target_device(ndt, nullptr, 0);
        return 2;
    }
    std::stable_sort(hits.begin(), hits.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
    for (int i = 0; i < 3; i++) lm->last[i] = pose_xyz[i];
    lm->have_last = true;
    lio_scan* s = lm->scan;
    uint32_t total = 0;
    int used = 0;
    float accum = 0.f;
    for (const auto& h : hits) {
        const float distance = std::sqrt(h.first);
        if (total != 0 && (double)(distance - accum) < key_frame_distance) continue;
        accum = distance;
        const uint32_t c = lm->cnt[h.second];
        if (c) LIO_HIP_TRY(hipMemcpyAsync(s->raw_own + total, lm->pts + lm->off[h.second], (size_t)c * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
        total += c;
        used++;
        if (total >= lm->max_local) break;
    }
    s->raw = s->raw_own;
    s->n_raw = total;
    uint32_t n_ds = 0;
    int rc = lio_scan_voxel_downsample(s, leaf, 1, &n_ds);
    if (rc != LIO_OK) return rc;
    if (n_keyframes) *n_keyframes = used;
    if (n_points) *n_points = n_ds;
    if (hits[0].first >= 400.f) {  // nearest key frame 20 m away or more
        lm->have_map = false;
        lio_ndt_set_target_device(ndt, nullptr, 0);
        return 3;
    }
    rc = lio_ndt_set_target_device(ndt, s->ds_body, n_ds);
    if (rc != LIO_OK) return rc;
    lm->have_map = true;
    return 1;
}

// the assembled, downsampled local map of the last update (test visibility)
int lio_localmap_download(lio_localmap* lm, float* out_xyzi, uint32_t cap) { return (!lm || !out_xyzi) ? LIO_E_INVALID : lio_scan_download_ds(lm->scan, out_xyzi, cap); }

}  // extern "C"
