"""The FastLIO-odometry slice of the reference's outer boundary, `slam_wrapper` (pybind module of
/root/reference/slam/src/slam_wrapper.cpp), over the HIP engine: the same function names, argument order, units and return
dictionary for the calls that sit on the hot path --

    init_slam, set_ins_external_param, set_imu_external_param, setup_slam, process, deinit_slam

`process(points, points_attr, image_dict, image_stream_dict, image_param, rtk_dict, imu_list, timestamp)` does what the
reference does between Python and the filter: numpy_to_imu (py_utils.cpp:244-258: deg/s -> rad/s, g -> m/s^2, us -> s),
numpy_to_pointcloud / pydict_to_cloud (py_utils.cpp:149-181: N x 4 f32 + N x 2 attr, col 0 = per-point offset in us), the
lidar -> INS static transform (slam_base.h:83-85, pcl::transformPointCloud in f32), HDL_FastLIO::feedImuData /
feedPointData (fastlio.cpp:190-210), the LIO loop (fastlio.cpp:262-276) and getPose's frame change
odom = T_imu_ins^-1 * odom * T_imu_ins (fastlio.cpp:269-270), returned as pose["odom_matrix"].

Everything behind the odometry -- graph backend, GNSS fusion, floor detection, key-frame management, map export, image
handling -- is outside the hot path this repository covers: those entry points raise NotImplementedError, and the
pose's geographic fields stay zero.  One global instance per process, like the reference's `slam_ptr`."""
import numpy as np

from . import capi, lio

_ANG2RAD = 0.01745329251994  # the reference's truncated constant (slam_utils.cpp:88), not pi / 180
_state = None


def get_transform_from_rpyt(x, y, z, yaw, pitch, roll):
    """getTransformFromRPYT (slam/common/... :89-96): translation * Rz(yaw) * Rx(pitch) * Ry(roll), angles in degrees"""
    cy, sy = np.cos(yaw * _ANG2RAD), np.sin(yaw * _ANG2RAD)
    cp, sp = np.cos(pitch * _ANG2RAD), np.sin(pitch * _ANG2RAD)
    cr, sr = np.cos(roll * _ANG2RAD), np.sin(roll * _ANG2RAD)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Rx = np.array([[1.0, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Ry = np.array([[cr, 0, sr], [0, 1.0, 0], [-sr, 0, cr]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Rx @ Ry
    T[:3, 3] = [x, y, z]
    return T


class _Slam:
    def __init__(self, mode, method, sensors, scan_period=0.1):
        self.mode, self.method, self.sensors = mode, method, list(sensors)
        self.static = np.eye(4)      # lidar -> INS   (set_ins_external_param)
        self.imu_static = np.eye(4)  # IMU extrinsic  (set_imu_external_param)
        self.scan_period = scan_period
        self.engine = None
        self.lidar = next((s for s in self.sensors if s not in ("RTK", "IMU")), None)


def init_slam(mode, map_path, method, sensor_input, resolution, dist_threshold, degree_threshold, frame_range):
    """slam_wrapper.cpp init_slam: returns the sensor list the back end will consume"""
    global _state
    if method != "FastLIO":
        raise NotImplementedError(f"only the FastLIO odometry path is built on the device (method={method!r})")
    _state = _Slam(mode, method, sensor_input)
    return list(sensor_input)


def set_ins_external_param(x, y, z, yaw, pitch, roll):
    _state.static = get_transform_from_rpyt(x, y, z, yaw, pitch, roll)


def set_imu_external_param(x, y, z, yaw, pitch, roll):
    _state.imu_static = get_transform_from_rpyt(x, y, z, yaw, pitch, roll)


def setup_slam(max_points=8_000_000, max_voxels=1 << 21, device=0):
    """HDL_FastLIO::init (fastlio.cpp:153-171): T_imu_ins = T_imu * T_static^-1 is the lidar(INS-frame cloud) -> IMU extrinsic"""
    s = _state
    s.T_imu_ins = s.imu_static @ np.linalg.inv(s.static)
    s.engine = lio.Engine(resolution=0.5, stencil=75, max_points=max_points, max_voxels=max_voxels, max_raw=1 << 18, max_ds=100000, device=device)
    s.engine.fastlio_init(extT=s.T_imu_ins[:3, 3], extR=s.T_imu_ins[:3, :3], filter_num=1, max_point_num=-1, scan_period=s.scan_period, undistort=True)
    return True


def deinit_slam():
    global _state
    if _state is not None and _state.engine is not None:
        _state.engine.close()
    _state = None


def _transform_f32(points, T):
    """pcl::transformPointCloud with a Matrix4f: x' = m00 x + m01 y + m02 z + m03, accumulated left to right in f32"""
    M = T.astype(np.float32)
    p = np.ascontiguousarray(points, np.float32)
    out = p.copy()
    for r in range(3):
        out[:, r] = ((M[r, 0] * p[:, 0] + M[r, 1] * p[:, 1]) + M[r, 2] * p[:, 2]) + M[r, 3]
    return out


def process(points, points_attr, image_dict, image_stream_dict, image_param, rtk_dict, imu_list, timestamp):
    s = _state
    if s is None or s.engine is None:
        raise RuntimeError("init_slam / setup_slam first")
    e = s.engine
    imu = np.asarray(imu_list, np.float64).reshape(-1, 7)
    for row in imu:  # numpy_to_imu + HDL_FastLIO::feedImuData
        e.fastlio_imu_enqueue(row[0] / 1000000.0, row[1:4] / 180.0 * np.pi, row[4:7] * 9.81)
    name = s.lidar if s.lidar in points else next(iter(points))
    attr = points_attr[name]
    cloud = _transform_f32(points[name], s.static)                     # preprocessPoints: lidar -> INS frame
    stamp = np.asarray(attr["points_attr"], np.float32)[:, 0].astype(np.uint32)  # pointcloud_attr[i].stamp = ref_attr(i, 0)
    e.fastlio_pcl_enqueue(cloud, stamp, int(attr["timestamp"]) / 1000000.0)
    rc = capi.MAIN_IDLE
    for _ in range(4):  # runLio(): call fastlio_main until it has consumed the scan
        rc = e.fastlio_main()
        if rc != capi.MAIN_IDLE:
            break
    odom_s, odom_e = e.fastlio_odometry()
    Ti = np.linalg.inv(s.T_imu_ins)
    odom_s, odom_e = Ti @ odom_s @ s.T_imu_ins, Ti @ odom_e @ s.T_imu_ins      # fastlio.cpp:269-270
    pose = dict(latitude=0.0, longitude=0.0, altitude=0.0, heading=0.0, pitch=0.0, roll=0.0, Ve=0, Vn=0, Vu=0, Status=0, state="",
                timestamp=int(timestamp), odom_matrix=odom_s.astype(np.float32))  # getPose returns odom2map * odom_start
    return dict(frame_start_timestamp=int(timestamp), pose=pose, slam_valid=True, _odom_end=odom_e, _rc=rc)


def _not_on_the_path(name):
    def f(*a, **k):
        raise NotImplementedError(f"slam_wrapper.{name}: graph back end / map management / GNSS are outside the hot path built here (DESIGN.md section 7)")
    f.__name__ = name
    return f


for _n in ("set_camera_param", "set_ins_config", "set_destination", "set_map_origin", "get_map_origin", "set_mapping_ground_constraint",
           "get_mapping_ground_constraint", "set_mapping_constraint", "set_map_colouration", "get_graph_map", "update_odom", "get_graph_status",
           "get_graph_edges", "get_graph_meta", "get_color_map", "run_graph_optimization", "run_robust_graph_optimization", "merge_map",
           "set_init_pose", "get_estimate_pose", "del_graph_vertex", "add_graph_edge", "del_graph_edge", "set_graph_vertex_fix", "pointcloud_align",
           "set_export_map_config", "export_points", "dump_map_points", "dump_odometry", "dump_graph", "dump_keyframe"):
    globals()[_n] = _not_on_the_path(_n)
