import sys, time, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'oracle'); sys.path.insert(0,'lidar-slam-detection_amd/python')
import gicp_cases, ref_gicp
from lsd_amd import lio, synth
sc = synth.Scene(half=60.0, n_boxes=30, seed=9)
pa, qa = np.array([0.5, -1.0, 1.8]), synth.quat_from_rotvec([0, 0, 0.2])
pb, qb = np.array([2.0, -0.2, 1.8]), synth.quat_from_rotvec([0.01, -0.02, 0.4])
ra, _ = synth.make_scan(sc, pa, qa, seed=21, max_range=80.0)
rb, _ = synth.make_scan(sc, pb, qb, seed=22, max_range=80.0)
for leaf in (0.3, 0.15):
    tg=gicp_cases._thin(ra[:, :4].astype(np.float32), leaf); sr=gicp_cases._thin(rb[:, :4].astype(np.float32), leaf)
    truth = np.linalg.inv(gicp_cases._pose(pa, qa)) @ gicp_cases._pose(pb, qb)
    guess = truth @ gicp_cases._pose([0.2, -0.1, 0.05], synth.quat_from_rotvec([0.005, 0.01, -0.02]))
    for grid in (0.5,1.0,2.0):
        g=lio.Gicp(grid_resolution=grid,max_points=max(len(tg),len(sr)),k=20)
        g.set_target(tg); g.set_source(sr)
        t0=time.perf_counter(); g.set_target(tg); t1=time.perf_counter(); g.set_source(sr); t2=time.perf_counter()
        for _ in range(3): r=g.linearize(guess)
        t3=time.perf_counter()
        for _ in range(5): r=g.linearize(guess)
        t4=time.perf_counter()
        T,conv,it=g.align(guess); t5=time.perf_counter()
        print(f"leaf {leaf} n_tgt {len(tg)} n_src {len(sr)} grid {grid}: set_target {1e3*(t1-t0):.2f} ms set_source {1e3*(t2-t1):.2f} ms linearize {1e3*(t4-t3)/5:.3f} ms align {1e3*(t5-t4):.2f} ms it {it} conv {conv} err_t {np.abs(T[:3,3]-truth[:3,3]).max():.4f}")
        g.close()
    h=ref_gicp.RefGicp(k=20,max_corr_dist=2.0,num_threads=4)
    t0=time.perf_counter(); h.set_target(tg); t1=time.perf_counter(); h.set_source(sr); t2=time.perf_counter()
    h.linearize(guess); t3=time.perf_counter(); T,it,conv=h.align(guess.astype(np.float32)); t4=time.perf_counter()
    print(f"  reference (4 threads, grid k-NN stand-in): set_target {1e3*(t1-t0):.1f} ms linearize {1e3*(t3-t2):.1f} ms align {1e3*(t4-t3):.1f} ms it {it}")
