"""Timing of the full-size skin-weight voxel grid (196 x 142 x 116 voxels) built from the MANO rest mesh on the GPU."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manus_amd import mano_init as MI, dataset as D
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
m = np.load(g + "/mano_rest.npz"); data = {"verts": m["verts"], "weights": m["weights"], "face": m["faces"]}
ds = D.TestDataset(dict(cam_path=g + "/eval_inputs/camera_path.npz", cano_cam_path=g + "/eval_inputs/cano_camera.npz", metadata_path=g + "/eval_inputs/novel_pose.npz"))
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    out = MI.build_voxel_grid(ds.bones_rest, data, res=128, ratio=(1.1, 0.9, 0.65), device="cuda:0")
    torch.cuda.synchronize(); print("build_voxel_grid", tuple(out[3].shape), "%.2f s" % (time.time() - t))
pts = out[2].reshape(-1, 3).cuda(); v = torch.tensor(data["verts"]).cuda(); f = torch.tensor(data["face"].astype(np.int64)).cuda(); r = torch.tensor(data["weights"][:, MI.MANO_TO_OURS]).cuda().contiguous()
for name, fn in (("knn4", lambda: MI.knn_mean_rows(pts, v, r, 4)), ("knn20", lambda: MI.knn_mean_rows(pts[:300000], v, r, 20)), ("sdf", lambda: MI.mesh_sdf(pts, v, f))):
    fn(); torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize(); print(name, "%.1f ms" % ((time.time() - t) * 1e3))
