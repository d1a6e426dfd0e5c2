import sys, os, math
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch
from oracle import RasterOracle
from oracle import torch_ref as tr
from util import *
from manus_amd.synthetic import make_scene, camera_table
from manus_amd import ops
from manus_amd.rasterizer import rasterize_views
DEV='cuda:0'
sc = make_scene(n_gaussians=3000, kind="hand", seed=5, grid_res=24, n_cameras=1, width=96, height=64, cam_radius=0.5, sigma_range=(2e-3, 8e-3), device="cpu")
c = sc["cameras"][0]
Pc = {k: v.clone().requires_grad_(True) for k, v in sc["params"].items()}
o = tr.hand_forward(Pc, sc["grid"], sc["grid_center"], sc["grid_scale"], sc["posed"][0], sc["rest"], torch.tensor(c["camera_center"], dtype=torch.float32))
a = cam_args(c)
ro = RasterOracle(a["W"], a["H"], a["tanfovx"], a["tanfovy"], a["view"], a["proj"], o["posed_xyz"].detach().numpy(), o["posed_cov"].detach().numpy(), o["colors"].detach().numpy(), o["opacity"].detach().numpy()[:, 0], np.ones(3, np.float32))
g = torch.randn((64, 96, 3), generator=torch.Generator().manual_seed(0))
b = ro.backward(np.transpose(g.numpy(), (2, 0, 1)))
ct = camera_table(sc["cameras"], DEV)
P = {k: v.clone().to(DEV).requires_grad_(True) for k, v in sc["params"].items()}
w = ops.skin_weights(P['_xyz'], sc['grid'].to(DEV), sc['grid_center'].to(DEV), sc['grid_scale'].to(DEV))
px,pc_,tf = ops.lbs_cov(P['_xyz'],P['_scaling'],P['_rotation'],w,sc['transforms'].to(DEV))
col = ops.sh_colors(torch.cat([P['_features_dc'],P['_features_rest']],1), P['_xyz'], tf, ct)
opac = torch.sigmoid(P['_opacity'])
for t in (px,pc_,col,opac): t.retain_grad()
m2d=torch.zeros((1,3000,3),device=DEV,requires_grad=True)
img,_=rasterize_views(ct,px,m2d,col,opac,pc_,torch.ones(3,device=DEV),96,64)
img.backward(g.permute(2,0,1)[None].to(DEV))
print('fwd diffs', max_rel_err(px[0].detach().cpu().numpy(), o['posed_xyz'].detach().numpy()), max_rel_err(pc_[0].detach().cpu().numpy(), o['posed_cov'].detach().numpy()), max_rel_err(col[0].detach().cpu().numpy(), o['colors'].detach().numpy()))
for k,t in (('means3D',px),('cov3D',pc_),('colors',col)):
    d = t.grad[0].cpu().numpy()-b[k]
    print('rastergrad', k, max_rel_err(t.grad[0].cpu().numpy(), b[k]), np.linalg.norm(d)/np.linalg.norm(b[k]), 'n bad rows', (np.abs(d).max(1) > 1e-4*np.abs(b[k]).max()).sum())
(o["posed_xyz"] * torch.tensor(b["means3D"])).sum().backward(retain_graph=True)
(o["posed_cov"] * torch.tensor(b["cov3D"])).sum().backward(retain_graph=True)
(o["colors"] * torch.tensor(b["colors"])).sum().backward(retain_graph=True)
(o["opacity"][:, 0] * torch.tensor(b["opacity"])).sum().backward()
for k in P:
    a_, b_ = P[k].grad.cpu().numpy().astype(np.float64), Pc[k].grad.numpy().astype(np.float64)
    d = np.abs(a_-b_).reshape(3000,-1).max(1)
    print(k, max_rel_err(a_,b_), np.linalg.norm(a_-b_)/np.linalg.norm(b_), 'rows>1e-5*max', (d>1e-5*np.abs(b_).max()).sum(), 'top', np.sort(d)[-5:])
# the cov gradient magnitude and conditioning
print('cov grad max', np.abs(b['cov3D']).max(), 'cov max', np.abs(o['posed_cov'].detach().numpy()).max())
