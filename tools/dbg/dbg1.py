import sys; sys.path.insert(0, '.')
import torch
from taichi_splatting_amd import RasterConfig, frame, render_gaussians
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
from taichi_splatting_amd.mapper.tile_mapper import map_to_tiles_strip
DEV='cuda:0'
torch.manual_seed(5)
cam = random_camera(image_size=(256,256))
g = random_3d_gaussians(30000, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.6)
g = g.replace(feature=(torch.rand(30000, 3, 9) - 0.5) * 0.5).to(DEV); cam = cam.to(device=DEV)
cfg = RasterConfig()
rf = render_gaussians(g, cam, cfg, use_sh=True)
frame.USE_FRAME=False
rl = render_gaussians(g, cam, cfg, use_sh=True)
d = (rf.image - rl.image).abs()
print('max diff', float(d.max()), 'pixels differing', int((d.max(-1).values > 0).sum()))
st = rf.frame
k = int(st.counters()[0]); print('K frame', k, 'cap', st.capacity)
o2p_f = st.overlap_to_point()[:k].long(); ranges_f = st.tile_ranges().view(-1, 2)
idx = rl.points.idx
o2p_l, ranges_l = map_to_tiles_strip(rl.points.gaussians2d, rl.points.depths, cam.image_size, cfg, ndc_range=(cam.near_plane, cam.far_plane))
print('K legacy', o2p_l.shape[0])
print('ranges equal', torch.equal(ranges_f, ranges_l.view(-1,2)))
if k == o2p_l.shape[0]:
  print('o2p equal', torch.equal(o2p_f, idx[o2p_l.long()]))
  bad = (o2p_f != idx[o2p_l.long()]).nonzero().flatten()
  print('first mismatches', bad[:10].tolist())
  if len(bad):
    j = int(bad[0]); print(o2p_f[j-2:j+3].tolist(), idx[o2p_l.long()][j-2:j+3].tolist())
    a, b = int(o2p_f[j]), int(idx[o2p_l.long()][j])
    dep = _ = rf.points  # materialise
    full_depth = st.keep_n[st.layout.depth: st.layout.depth + 4*30000].view(torch.float32)
    print('depths', float(full_depth[a]), float(full_depth[b]))
print('V', idx.shape[0], 'points equal', torch.equal(rf.points.gaussians2d, rl.points.gaussians2d), torch.equal(rf.points.features, rl.points.features))
