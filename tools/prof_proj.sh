cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
cat > /tmp/pp.py <<'PY'
import sys, torch, time
sys.path.insert(0, '.')
from taichi_splatting_amd import RasterConfig
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
from taichi_splatting_amd.perspective.projection import project_to_image
dev = torch.device('cuda', 0)
torch.manual_seed(0)
cam = random_camera(image_size=(1024, 768))
g = random_3d_gaussians(2_000_000, cam, margin=0.5).to(dev)
cam = cam.to(device=dev)
cfg = RasterConfig()
with torch.no_grad():
  for _ in range(5): out = project_to_image(g, cam, cfg)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(50): out = project_to_image(g, cam, cfg)
  torch.cuda.synchronize()
  print("ms per call", (time.perf_counter() - t0) / 50 * 1e3, "visible", out[0].shape[0])
PY
python /tmp/pp.py
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_proj -o pp -- python /tmp/pp.py > /dev/null 2>&1
db=$(find gpurun_out/prof_proj -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --steps 55 | head -16
python tools/trace_timeline.py $db | tail -12
