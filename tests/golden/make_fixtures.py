"""Generates the golden fixtures under tests/golden/ from the REFERENCE's own pure-torch code.

Runs ONLY in the build container (needs /root/reference); nothing here is imported by the tests
or shipped to the GPU box — the tests read the committed ``.pt`` files.  The reference's
``torch_lib`` (projection, SH, ndc depth) and ``tests/random_data.py`` generators are loaded by
path with stub modules standing in for the packages that are not installed (beartype,
tensordict-based containers), as described in SURVEY.md appendix C.

    python tests/golden/make_fixtures.py
"""
import importlib.util
import sys
import types
import typing
from pathlib import Path

import torch

REF = '/root/reference/taichi_splatting'
OUT = Path(__file__).resolve().parent


def load_reference():
  bt = types.ModuleType('beartype')
  bt.beartype = lambda f=None, **k: f if f else (lambda g: g)
  btt = types.ModuleType('beartype.typing')
  btt.__dict__.update({k: getattr(typing, k) for k in dir(typing) if not k.startswith('_')})
  bt.typing = btt
  sys.modules['beartype'] = bt
  sys.modules['beartype.typing'] = btt

  def pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m

  def load(name, file):
    spec = importlib.util.spec_from_file_location(name, file)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m

  pkg('taichi_splatting', REF)
  pkg('taichi_splatting.torch_lib', REF + '/torch_lib')
  persp = pkg('taichi_splatting.perspective', REF + '/perspective')
  persp.CameraParams = load('taichi_splatting.perspective.params', REF + '/perspective/params.py').CameraParams

  class TC:   # stand-in for the tensordict TensorClass containers
    def __init__(self, batch_size=None, **kw):
      self.__dict__.update(kw)
      self.batch_size = batch_size

    def shape_tensors(self):
      return (self.position, self.log_scaling, self.rotation, self.alpha_logit)

  dt = types.ModuleType('taichi_splatting.data_types')
  dt.Gaussians3D = dt.Gaussians2D = TC
  dt.RasterConfig = object
  sys.modules['taichi_splatting.data_types'] = dt
  tq = types.ModuleType('taichi_splatting.taichi_queue')
  tq.queued = lambda f: f
  sys.modules['taichi_splatting.taichi_queue'] = tq
  load('taichi_splatting.torch_lib.transforms', REF + '/torch_lib/transforms.py')
  load('taichi_splatting.torch_lib.rsh', REF + '/torch_lib/rsh.py')
  sh = load('taichi_splatting.torch_lib.spherical_harmonics', REF + '/torch_lib/spherical_harmonics.py')
  proj = load('taichi_splatting.torch_lib.projection', REF + '/torch_lib/projection.py')
  pkg('taichi_splatting.tests', REF + '/tests')
  rd = load('taichi_splatting.tests.random_data', REF + '/tests/random_data.py')
  return proj, sh, rd


def eval_with_grad(f, *args):
  """Protocol of the reference's tests/util.py:10-31: loss = sum of means of the float outputs."""
  args = [x.detach().clone().requires_grad_(True) if isinstance(x, torch.Tensor) and x.is_floating_point() else x
          for x in args]
  out = f(*args)
  loss = 0
  outs = out if isinstance(out, tuple) else (out,)
  for o in outs:
    if o.dtype in (torch.float32, torch.float64):
      loss = loss + o.mean()
  loss.backward()
  grads = [a.grad if a.grad is not None else torch.zeros_like(a)
           for a in args if isinstance(a, torch.Tensor) and a.is_floating_point()]
  return outs, grads


def main():
  proj, sh, rd = load_reference()

  # ---- projection: protocol of tests/test_projection.py:22-96 (margin 0.5, scale_factor 0.1) ----
  for seed in range(5):
    torch.manual_seed(seed)
    camera = rd.random_camera()
    n = [200, 37, 256, 1, 120][seed]
    g = rd.random_3d_gaussians(n=n, camera_params=camera, margin=0.5, scale_factor=0.1)
    fix = dict(seed=seed, n=n, image_size=camera.image_size, depth_range=camera.depth_range,
               blur_cov=0.3 if seed % 2 == 0 else 0.0)
    for dtype, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
      inputs = [t.to(dtype) for t in g.shape_tensors()] + [camera.T_camera_world.to(dtype), camera.projection.to(dtype)]

      def f(position, log_scaling, rotation, alpha_logit, T, P):
        return proj.apply(position, log_scaling, rotation, alpha_logit, T, P, camera.image_size,
                          camera.depth_range, blur_cov=fix['blur_cov'])
      outs, grads = eval_with_grad(f, *inputs)
      fix[tag] = dict(inputs=[t.detach() for t in inputs], points=outs[0].detach(), depth=outs[1].detach(),
                      indexes=outs[2].detach(), grads=[x.detach() for x in grads])
    torch.save(fix, OUT / f'projection_seed{seed}.pt')
    print('projection', seed, n, 'visible', fix['f64']['indexes'].shape[0])

  # ---- SH: protocol of tests/test_spherical_harmonics.py:16-45 (+ degree 0) ----
  for degree in range(4):
    torch.manual_seed(100 + degree)
    n, dim = 96, [3, 1, 2, 3][degree]
    params = torch.rand(n, dim, (degree + 1) ** 2, dtype=torch.float64)
    points = torch.randn(n, 3, dtype=torch.float64)
    camera_pos = torch.randn(3, dtype=torch.float64)
    indexes = torch.randint(0, n, (n // 2,))
    outs, grads = eval_with_grad(lambda p, x, c: sh.evaluate_sh_at(p, x, indexes, c), params, points, camera_pos)
    torch.save(dict(degree=degree, params=params, points=points, camera_pos=camera_pos, indexes=indexes,
                    out=outs[0].detach(), grads=[x.detach() for x in grads]), OUT / f'sh_deg{degree}.pt')
    print('sh', degree)

  # ---- ndc depth ----
  d = torch.tensor([0.1, 0.2, 1.0, 10.0, 57.3, 100.0], dtype=torch.float64)
  torch.save(dict(depth=d, near=0.1, far=100.0, ndc=proj.ndc_depth(d, 0.1, 100.0),
                  inverse=proj.inverse_ndc_depth(proj.ndc_depth(d, 0.1, 100.0), 0.1, 100.0)), OUT / 'ndc.pt')

  # ---- generator streams ----
  for seed in range(2):
    torch.manual_seed(seed)
    cam = rd.random_camera(image_size=(640, 480)) if seed == 0 else rd.random_camera()
    g3 = rd.random_3d_gaussians(64, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.1 * seed)
    g2 = rd.random_2d_gaussians(64, (320, 200), num_channels=3, scale_factor=0.7, alpha_range=(0.2, 0.8), depth_range=(0.1, 50.0))
    torch.save(dict(
      seed=seed,
      camera=dict(projection=cam.projection, T_camera_world=cam.T_camera_world, image_size=cam.image_size,
                  near_plane=cam.near_plane, far_plane=cam.far_plane),
      g3=dict(position=g3.position, log_scaling=g3.log_scaling, rotation=g3.rotation,
              alpha_logit=g3.alpha_logit, feature=g3.feature),
      g2=dict(position=g2.position, depths=g2.depths, log_scaling=g2.log_scaling, rotation=g2.rotation,
              alpha_logit=g2.alpha_logit, feature=g2.feature)), OUT / f'random_data_seed{seed}.pt')
    print('random_data', seed)


if __name__ == '__main__':
  main()
