"""-m gpu: end-to-end training loop (reference README demo examples/fit_image_gaussians.py): rasterize
fwd/bwd + visibility / split heuristics + VisibilityAwareLaProp + split / prune.  The fit must improve
the PSNR substantially and reach the requested number of gaussians."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('antialias', [False, True])
def test_fit_test_card(antialias):
  from taichi_splatting_amd.examples.fit_image_gaussians import fit, test_card, psnr
  ref = test_card(192, 128, torch.device('cuda:0'))
  image, params, history = fit(ref, n=400, iters=240, target=800, seed=0, antialias=antialias)
  first, last = history[0][1], history[-1][1]
  assert all(torch.isfinite(t).all() for t in params.tensors.values())
  assert last > first + 4.0 and last > 19.0, history
  assert 700 <= params.batch_size[0] <= 800, params.batch_size
  assert image.shape == ref.shape and abs(psnr(ref, image) - last) < 1e-3
