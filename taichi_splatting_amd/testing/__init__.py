from .random_data import random_camera, random_3d_gaussians, random_2d_gaussians

__all__ = ['random_camera', 'random_3d_gaussians', 'random_2d_gaussians']
