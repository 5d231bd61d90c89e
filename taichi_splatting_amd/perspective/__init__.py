from .projection import project_to_image
from .params import CameraParams
from . import projection

__all__ = ['project_to_image', 'CameraParams', 'projection']
