from .function import rasterize, rasterize_with_tiles, RasterOut
from ..data_types import RasterConfig

__all__ = ['rasterize', 'rasterize_with_tiles', 'RasterConfig', 'RasterOut']
