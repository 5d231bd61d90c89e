from .tile_mapper import map_to_tiles, pad_to_tile

__all__ = ['map_to_tiles', 'pad_to_tile']
