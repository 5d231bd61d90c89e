"""Alias module: the reference exposes ``evaluate_sh_at`` from ``indexed_spherical_harmonics``."""
from .spherical_harmonics import evaluate_sh_at, check_sh_degree

__all__ = ['evaluate_sh_at', 'check_sh_degree']
