"""Visibility-aware sparse optimisers (reference ``optim/__init__.py``): the step that consumes the
render path's outputs (gradients + visibility of the points in view) each iteration."""
from .parameter_class import ParameterClass
from .fractional import FractionalAdam, FractionalLaProp, SparseAdam, SparseLaProp, restore_grad
from .visibility_aware import VisibilityAwareAdam, VisibilityAwareLaProp, VisibilityOptimizer

__all__ = ['ParameterClass',
           'FractionalAdam', 'FractionalLaProp',
           'SparseAdam', 'SparseLaProp',
           'VisibilityAwareAdam', 'VisibilityAwareLaProp',
           'VisibilityOptimizer',
           'restore_grad']
