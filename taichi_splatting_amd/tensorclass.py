"""Minimal stand-in for ``tensordict.TensorClass``.

The reference containers (``Gaussians3D``, ``Gaussians2D``, ``RenderedPoints``: reference
``data_types.py:57``, ``data_types.py:122``, ``rendering.py:27``) derive from tensordict's
``TensorClass``.  tensordict is not available on the target image, so this module supplies
the small subset of behaviour the hot path and its callers rely on:
keyword construction with ``batch_size``, field access, ``.to()/.cuda()/.cpu()``,
``.requires_grad_()``, ``.detach()``, ``.apply()``, ``.replace()``, index/mask ``__getitem__``,
``to_dict()`` and ``cat``.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Tuple, get_type_hints

import torch


class TensorClass:
  """Base class: subclasses declare fields as class annotations.

  Fields annotated ``Optional[...]`` (with a class-level default of ``None``) may be omitted.
  Every non-None tensor field must share the leading ``batch_size`` dimensions.
  """

  _fields: Tuple[str, ...] = ()

  def __init_subclass__(cls, **kwargs):
    super().__init_subclass__(**kwargs)
    hints = {}
    for klass in reversed(cls.__mro__):
      hints.update(getattr(klass, '__annotations__', {}))
    cls._fields = tuple(k for k in hints
                        if not k.startswith('__') and k not in ('_fields', 'batch_size'))

  def __init__(self, batch_size=None, **kwargs):
    unknown = set(kwargs) - set(self._fields)
    if unknown:
      raise TypeError(f"{type(self).__name__}: unknown fields {sorted(unknown)}")

    for name in self._fields:
      if name in kwargs:
        value = kwargs[name]
      elif hasattr(type(self), name) and not isinstance(getattr(type(self), name), property):
        value = getattr(type(self), name)   # class-level default (None)
      else:
        raise TypeError(f"{type(self).__name__}: missing field '{name}'")
      object.__setattr__(self, name, value)

    if batch_size is None:
      first = next((getattr(self, f) for f in self._fields
                    if isinstance(getattr(self, f), torch.Tensor)), None)
      batch_size = (first.shape[0],) if first is not None else ()
    if isinstance(batch_size, int):
      batch_size = (batch_size,)
    object.__setattr__(self, 'batch_size', torch.Size(tuple(int(b) for b in batch_size)))

    nb = len(self.batch_size)
    for name in self._fields:
      v = getattr(self, name)
      if isinstance(v, torch.Tensor):
        assert tuple(v.shape[:nb]) == tuple(self.batch_size), \
          f"{type(self).__name__}.{name}: shape {tuple(v.shape)} does not start with batch_size {tuple(self.batch_size)}"

    post = getattr(self, '__post_init__', None)
    if post is not None:
      post()

  # -- functional helpers -------------------------------------------------------------------
  def _map(self, f: Callable[[Any], Any], batch_size=None):
    def g(v):
      if isinstance(v, (torch.Tensor, TensorClass)):
        return f(v)
      return v
    out = {name: g(getattr(self, name)) for name in self._fields}
    return self._rebuild(out, batch_size)

  def _rebuild(self, values: Dict[str, Any], batch_size=None):
    obj = object.__new__(type(self))
    for name in self._fields:
      object.__setattr__(obj, name, values[name])
    object.__setattr__(obj, 'batch_size',
                       torch.Size(batch_size) if batch_size is not None else self.batch_size)
    return obj

  def apply(self, f, batch_size=None):
    return self._map(f, batch_size)

  def replace(self, **kwargs):
    values = self.to_dict()
    values.update(kwargs)
    batch = kwargs.pop('batch_size', None)
    values.pop('batch_size', None)
    return type(self)(**values, batch_size=batch if batch is not None else self.batch_size)

  def to_dict(self) -> Dict[str, Any]:
    return {name: getattr(self, name) for name in self._fields}

  def to_tensordict(self):
    return self.to_dict()

  @classmethod
  def from_tensordict(cls, d):
    return cls(**dict(d))

  def keys(self):
    return list(self._fields)

  def items(self):
    return self.to_dict().items()

  # -- torch-like API -----------------------------------------------------------------------
  def to(self, *args, **kwargs):
    def conv(v):
      if isinstance(v, torch.Tensor) and not v.is_floating_point():
        # dtype conversions only apply to floating point fields (index fields keep their type)
        kw = {k: x for k, x in kwargs.items() if k != 'dtype'}
        a = tuple(x for x in args if not isinstance(x, torch.dtype))
        return v.to(*a, **kw) if (a or kw) else v
      return v.to(*args, **kwargs)
    return self._map(conv)

  def cuda(self, device=None):
    return self._map(lambda v: v.cuda(device))

  def cpu(self):
    return self._map(lambda v: v.cpu())

  def detach(self):
    return self._map(lambda v: v.detach())

  def clone(self):
    return self._map(lambda v: v.clone())

  def contiguous(self):
    return self._map(lambda v: v.contiguous())

  def requires_grad_(self, requires_grad: bool = True):
    for name in self._fields:
      v = getattr(self, name)
      if isinstance(v, torch.Tensor) and v.is_floating_point():
        v.requires_grad_(requires_grad)
      elif isinstance(v, TensorClass):
        v.requires_grad_(requires_grad)
    return self

  @property
  def device(self):
    for name in self._fields:
      v = getattr(self, name)
      if isinstance(v, torch.Tensor):
        return v.device
    return torch.device('cpu')

  @property
  def shape(self):
    return self.batch_size

  def __len__(self):
    return self.batch_size[0] if len(self.batch_size) > 0 else 0

  def __getitem__(self, idx):
    probe = None
    for name in self._fields:
      v = getattr(self, name)
      if isinstance(v, torch.Tensor):
        probe = v[idx]
        break
    nb = len(self.batch_size)
    new_batch = None
    if probe is not None:
      ref = getattr(self, name)
      trailing = ref.ndim - nb
      new_batch = tuple(probe.shape[:probe.ndim - trailing])
    return self._map(lambda v: v[idx], new_batch)

  def __setattr__(self, name, value):
    # fields are plain attributes; allow updating (e.g. ``gaussians.feature = ...``)
    object.__setattr__(self, name, value)

  def __repr__(self):
    parts = []
    for name in self._fields:
      v = getattr(self, name)
      if isinstance(v, torch.Tensor):
        parts.append(f"{name}=Tensor{tuple(v.shape)}:{str(v.dtype).replace('torch.', '')}")
      else:
        parts.append(f"{name}={v!r}")
    return f"{type(self).__name__}({', '.join(parts)}, batch_size={tuple(self.batch_size)})"

  @classmethod
  def cat(cls, items, dim=0):
    assert len(items) > 0
    first = items[0]
    values = {}
    for name in first._fields:
      vs = [getattr(i, name) for i in items]
      if isinstance(vs[0], torch.Tensor):
        values[name] = torch.cat(vs, dim=dim)
      elif isinstance(vs[0], TensorClass):
        values[name] = type(vs[0]).cat(vs, dim=dim)
      else:
        values[name] = vs[0]
    batch = list(first.batch_size)
    batch[dim] = sum(i.batch_size[dim] for i in items)
    return cls(**values, batch_size=tuple(batch))
