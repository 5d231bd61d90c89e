"""``ParameterClass``: a group of mixed parameter / non-parameter tensors sharing a leading point
dimension, with optimizer state kept in step with them under filtering and appending.

Same behaviour and method names as reference ``optim/parameter_class.py:12-267``; tensordict is not
available on the target image, so the tensor collection is a plain ``dict`` of name -> tensor.
"""
from __future__ import annotations

import copy
from typing import Dict, Iterable, Optional, Tuple

import torch
import torch.optim as optim


def _map(d: Dict[str, torch.Tensor], f):
  return {k: f(v) for k, v in d.items()}


def _map_nested(d: Dict[str, Dict[str, torch.Tensor]], f):
  return {k: _map(v, f) for k, v in d.items()}


def as_parameters(tensors: Dict[str, torch.Tensor], keys: Iterable[str]):
  keys = set(keys)
  return {k: torch.nn.Parameter(x.detach(), requires_grad=True) if k in keys else x for k, x in tensors.items()}


def replace_dict(d, **kwargs):
  d = copy.copy(d)
  d.update(kwargs)
  return d


class ParameterClass:
  """
  Parameters:
    tensors: dict name -> tensor, all with the same first dimension (points)
    parameter_groups: dict name -> optimizer group options for the tensors to optimise
    optimizer_state: optional (tensor_state, other_state) to insert into the optimizer
    optimizer: optimizer class; extra keyword arguments are passed to it
  """

  def __init__(self, tensors: Dict[str, torch.Tensor], parameter_groups: Dict[str, Dict],
               optimizer_state: Optional[Tuple[Dict, Dict]] = None, optimizer=optim.Optimizer, **optim_kwargs):
    sizes = {v.shape[0] for v in tensors.values()}
    assert len(sizes) == 1 and next(iter(sizes)) > 0, f"tensors must share a non-empty first dimension, got {sizes}"
    for k in parameter_groups:
      assert k in tensors, f"parameter group {k} not in tensors {list(tensors)}"

    self.tensors = as_parameters(dict(tensors), parameter_groups.keys())
    param_groups = [dict(params=[self.tensors[name]], name=name, **group)
                    for name, group in parameter_groups.items()]
    self.optimizer = optimizer(param_groups, **optim_kwargs)
    self.optim_kwargs = optim_kwargs

    if optimizer_state is not None:
      tensor_state, other_state = optimizer_state
      for k in tensor_state.keys():
        assert k in self.tensors, f"state parameter {k} not in {list(self.tensors)}"
        self.optimizer.state[self.tensors[k]] = {**tensor_state[k], **other_state.get(k, {})}

  # -- groups -------------------------------------------------------------------------------------
  @property
  def parameter_groups(self):
    return {group['name']: {k: v for k, v in group.items() if k not in ['params', 'name']}
            for group in self.optimizer.param_groups}

  @property
  def learning_rates(self):
    return {group['name']: group['lr'] for group in self.optimizer.param_groups}

  def set_learning_rate(self, **kwargs: float):
    learning_rates = replace_dict(self.learning_rates, **kwargs)
    for group in self.optimizer.param_groups:
      group['lr'] = learning_rates[group['name']]
    return self

  def update_group(self, name: str, **kwargs):
    for group in self.optimizer.param_groups:
      if group['name'] == name:
        group.update(kwargs)
        return
    raise ValueError(f"Group {name} not found in optimizer")

  def update_groups(self, **kwargs):
    for name, params in kwargs.items():
      self.update_group(name, **params)
    return {name: params['lr'] for name, params in kwargs.items()}

  # -- state --------------------------------------------------------------------------------------
  def _get_state(self, f):
    return {name: f(self.optimizer.state[param]) for name, param in self.tensors.items()
            if param in self.optimizer.state}

  @property
  def tensor_state(self) -> Dict[str, Dict[str, torch.Tensor]]:
    return self._get_state(lambda state: {k: v for k, v in state.items() if torch.is_tensor(v)})

  @property
  def other_state(self) -> Dict:
    return self._get_state(lambda state: {k: v for k, v in state.items() if not torch.is_tensor(v)})

  @property
  def optimizer_state(self):
    return self.tensor_state, self.other_state

  def state_dict(self) -> Dict:
    return {'tensors': _map(self.tensors, lambda t: t.detach()),
            'optimizer': (self.tensor_state, self.other_state),
            'parameter_groups': self.parameter_groups}

  @staticmethod
  def from_state_dict(state: dict, device, optimizer=optim.Adam, **optim_kwargs) -> 'ParameterClass':
    tensor_state, other_state = state['optimizer']
    return ParameterClass(_map(state['tensors'], lambda t: t.to(device)),
                          parameter_groups=state['parameter_groups'],
                          optimizer_state=(_map_nested(tensor_state, lambda t: t.to(device)), other_state),
                          optimizer=optimizer, **optim_kwargs)

  # -- optimizer passthrough ----------------------------------------------------------------------
  def zero_grad(self):
    self.optimizer.zero_grad()

  def step(self, **kwargs):
    self.optimizer.step(**kwargs)

  # -- collection API -----------------------------------------------------------------------------
  def keys(self):
    return self.tensors.keys()

  def optimized_keys(self):
    return self.parameter_groups.keys()

  def items(self):
    return self.tensors.items()

  def __getattr__(self, name):
    tensors = self.__dict__.get('tensors', {})
    if name in tensors:
      return tensors[name]
    raise AttributeError(name)

  def _rebuild(self, tensors, tensor_state):
    return ParameterClass(tensors, self.parameter_groups, optimizer_state=(tensor_state, self.other_state),
                          optimizer=type(self.optimizer), **self.optim_kwargs)

  def modify_tensors(self, f):
    return self._rebuild(f(_map(self.tensors, lambda t: t.detach())),
                         {k: f(v) for k, v in self.tensor_state.items()})

  def apply(self, f):
    return self.modify_tensors(lambda d: _map(d, f))

  def to(self, device):
    return self.apply(lambda t: t.to(device))

  def replace(self, **kwargs):
    tensors = _map(self.tensors, lambda t: t.detach())
    tensors.update(kwargs)
    return self._rebuild(tensors, self.tensor_state)

  def detach(self) -> Dict[str, torch.Tensor]:
    return _map(self.tensors, lambda t: t.detach())

  def to_dict(self):
    return dict(self.tensors)

  @property
  def batch_size(self):
    return torch.Size((next(iter(self.tensors.values())).shape[0],))

  @property
  def batch_dims(self):
    return 1

  def __getitem__(self, idx):
    if isinstance(idx, str):
      return self.tensors[idx]
    if idx.dtype == torch.bool:
      idx = idx.nonzero().squeeze(1)
    return self._rebuild(_map(self.tensors, lambda t: t.detach()[idx]),
                         _map_nested(self.tensor_state, lambda t: t[idx]))

  def append_tensors(self, tensors: Dict[str, torch.Tensor], tensor_state: Optional[Dict] = None):
    assert set(tensors.keys()) == set(self.tensors.keys()), f"{tensors.keys()} != {self.tensors.keys()}"
    n = next(iter(tensors.values())).shape[0]
    if tensor_state is None:
      tensor_state = _map_nested(self.tensor_state, lambda t: t.new_zeros((n, *t.shape[1:])))
    joined = {k: torch.cat([self.tensors[k].detach(), tensors[k].to(self.tensors[k].device)]) for k in self.tensors}
    state = {k: {s: torch.cat([v[s], tensor_state[k][s]]) for s in v} for k, v in self.tensor_state.items()}
    return self._rebuild(joined, state)

  def append(self, params: 'ParameterClass'):
    return self.append_tensors(_map(params.tensors, lambda t: t.detach()))
