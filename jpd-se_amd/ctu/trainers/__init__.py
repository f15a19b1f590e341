"""Trainer registry with the reference's lookup contract (ctu/trainers/__init__.py:5-20):
`opt.model == NAME` resolves to the BaseTrainer subclass `<NAME>Trainer` (case-insensitive)."""
import importlib

from ctu.trainers.base_trainer import BaseTrainer


def get_trainer(opt):
  module_name = 'ctu.trainers.' + opt.model + '_trainer'
  module = importlib.import_module(module_name)
  wanted = (opt.model + 'trainer').lower()
  for name, cls in vars(module).items():
    if name.lower() == wanted and isinstance(cls, type) and issubclass(cls, BaseTrainer):
      return cls
  raise ValueError('In {}.py, there should be a subclass of BaseTrainer with class name that matches {} '
                   'in lowercase.'.format(module_name, wanted))
