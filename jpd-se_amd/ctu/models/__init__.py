"""Model registry with the reference's lookup contract (ctu/models/__init__.py:10-44):
`--model NAME` resolves to class `<NAME>Model` (case-insensitive) in `ctu.models.<NAME>_model`."""
import importlib

import torch


def find_model_using_name(model_name):
  module = importlib.import_module('ctu.models.' + model_name + '_model')
  wanted = (model_name.replace('_', '') + 'model').lower()
  for name, cls in vars(module).items():
    if name.lower() == wanted and isinstance(cls, type) and issubclass(cls, torch.nn.Module):
      return cls
  raise ValueError('In ctu/models/%s_model.py there should be a torch.nn.Module subclass named %s '
                   '(case-insensitive)' % (model_name, wanted))


def get_option_setter(model_name):
  return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
  instance = find_model_using_name(opt.model)(opt)
  print('model [%s] was created' % type(instance).__name__)
  return instance
