"""Build-container-only: dump the flag table of the REFERENCE option setter
(ctu/models/pix2pixHD_model.py:21-102 `Pix2PixHDModel.modify_commandline_options`) to
tests/golden/option_setter_flags.json (TEST INFRASTRUCTURE; data only: names, defaults, types, choices).

  python -m oracle.make_option_fixture

tests/test_host_abi.py::test_option_setter_is_superset_of_reference checks the product's setter against it: a
reference command line or opt.pkl must parse unchanged on the HIP path.
"""
import argparse
import json
import os

from oracle import _refbridge

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                   'option_setter_flags.json')


def flag_table(setter, train=True):
  parser = argparse.ArgumentParser(add_help=False)
  setter(parser, train)
  rows = []
  for a in parser._actions:
    rows.append(dict(dest=a.dest, flags=list(a.option_strings), default=a.default,
                     type=(a.type.__name__ if a.type is not None else None),
                     choices=(list(a.choices) if a.choices is not None else None),
                     action=type(a).__name__, nargs=a.nargs))
  return rows


def main():
  _networks, RefModel, _RefTrainer = _refbridge.import_reference()
  rows = flag_table(RefModel.modify_commandline_options)
  with open(OUT, 'w') as fh:
    json.dump(dict(source='ctu/models/pix2pixHD_model.py:21-102 (reference option setter), train=True', flags=rows),
              fh, indent=1, sort_keys=True)
  print('wrote %s: %d flags' % (OUT, len(rows)))


if __name__ == '__main__':
  main()
