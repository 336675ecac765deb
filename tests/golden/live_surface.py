#!/usr/bin/env python3
"""The reference's nn.Module class surface as JSON (build container only: imports /root/reference through make_goldens.py's shims):
for every model class of jTransUP/models -- constructor arguments and public methods with their argument names.
tests/test_cli_dropin_live.py compares it with this build's mirror of the interface."""
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[sys.argv.index('--ref') + 1] if '--ref' in sys.argv else '/root/reference'
sys.argv = [sys.argv[0], '--ref', REF, '--out', os.devnull]
sys.path.insert(0, HERE)
import make_goldens as MG                                        # noqa: E402,F401  (puts the reference on sys.path, installs the shims)
import torch.nn as nn                                            # noqa: E402
import importlib                                                 # noqa: E402

out = {}
for mod in ('bprmf', 'fm', 'transE', 'transH', 'transR', 'transUP', 'jTransUP', 'CKE', 'CFKG', 'cofm'):
    m = importlib.import_module('jTransUP.models.' + mod)
    for cname, cls in inspect.getmembers(m, inspect.isclass):
        if cls.__module__ != m.__name__ or not issubclass(cls, nn.Module):
            continue
        methods = {}
        for name, fn in inspect.getmembers(cls, inspect.isfunction):
            if fn.__qualname__.split('.')[0] != cname or (name.startswith('_') and name != '__init__'):
                continue                                         # the class's own public methods (+ its constructor)
            methods[name] = list(inspect.signature(fn).parameters)
        out['%s.%s' % (mod, cname)] = methods
print(json.dumps(out, sort_keys=True))
