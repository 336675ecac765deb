"""Drop-in check of the command-line surface against the reference ITSELF (build container only: reads /root/reference at run
time, nothing of it is copied): every flag the reference defines (jTransUP/models/base.py: name, kind, default, enum choices) is
defined here with the same default, and every `python run_*.py ...` command line of the reference's own shell scripts
(ktup.sh, transup.sh, transe.sh, ...) parses under this build's registry to the values python-gflags would give it."""
import ast
import glob
import os
import re
import shlex

import pytest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'jTransUP')), reason='the reference is present in the build container only')


def _reference_flags():
    tree = ast.parse(open(os.path.join(REF, 'jTransUP', 'models', 'base.py')).read())
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith('DEFINE_'):
            kind = node.func.attr[len('DEFINE_'):]
            name = ast.literal_eval(node.args[0])
            default = ast.literal_eval(node.args[1])
            enum = ast.literal_eval(node.args[2]) if kind == 'enum' else None
            out[name] = (kind, default, enum)
    return out


def test_every_reference_flag_is_defined_with_its_default():
    from jTransUP.models.base import get_flags
    from jTransUP.utils.flags import FLAGS
    get_flags(); FLAGS.reset()
    ref = _reference_flags()
    assert len(ref) == 40
    kinds = {'bool': 'bool', 'boolean': 'bool', 'integer': 'int', 'float': 'float', 'string': 'str', 'enum': 'enum'}
    for name, (kind, default, enum) in ref.items():
        assert FLAGS.is_defined(name), name
        f = FLAGS._flags[name]
        assert f.kind == kinds[kind] or (kind == 'string' and f.kind not in ('bool', 'int', 'float', 'enum')), (name, f.kind, kind)
        assert f.default == default, (name, f.default, default)
        if enum is not None:
            assert set(enum) <= set(f.enum), (name, enum, f.enum)         # (this build adds choices, e.g. model types it also serves)
    FLAGS.reset()


def _script_command_lines():
    for path in sorted(glob.glob(os.path.join(REF, '*.sh'))):
        for line in open(path):
            m = re.search(r'python\s+(run_\w+\.py)\s+(.*)$', line.strip())
            if m:
                yield os.path.basename(path), m.group(1), shlex.split(m.group(2).replace('~', '/home/u'))


def test_the_reference_scripts_command_lines_parse_to_the_same_values():
    from jTransUP.models.base import get_flags
    from jTransUP.utils.flags import FLAGS
    ref = _reference_flags()
    seen = 0
    for script, prog, toks in _script_command_lines():
        if any('$' in t for t in toks):
            toks = [('x.dat' if '$' in t else t) for t in toks]            # ktup_eval.sh's loop variables: any file name
        get_flags(); FLAGS.reset()
        i, expect, rejected = 0, {}, None
        while i < len(toks):                                              # what python-gflags makes of the line
            name = toks[i].lstrip('-')
            if name in ref and ref[name][0] in ('bool', 'boolean'):
                expect[name] = True; i += 1
            elif name.startswith('no') and name[2:] in ref and ref[name[2:]][0] in ('bool', 'boolean'):
                expect[name[2:]] = False; i += 1
            else:
                assert name in ref, (script, name)
                kind = ref[name][0]
                if i + 1 < len(toks) and not (toks[i + 1].startswith('-') and toks[i + 1].lstrip('-') in ref):
                    val = toks[i + 1]; i += 2
                else:                                                     # ktup_eval.sh: `-kg_test_files -l2_lambda 0` (a flag without its value)
                    expect = None; break
                expect[name] = {'integer': int, 'float': float}.get(kind, str)(val)
                if kind == 'enum' and val not in ref[name][2]:
                    rejected = name                                       # ktup.sh / ktup_eval.sh: -model_type cjtransup, a name the
        if expect is None:                                                # reference's own enum no longer has: python-gflags refuses it
            continue                                                      # (python-gflags would take the next flag as the value: a typo in the script)
        if rejected:
            from jTransUP.utils.flags import FlagError
            with pytest.raises(FlagError):
                FLAGS(['prog'] + toks)
            seen += 1
            continue
        rest = FLAGS(['prog'] + toks)
        assert rest == ['prog'], (script, rest)
        for name, val in expect.items():
            assert getattr(FLAGS, name) == val, (script, name, getattr(FLAGS, name), val)
        assert os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'joint-kg-recommender_amd', prog))
        seen += 1
    assert seen >= 10                                                     # bprmf, fm, cofm, cke, ktup, ktup_eval, transe, transh, transr, transup ...
    FLAGS.reset()


IN_SCOPE = ['models/base', 'models/bprmf', 'models/fm', 'models/transE', 'models/transH', 'models/transR', 'models/transUP', 'models/jTransUP',
            'models/CKE', 'models/CFKG', 'models/cofm', 'models/item_recommendation', 'models/knowledge_representation',
            'models/knowledgable_recommendation', 'utils/loss', 'utils/misc', 'utils/trainer', 'utils/evaluation', 'utils/data',
            'data/load_rating_data', 'data/load_triple_data', 'data/load_kg_rating_data']
# the reference's multiprocessing worker classes behind evalRecProcess / evalKGProcess: this build ranks on the device, there is no
# process fan-out to mirror
NOT_MIRRORED = {'utils/misc': {'MyEvalKGProcess', 'MyEvalRecProcess'}}


def _surface(path):
    """Top-level functions and classes' public methods of a source file: name -> (positional argument names, number with defaults,
    takes *args / **kwargs); classes also list their bases.  Read with ast: nothing is imported or executed."""
    tree = ast.parse(open(path).read())
    sig = lambda fn: ([a.arg for a in fn.args.args], len(fn.args.defaults), fn.args.vararg is not None or fn.args.kwarg is not None)
    funcs, classes = {}, {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            funcs[node.name] = sig(node)
        elif isinstance(node, ast.ClassDef):
            methods = {m.name: sig(m) for m in node.body if isinstance(m, ast.FunctionDef) and (not m.name.startswith('_') or m.name == '__init__')}
            classes[node.name] = methods
    return funcs, classes


def test_module_surface_matches_the_reference():
    """Every in-scope module of the reference's package (SURVEY.md section 2 minus the rows DESIGN.md section 0 puts out of scope: transD,
    log parsers, preprocessing, plotting, the visdom UI): each of its top-level functions and each public method of its classes exists
    here under the same name with the same leading argument names in the same order; whatever this build adds to a signature comes
    after them and is optional.  Methods may live in a base class here (GradToggle, GateHelpers): resolved through the imported class."""
    import importlib
    import inspect
    ours_root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'joint-kg-recommender_amd', 'jTransUP')
    checked = 0
    for mod in IN_SCOPE:
        rfuncs, rclasses = _surface(os.path.join(REF, 'jTransUP', mod + '.py'))
        ofuncs, oclasses = _surface(os.path.join(ours_root, mod + '.py'))
        for name, (args, ndef, star) in rfuncs.items():
            assert name in ofuncs, (mod, name)
            oargs, ondef, ostar = ofuncs[name]
            assert oargs[:len(args)] == args, (mod, name, oargs, args)
            assert ostar or ondef >= len(oargs) - len(args), (mod, name, oargs, args)      # the extras are optional
            checked += 1
        for cname, methods in rclasses.items():
            if cname in NOT_MIRRORED.get(mod, ()):
                continue
            assert cname in oclasses, (mod, cname)
            cls = getattr(importlib.import_module('jTransUP.' + mod.replace('/', '.')), cname)
            for name, (args, ndef, star) in methods.items():
                assert hasattr(cls, name), (mod, cname, name)
                p = inspect.signature(getattr(cls, name)).parameters
                oargs = [k for k, v in p.items() if v.kind in (v.POSITIONAL_ONLY, v.POSITIONAL_OR_KEYWORD)]
                assert oargs[:len(args)] == args, (mod, cname, name, oargs, args)
                assert all(p[k].default is not inspect.Parameter.empty for k in oargs[len(args):]), (mod, cname, name, oargs, args)
                checked += 1
    assert checked >= 150


def test_host_side_functions_behave_like_the_reference():
    """tests/golden/live_host.py once against the reference's package and once against this build's (two subprocesses: both are called
    jTransUP): the loaders on the same synthetic dataset files, the train / eval iterators and negative samplers under the same
    random.seed (the same draws in the same order), the metric helpers -- equal JSON documents, and the same lines on stdout."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    env.pop('PYTHONPATH', None)
    docs = []
    for pkg in (REF, os.path.join(root, 'joint-kg-recommender_amd')):
        out = subprocess.run([sys.executable, os.path.join(here, 'golden', 'live_host.py'), '--pkg', pkg], capture_output=True, text=True,
                             timeout=120, env=env, cwd=root)
        assert out.returncode == 0, (pkg, out.stderr[-2000:])
        docs.append(out.stdout.strip().splitlines())
    assert docs[0][:-1] == docs[1][:-1]                                   # what the loaders print
    ref, ours = json.loads(docs[0][-1]), json.loads(docs[1][-1])
    assert sorted(ref) == sorted(ours) and len(ref) == 45
    for key in ref:
        assert ref[key] == ours[key], key
