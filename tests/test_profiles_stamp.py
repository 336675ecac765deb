"""The tracked profiles must be the final library's: tools/profile_summary.py stamps a round's files with a hash of the kernel sources
they were measured on (csrc/*.hip, csrc/*.h, include/ktup_hip.h), and the newest stamp has to match the tree -- a kernel edited after
the last collection fails here until `bash tools/collect_round.sh <tag>` has run again on a GPU box and its output was copied to
profiles/ (round 3's README quoted numbers that only an untracked re-run contained)."""
import glob
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _summary_module():
    spec = importlib.util.spec_from_file_location('profile_summary', os.path.join(ROOT, 'tools', 'profile_summary.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_newest_profiles_were_measured_on_this_tree():
    stamps = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_STAMP.json')))
    assert stamps, 'no profiles/r*_STAMP.json: run tools/collect_round.sh on a GPU box and copy gpurun_out/profiles/* to profiles/'
    rec = json.load(open(stamps[-1]))
    assert rec['kernel_src_sha16'] == _summary_module().source_sha(), \
        '%s was collected on kernel sources %s; the tree has %s -- re-collect' % (os.path.basename(stamps[-1]), rec['kernel_src_sha16'],
                                                                               _summary_module().source_sha())
    tag = rec['tag']
    assert os.path.isfile(os.path.join(ROOT, 'profiles', tag + '_SUMMARY.md'))
    for name in rec['files']:
        assert os.path.isfile(os.path.join(ROOT, 'profiles', name)), name + ' is listed in the stamp but not tracked'
