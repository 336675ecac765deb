"""DESIGN.md section 0a ("Current figures") is GENERATED from the newest round's files under profiles/ (tools/design_figures.py): this test
regenerates the block and requires DESIGN.md to hold exactly that text -- the prose cannot quote a figure the committed profiles do not
contain (round 5's verdict found DESIGN.md:228 still quoting round 2's command-line throughput)."""
import glob
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location('design_figures', os.path.join(ROOT, 'tools', 'design_figures.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_design_current_figures_are_the_committed_profiles():
    tags = sorted({re.match(r'(r\d+)_STAMP\.json', os.path.basename(p)).group(1) for p in glob.glob(os.path.join(ROOT, 'profiles', 'r*_STAMP.json'))})
    newest = tags[-1]
    mod = _tool()
    want = mod.block(newest)
    text = open(os.path.join(ROOT, 'DESIGN.md')).read()
    i, k = text.index(mod.BEGIN), text.index(mod.END) + len(mod.END)
    assert text[i:k] == want, 'DESIGN.md section 0a is not what `python tools/design_figures.py %s --write` produces from profiles/' % newest
    assert '## 0a. Current figures (round %d)' % int(newest[1:]) in text
