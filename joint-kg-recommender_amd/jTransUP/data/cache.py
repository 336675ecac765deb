"""Binary cache of parsed dataset files.

The reference re-parses its tab-separated text files (ml1m: ~1 M rating lines, ~0.4 M triples) into python lists and
dict-of-sets on every start.  `cached(path, parser)` keeps the parser's result as a pickle under `<dir>/.ktup_cache/`,
keyed by the source file's size and mtime, and falls back to parsing when the cache is missing, stale, unreadable or the
directory is not writable.  KTUP_DATA_CACHE=0 disables it; KTUP_DATA_CACHE=<dir> redirects the cache files."""
import hashlib
import os
import pickle

FORMAT = 1


def _cache_file(path):
    where = os.environ.get('KTUP_DATA_CACHE', '1')
    if where == '0':
        return None
    full = os.path.abspath(path)
    if where in ('1', ''):
        folder = os.path.join(os.path.dirname(full), '.ktup_cache')
    else:
        folder = where
    tag = hashlib.sha1(full.encode('utf-8')).hexdigest()[:12]
    return os.path.join(folder, '%s.%s.pkl' % (os.path.basename(full), tag))


def cached(path, parser):
    """parser(path) -> picklable object."""
    target = _cache_file(path)
    if target is None:
        return parser(path)
    st = os.stat(path)
    stamp = (FORMAT, st.st_size, st.st_mtime_ns, parser.__module__ + '.' + parser.__name__)
    try:
        with open(target, 'rb') as f:
            have, value = pickle.load(f)
        if have == stamp:
            return value
    except Exception:      # noqa: BLE001 -- missing / stale / truncated / unpicklable: parse
        pass
    value = parser(path)
    try:
        os.makedirs(os.path.dirname(target), exist_ok=True)
        tmp = '%s.%d.tmp' % (target, os.getpid())
        with open(tmp, 'wb') as f:
            pickle.dump((stamp, value), f, protocol=4)
        os.replace(tmp, target)
    except OSError:
        pass               # read-only dataset directory: no cache
    return value
