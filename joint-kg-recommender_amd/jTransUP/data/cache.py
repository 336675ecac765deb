"""Binary cache of parsed dataset files.

The reference re-parses its tab-separated text files (ml1m: ~1 M rating lines, ~0.4 M triples) into python lists and
dict-of-sets on every start.  `cached(path, parser)` keeps the parsed integers as a plain `.npz` (numpy arrays only,
loaded with allow_pickle=False: a cache file is DATA, never code -- a file dropped into a shared dataset directory cannot
execute anything) under `<dir>/.ktup_cache/`, stamped with the source file's size and mtime, and rebuilds the parser's
structures from the arrays in file order, so lists, dict key order and set contents are exactly what the parser returns.
It falls back to parsing when the cache is missing, stale, unreadable or the directory is not writable.
KTUP_DATA_CACHE=0 disables it; KTUP_DATA_CACHE=<dir> redirects the cache files."""
import hashlib
import os

import numpy as np

FORMAT = 2


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
    return os.path.join(folder, '%s.%s.npz' % (os.path.basename(full), tag))


# ---- codecs: parser result <-> {name: ndarray}.  Keyed by the parser's function name (loadRatings / loadTriples / loadVocab).
def _enc_ratings(value):
    _, pairs, _ = value
    return {'rows': np.asarray(pairs, dtype=np.int64).reshape(-1, 2)}


def _dec_ratings(arrs):
    pairs, by_user = [], {}
    for u, i in arrs['rows'].tolist():
        pairs.append((u, i))
        by_user.setdefault(u, set()).add(i)
    return len(pairs), pairs, by_user


def _enc_triples(value):
    return {'rows': np.asarray(value[1], dtype=np.int64).reshape(-1, 3)}


def _dec_triples(arrs):
    triples, heads, tails = [], {}, {}
    for h, t, r in arrs['rows'].tolist():
        triples.append((h, t, r))
        heads.setdefault((t, r), set()).add(h)
        tails.setdefault((h, r), set()).add(t)
    return len(triples), triples, heads, tails


def _enc_vocab(value):
    keys = list(value.keys())
    if any('\n' in k for k in keys):
        raise ValueError('vocabulary key with a newline')
    return {'keys': np.frombuffer('\n'.join(keys).encode('utf-8'), dtype=np.uint8),
            'vals': np.asarray([value[k] for k in keys], dtype=np.int64), 'n': np.asarray([len(keys)], dtype=np.int64)}


def _dec_vocab(arrs):
    n = int(arrs['n'][0])
    keys = arrs['keys'].tobytes().decode('utf-8').split('\n') if n else []
    vals = arrs['vals'].tolist()
    if len(keys) != n or len(vals) != n:
        raise ValueError('corrupt vocabulary cache')
    return dict(zip(keys, vals))


CODECS = {'loadRatings': (_enc_ratings, _dec_ratings), 'loadTriples': (_enc_triples, _dec_triples),
          'loadVocab': (_enc_vocab, _dec_vocab)}


def _read(target):
    """-> {name: ndarray}; numpy-only container, nothing in it can run code."""
    with np.load(target, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def cached(path, parser):
    """parser(path) -> one of the structures CODECS knows (other parsers are simply not cached)."""
    target = _cache_file(path)
    codec = CODECS.get(parser.__name__)
    if target is None or codec is None:
        return parser(path)
    st = os.stat(path)
    stamp = np.asarray([FORMAT, st.st_size, st.st_mtime_ns], dtype=np.int64)
    try:
        arrs = _read(target)
        if np.array_equal(arrs.pop('stamp'), stamp):
            return codec[1](arrs)
    except Exception:      # noqa: BLE001 -- missing / stale / truncated / foreign file: parse
        pass
    value = parser(path)
    try:
        arrs = codec[0](value)
        os.makedirs(os.path.dirname(target), exist_ok=True)
        tmp = '%s.%d.tmp.npz' % (target, os.getpid())
        np.savez(tmp, stamp=stamp, **arrs)
        os.replace(tmp, target)
    except (OSError, ValueError):
        pass               # read-only dataset directory / unencodable value: no cache
    return value
