"""A small gflags-compatible flag registry (python-gflags is not installable here; jTransUP/models/base.py of the
reference depends on it).  Supports what the reference's command lines use: single- or double-dash flags,
`-flag value`, `-flag=value`, boolean `-flag` / `-noflag`, enum validation, FLAGS(argv), FLAGS.FlagValuesDict()."""
import sys


class FlagError(ValueError):
    pass


class _Flag(object):
    def __init__(self, name, default, kind, help_, enum=None):
        self.name, self.default, self.kind, self.help, self.enum = name, default, kind, help_, enum
        self.value = default

    def parse(self, text):
        if self.kind == 'bool':
            if isinstance(text, bool):
                return text
            low = str(text).lower()
            if low in ('true', 't', '1', 'yes'):
                return True
            if low in ('false', 'f', '0', 'no'):
                return False
            raise FlagError('flag -%s: %r is not a boolean' % (self.name, text))
        if self.kind == 'int':
            return int(text)
        if self.kind == 'float':
            return float(text)
        if self.kind == 'enum':
            if text not in self.enum:
                raise FlagError('flag -%s=%s: value should be one of <%s>' % (self.name, text, '|'.join(self.enum)))
            return text
        return text


class FlagValues(object):
    def __init__(self):
        object.__setattr__(self, '_flags', {})

    # -- definition
    def _define(self, name, default, kind, help_, enum=None):
        self._flags[name] = _Flag(name, default, kind, help_, enum)

    def is_defined(self, name):
        return name in self._flags

    # -- access
    def __getattr__(self, name):
        flags = object.__getattribute__(self, '_flags')
        if name in flags:
            return flags[name].value
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in self._flags:
            self._flags[name].value = value
        else:
            object.__setattr__(self, name, value)

    def FlagValuesDict(self):
        return {k: f.value for k, f in self._flags.items()}

    def reset(self):
        for f in self._flags.values():
            f.value = f.default

    # -- parsing: FLAGS(argv) like gflags; returns the non-flag arguments
    def __call__(self, argv):
        rest, i = [argv[0]] if argv else [], 1
        while i < len(argv):
            tok = argv[i]
            i += 1
            if not tok.startswith('-') or tok in ('-', '--'):
                rest.append(tok)
                continue
            body = tok.lstrip('-')
            name, eq, val = body.partition('=')
            if name in self._flags:
                flag = self._flags[name]
                if flag.kind == 'bool' and not eq:
                    flag.value = True
                    continue
                if not eq:
                    if i >= len(argv):
                        raise FlagError('flag -%s needs a value' % name)
                    val = argv[i]
                    i += 1
                flag.value = flag.parse(val)
            elif name.startswith('no') and name[2:] in self._flags and self._flags[name[2:]].kind == 'bool' and not eq:
                self._flags[name[2:]].value = False
            else:
                raise FlagError('Unknown command line flag %r' % name)
        return rest


FLAGS = FlagValues()


def DEFINE_string(name, default, help_=''):
    FLAGS._define(name, default, 'string', help_)


def DEFINE_integer(name, default, help_=''):
    FLAGS._define(name, default, 'int', help_)


def DEFINE_float(name, default, help_=''):
    FLAGS._define(name, default, 'float', help_)


def DEFINE_bool(name, default, help_=''):
    FLAGS._define(name, default, 'bool', help_)


DEFINE_boolean = DEFINE_bool


def DEFINE_enum(name, default, enum_values, help_=''):
    FLAGS._define(name, default, 'enum', help_, list(enum_values))
