"""gin-config stand-in (TEST INFRASTRUCTURE, see oracle/tf_shim/__init__.py): the reference only uses
`@gin.configurable(name)` on `Options` (models/film_net/options.py:20) - identity decorator here;
`parse_config_file` reads the `scope.key = literal` lines of a .gin file so that the golden script
takes the hyper-parameters from training/config/film_net-L1.gin:17-23 itself."""
import ast
import sys
import types

__film_shim__ = True
_BINDINGS = {}


def configurable(name_or_fn=None, **_kw):
    def wrap(obj, scope):
        if isinstance(obj, type):
            orig = obj.__init__

            def __init__(self, *a, **kw):
                for k, v in _BINDINGS.get(scope, {}).items():
                    kw.setdefault(k, v)
                orig(self, *a, **kw)
            obj.__init__ = __init__
        return obj
    if callable(name_or_fn):
        return wrap(name_or_fn, name_or_fn.__name__)
    return lambda obj: wrap(obj, name_or_fn or obj.__name__)


def parse_config_file(path):
    with open(path) as f:
        for line in f:
            line = line.split('#')[0].strip()
            if '=' not in line:
                continue
            lhs, rhs = [s.strip() for s in line.split('=', 1)]
            if '.' not in lhs:
                continue
            scope, key = lhs.rsplit('.', 1)
            try:
                _BINDINGS.setdefault(scope, {})[key] = ast.literal_eval(rhs)
            except (ValueError, SyntaxError):
                pass  # references / macros: not used by the film_net scope


def clear_config():
    _BINDINGS.clear()


tf = types.ModuleType('gin.tf')
tf.__film_shim__ = True
sys.modules['gin.tf'] = tf
