"""tensorflow_addons stand-in (TEST INFRASTRUCTURE, see oracle/tf_shim/__init__.py)."""
__film_shim__ = True
from . import image  # noqa: F401,E402
