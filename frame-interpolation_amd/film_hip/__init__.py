"""film_hip: host side of the MI355X-native FILM inference engine (ctypes over libfilm_hip.so)."""
from .options import Options, PUBLISHED, TINY  # noqa: F401
