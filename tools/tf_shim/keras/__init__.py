"""stand-in for the `keras` imports at the top of the reference's model.py (nothing of it runs at inference)"""
from . import backend, regularizers, losses  # noqa: F401
