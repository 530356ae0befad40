"""Inert stand-in for the absent `toolz` package (dict plumbing only; no arithmetic of the path runs through it)."""
from . import dicttoolz  # noqa: F401
