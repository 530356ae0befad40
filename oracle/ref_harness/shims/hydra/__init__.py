def main(*a, **k):
    def deco(fn):
        return fn
    return deco
from . import utils  # noqa
