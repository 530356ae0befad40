class EasyDict(dict):
    """attr-dict that mirrors keys into __dict__ (the reference tests `"k" in obj.__dict__`)."""
    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for k, v in d.items():
            self[k] = v
    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)) and not isinstance(v, str):
            return type(v)(cls._wrap(x) for x in v)
        return v
    def __setitem__(self, k, v):
        v = self._wrap(v)
        super().__setitem__(k, v)
        if isinstance(k, str):
            object.__setattr__(self, k, v)
    def __setattr__(self, k, v):
        self[k] = v
    def __delitem__(self, k):
        super().__delitem__(k)
        if isinstance(k, str) and k in self.__dict__:
            object.__delattr__(self, k)
    def pop(self, k, *a):
        if isinstance(k, str) and k in self.__dict__:
            object.__delattr__(self, k)
        return super().pop(k, *a)
    def update(self, *a, **k):
        for kk, vv in dict(*a, **k).items():
            self[kk] = vv
    def get(self, k, default=None):
        return super().get(k, default)
