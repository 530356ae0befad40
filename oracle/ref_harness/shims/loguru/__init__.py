class _Logger:
    def __getattr__(self, name):
        def f(*a, **k):
            return None
        return f
logger = _Logger()
