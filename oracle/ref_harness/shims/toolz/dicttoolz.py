def valmap(func, d):
    return {k: func(v) for k, v in d.items()}
