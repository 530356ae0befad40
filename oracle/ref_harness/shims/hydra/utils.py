import importlib
def get_class(path):
    mod, name = path.rsplit(".", 1)
    return getattr(importlib.import_module(mod), name)
def instantiate(config, *args, **kwargs):
    cls = get_class(config["_target_"])
    kw = {k: v for k, v in config.items() if not k.startswith("_")}
    kw.update(kwargs)
    return cls(*args, **kw)
