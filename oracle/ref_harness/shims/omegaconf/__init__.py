class DictConfig(dict):
    pass
class ListConfig(list):
    pass
class OmegaConf:
    @staticmethod
    def set_struct(cfg, flag):
        return None
    @staticmethod
    def register_new_resolver(*a, **k):
        return None
    @staticmethod
    def to_container(cfg, resolve=True):
        return cfg
    @staticmethod
    def save(cfg, path):
        return None


def _load(path):
    # config container only: the repo's own YAML loader (resolved tree with attribute access)
    from pbhc_amd.utils.config import load_config
    return load_config(str(path), now="golden")


OmegaConf.load = staticmethod(_load)
OmegaConf.resolve = staticmethod(lambda cfg: None)
