class HydraConfig:
    @staticmethod
    def get():
        raise RuntimeError("hydra stand-in: no runtime configuration")
