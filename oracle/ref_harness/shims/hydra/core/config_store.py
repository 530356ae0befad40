class ConfigStore:
    @staticmethod
    def instance():
        return ConfigStore()

    def store(self, *a, **k):
        return None
