"""Stand-in for third-party omegaconf (absent here)."""


class ListConfig(list):
    pass


class DictConfig(dict):
    pass
