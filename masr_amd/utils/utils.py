"""Config helper: YAML mappings with attribute access, like the objects the reference's code reads its configs through
(``configs.preprocess_conf.sample_rate``; masr/utils/utils.py:45-56)."""


class AttrDict(dict):
    """a dict whose keys are also attributes; a missing attribute raises KeyError like the reference's mapping does"""

    def __getattr__(self, name):
        return self[name]

    def __setattr__(self, name, value):
        self[name] = value


def dict_to_object(config):
    """nested dicts -> AttrDict (utils.py:37-45 of the reference); lists and everything else are returned as they are"""
    if isinstance(config, dict):
        return AttrDict((key, dict_to_object(value)) for key, value in config.items())
    return config
