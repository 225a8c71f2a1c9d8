"""Small host-side helpers (config loading) mirroring crank/utils/utils.py:67-84."""
import copy
import os

import yaml

DEFAULT_YAML = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf", "default.yml")


def _merge(base, new):
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            _merge(base[k], v)
        else:
            base[k] = v


def load_yaml(ymlf=None, **overrides):
    """Recipe YAML deep-merged over the defaults.  The defaults come from
    $CRANK_DEFAULT_YAML when set (like the reference) else from crank_amd/conf/default.yml.
    Keyword overrides are merged last."""
    with open(os.environ.get("CRANK_DEFAULT_YAML", DEFAULT_YAML)) as fp:
        conf = yaml.safe_load(fp)
    if ymlf is not None:
        with open(ymlf) as fp:
            _merge(conf, yaml.safe_load(fp) or {})
    _merge(conf, copy.deepcopy(overrides))
    return conf
