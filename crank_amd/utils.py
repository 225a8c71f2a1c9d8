"""Small host-side helpers mirroring crank/utils/utils.py: config loading (:67-84) and the Kaldi-style
list files of a recipe (:33-64)."""
import copy
import os

import yaml

from .config import cfg  # noqa: E402

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
    with open(cfg.default_yaml or DEFAULT_YAML) as fp:
        conf = yaml.safe_load(fp)
    if ymlf is not None:
        with open(ymlf) as fp:
            _merge(conf, yaml.safe_load(fp) or {})
    _merge(conf, copy.deepcopy(overrides))
    return conf


def _table(path, min_cols=2):
    """Whitespace separated list file -> [(key, [values...])], blank lines skipped."""
    rows = []
    with open(path) as fp:
        for ln, line in enumerate(fp, 1):
            cols = line.split()
            if not cols:
                continue
            if len(cols) < min_cols:
                raise ValueError(f"{path}:{ln}: expected at least {min_cols} columns, got {line!r}")
            rows.append((cols[0], cols[1:]))
    return rows


def open_featsscp(featsscp):
    """feats.scp: "<utterance id> <feature file>" per line -> {id: file} (crank/utils/utils.py:33-39)."""
    out = {}
    for uid, vals in _table(featsscp):
        if len(vals) != 1:
            raise ValueError(f"{featsscp}: {uid} has {len(vals)} columns after the id, expected 1")
        out[uid] = vals[0]
    return out


def open_scpdir(scpdir):
    """A list directory (wav.scp, utt2spk, spk2utt) -> the dict the reference's train / dataset code reads:
    wav, feats (filled by the caller), utt2spk, spk2utt and the speaker list in spk2utt order
    (crank/utils/utils.py:42-64)."""
    d = os.fspath(scpdir)
    scp = {"wav": {}, "feats": {}, "utt2spk": {}, "spk2utt": {}, "spkrs": []}
    for key, name in (("wav", "wav.scp"), ("utt2spk", "utt2spk")):
        for uid, vals in _table(os.path.join(d, name)):
            if len(vals) != 1:
                raise ValueError(f"{name}: {uid} has {len(vals)} columns after the id, expected 1")
            scp[key][uid] = vals[0]
    for spkr, utts in _table(os.path.join(d, "spk2utt"), min_cols=1):
        scp["spkrs"].append(spkr)
        scp["spk2utt"][spkr] = utts
    return scp
