"""Label files and config-string expansion (reference ``codes/utils/io_utils.py``)."""
import json

BLANK_TOKEN = '_'


def read_labels(labels_filepath, blank_label_id=0):
    """JSON list of symbols with the CTC blank '_' forced to ``blank_label_id`` (io_utils.py:6-27)."""
    with open(labels_filepath, 'r', encoding='utf8') as f:
        labels = list(json.load(f))
    if BLANK_TOKEN in labels:
        labels.remove(BLANK_TOKEN)
    labels.insert(blank_label_id, BLANK_TOKEN)
    return labels


def write_labels(labels, filepath):
    with open(filepath, 'w', encoding='utf8') as f:
        json.dump(labels, f, indent=4)


def expand_values(obj, **kwargs):
    """Recursive ``str.format(**kwargs)`` over a config tree, in place (io_utils.py:37-50)."""
    if isinstance(obj, str):
        return obj.format(**kwargs)
    if isinstance(obj, list):
        for i, v in enumerate(obj):
            obj[i] = expand_values(v, **kwargs)
    elif isinstance(obj, dict):
        for k, v in obj.items():
            obj[k] = expand_values(v, **kwargs)
    return obj


class AttrDict(dict):
    """Minimal stand-in for easydict.EasyDict (absent here): attribute access on nested dicts."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, list):
            return [AttrDict._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]
