"""Label <-> index map in first-occurrence order (reference ``codes/preprocessing.py:10-59``).

The reference subclasses a scikit-learn module that no longer exists; this is a dependency-free restatement
with the same method names and return conventions (``transform`` returns a squeezed numpy int array).
"""
import numpy as np


class OrderedLabelEncoder(object):
    def fit(self, y):
        seen = {}
        for c in list(y):
            if c not in seen:
                seen[c] = len(seen)
        self.classes_ = np.asarray(list(seen.keys()))
        self.map_classes_ = seen
        return self

    def fit_transform(self, y):
        return self.fit(y).transform(y)

    def transform(self, y):
        y = list(y)
        unknown = sorted(set(c for c in y if c not in self.map_classes_))
        if unknown:
            raise ValueError('y contains new labels: %s' % str(unknown))
        return np.asarray([self.map_classes_[c] for c in y]).squeeze()

    def inverse_transform(self, ids):
        return self.classes_[np.asarray(ids, dtype=np.int64)]
