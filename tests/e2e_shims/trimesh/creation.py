import itertools

import numpy as np


class _Box:
    def __init__(self):
        # the 8 corners of the unit cube centred at the origin, in binary (x, y, z) order
        self.vertices = np.array(list(itertools.product((-0.5, 0.5), repeat=3)), dtype=np.float64)


def box(*a, **k):
    return _Box()
