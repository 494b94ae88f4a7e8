"""Data-driven initialisation (counterpart of pyglm/inference/smart_init.py:7-18).
Only the dense-graph initialisation is on the hot path; the STA initialisation of
stimulus weights (smart_init.py:28-98) is a SURVEY §8(f) "next" row and is a no-op for
NoStimulus models exactly like the reference (smart_init.py:42-43)."""
import numpy as np


def initialize_with_dense_graph(population, data, x0):
    if 'A' in x0['net']['graph']:
        x0['net']['graph']['A'] = np.ones_like(x0['net']['graph']['A'])


def initialize_with_data(population, data, x0, Ns=None):
    initialize_with_dense_graph(population, data, x0)
