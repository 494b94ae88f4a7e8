"""
Variable descriptors.  The reference passes dictionaries of Theano symbols around
(`syms`) and binds numpy values to them by sorted key order (theano_func_wrapper.py:53-105);
here a symbol is just (name, shape, dtype) -- enough to reproduce the packing order, the
`differentiable` filter (utils/grads.py:97-117) and the key checks of `_extract_vals`.
"""


class Sym(object):
    __slots__ = ('name', 'shape', 'dtype')

    def __init__(self, name, shape, dtype='float64'):
        self.name, self.shape, self.dtype = name, tuple(shape), dtype

    def __repr__(self):
        return "Sym(%s,%s,%s)" % (self.name, self.shape, self.dtype)

    def __str__(self):
        return self.name


def from_shapes(shapes, dtype='float64', int_keys=()):
    """{name: shape} (possibly nested) -> {name: Sym}."""
    out = {}
    for k, v in shapes.items():
        if isinstance(v, dict):
            out[k] = from_shapes(v, dtype, int_keys)
        else:
            out[k] = Sym(k, v, 'int8' if k in int_keys else dtype)
    return out


def differentiable(syms):
    """utils/grads.py:97-117: keep float symbols, replace everything else by {}."""
    diff = {}
    for k, v in syms.items():
        if isinstance(v, dict):
            diff[k] = differentiable(v)
        elif 'float' in v.dtype:
            diff[k] = v
        else:
            diff[k] = {}
    return diff


def flatten(d):
    """theano_func_wrapper._flatten (53-67): sorted-key DFS list of leaves."""
    out = []
    for k in sorted(d.keys()):
        v = d[k]
        if isinstance(v, dict):
            out.extend(flatten(v))
        else:
            out.append(v)
    return out


def check_bound(syms, vals, path=''):
    """_extract_vals' contract (theano_func_wrapper.py:69-105): every symbol must have a value."""
    for k in sorted(syms.keys()):
        if not isinstance(vals, dict) or k not in vals or vals[k] is None:
            raise Exception("Key %s not found in either vals or defaults!" % k)
        if isinstance(syms[k], dict):
            check_bound(syms[k], vals[k], path + k + '.')
