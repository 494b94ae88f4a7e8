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


def compile_bound(syms):
    """The key tree of `syms` in check_bound's order, built once per model: ((key, subtree or None), ...)."""
    return tuple((k, compile_bound(syms[k]) if isinstance(syms[k], dict) else None) for k in sorted(syms.keys()))


def check_compiled(tree, vals):
    """check_bound against a compiled tree: the same contract and the same error, without sorting and re-reading the symbol
    dictionaries on every call (a population check walks 5 N + 3 nodes per evaluation of log p)."""
    for k, sub in tree:
        if not isinstance(vals, dict) or k not in vals or vals[k] is None:
            raise Exception("Key %s not found in either vals or defaults!" % k)
        if sub is not None:
            check_compiled(sub, vals[k])


def compile_paths(tree, prefix=()):
    """Every node of a compiled tree as a key path (internal nodes included: an empty sub-dictionary must still be bound)."""
    out = []
    for k, sub in tree:
        out.append(prefix + (k,))
        if sub is not None:
            out.extend(compile_paths(sub, prefix + (k,)))
    return out


def check_paths(paths, tree, vals):
    """The binding check as plain look-ups along the key paths; anything unusual goes through check_compiled, which raises
    the reference's error."""
    try:
        for path in paths:
            v = vals
            for k in path:
                v = v[k]
            if v is None:
                break
        else:
            if isinstance(vals, dict):
                return
    except (KeyError, TypeError, IndexError):
        pass
    check_compiled(tree, vals)
