"""
Flatten / unflatten nested parameter dictionaries.

Behavioural counterpart of pyglm/utils/packvec.py (and of theano_func_wrapper._flatten):
leaves are visited depth-first in sorted-key order, which fixes the layout of every packed
per-neuron parameter / gradient vector (e.g. standard_glm: [bias, w_ir]).
Implemented as a single tree walk that yields (path, leaf) pairs.
"""
import numpy as np


def _walk(tree, path=()):
    """Yield (path, leaf) for every non-dict leaf, children in sorted-key order."""
    for key in sorted(tree):
        node = tree[key]
        if isinstance(node, dict):
            for item in _walk(node, path + (key,)):
                yield item
        else:
            yield path + (key,), node


def _put(tree, path, value):
    for key in path[:-1]:
        tree = tree.setdefault(key, {})
    tree[path[-1]] = value


def _skeleton(tree):
    """Same nesting with empty dicts kept (the shapes dict mirrors sub-dicts even when empty)."""
    return dict((k, _skeleton(v)) for k, v in tree.items() if isinstance(v, dict))


def packdict(var_dict, on_unpackable_type='raise'):
    """(vector, shapes): concatenation of all leaves and a dict of their shapes
    (packvec.py:17-45).  Empty lists are skipped; non-arrays raise unless told otherwise."""
    strict = on_unpackable_type.lower() == 'raise'
    shapes = _skeleton(var_dict)
    chunks = []
    for path, leaf in _walk(var_dict):
        if isinstance(leaf, list) and not leaf:
            continue
        if not isinstance(leaf, np.ndarray):
            if strict:
                raise Exception("Can only pack numpy arrays!")
            leaf = np.asarray(leaf)
        _put(shapes, path, leaf.shape)
        chunks.append(np.asarray(leaf, dtype=float).reshape(-1))
    return (np.concatenate(chunks) if chunks else np.zeros((0,))), shapes


def unpackdict(vec, shapes):
    """Inverse of packdict (packvec.py:58-83)."""
    vec = np.asarray(vec)
    out = _skeleton(shapes)
    cursor = 0
    for path, shp in _walk(shapes):
        if not isinstance(shp, tuple):
            raise Exception("Can only unpack shape tuples!")
        size = int(np.prod(shp))
        _put(out, path, vec[cursor:cursor + size].reshape(shp))
        cursor += size
    return out


def pack(var_list):
    """List form (packvec.py:3-15)."""
    arrays = [np.asarray(v) for v in var_list]
    flat = [a.reshape(-1) for a in arrays]
    return (np.concatenate(flat) if flat else np.zeros((0,))), [a.shape for a in arrays]


def unpack(vec, shapes):
    """List form inverse (packvec.py:47-56)."""
    sizes = [int(np.prod(s)) for s in shapes]
    assert sum(sizes) == len(vec), "Unpack was called with incorrect shapes!"
    bounds = np.cumsum([0] + sizes)
    return [np.reshape(vec[a:b], s) for a, b, s in zip(bounds[:-1], bounds[1:], shapes)]


def get_vars(syms, vars):
    """The part of `vars` that `syms` names (packvec.py:86-96)."""
    picked = {}
    for name, sym in syms.items():
        if name not in vars:
            raise AssertionError("ERROR: syms key %s not found in vars!" % name)
        picked[name] = get_vars(sym, vars[name]) if isinstance(sym, dict) else vars[name]
    return picked


def set_vars(syms, vars, vals):
    """Write `vals` into `vars` for every variable `syms` names (packvec.py:98-113);
    `syms` may also be a single key."""
    if not isinstance(syms, dict):
        if syms not in vars:
            raise Exception("Can only set variables for a dictionary of symbolic vars"
                            "or a specific key in vars")
        vars[syms] = vals
        return vars
    for name, sym in syms.items():
        for where, d in (("vars", vars), ("vals", vals)):
            if name not in d:
                raise AssertionError("ERROR: syms key %s not found in %s!" % (name, where))
        vars[name] = set_vars(sym, vars[name], vals[name]) if isinstance(sym, dict) else vals[name]
    return vars


def get_shapes(x, syms):
    """Shapes of the variables `syms` names in state `x` (packvec.py:115-117)."""
    return packdict(get_vars(syms, x))[1]
