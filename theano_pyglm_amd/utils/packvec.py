"""
Flatten / unflatten nested parameter dicts in sorted-key depth-first order
(counterpart of pyglm/utils/packvec.py and theano_func_wrapper._flatten; the order
defines the layout of every packed parameter / gradient vector).
"""
import numpy as np


def _items(d):
    return sorted(d.items(), key=lambda kv: kv[0])


def packdict(var_dict, on_unpackable_type='raise'):
    """packvec.py:17-45.  Returns (vector, shapes) with shapes mirroring the dict."""
    parts, shapes = [], {}
    for key, val in _items(var_dict):
        if isinstance(val, dict):
            sub, sshapes = packdict(val, on_unpackable_type)
            parts.append(sub)
            shapes[key] = sshapes
            continue
        if isinstance(val, list) and len(val) == 0:
            continue
        if not isinstance(val, np.ndarray):
            if on_unpackable_type.lower() == 'raise':
                raise Exception("Can only pack numpy arrays!")
            val = np.asarray(val)
        shapes[key] = val.shape
        parts.append(np.reshape(val, (val.size,)).astype(float))
    vec = np.concatenate(parts) if parts else np.zeros((0,))
    return vec, shapes


def _unpack(vec, shapes, offset):
    out = {}
    for key, shp in _items(shapes):
        if isinstance(shp, dict):
            out[key], offset = _unpack(vec, shp, offset)
        elif isinstance(shp, tuple):
            n = int(np.prod(shp))
            out[key] = np.reshape(vec[offset:offset + n], shp)
            offset += n
        else:
            raise Exception("Can only unpack shape tuples!")
    return out, offset


def unpackdict(vec, shapes):
    """packvec.py:58-83."""
    out, used = _unpack(np.asarray(vec), shapes, 0)
    return out


def pack(var_list):
    """packvec.py:3-15."""
    shapes = [np.shape(v) for v in var_list]
    vec = np.concatenate([np.reshape(v, (-1,)) for v in var_list]) if var_list else np.zeros((0,))
    return vec, shapes


def unpack(vec, shapes):
    """packvec.py:47-56."""
    out, off = [], 0
    for shp in shapes:
        n = int(np.prod(shp))
        out.append(np.reshape(vec[off:off + n], shp))
        off += n
    assert off == len(vec), "Unpack was called with incorrect shapes!"
    return out


def get_vars(syms, vars):
    """packvec.py:86-96: the sub-dict of `vars` named by `syms`."""
    out = {}
    for k, v in syms.items():
        assert k in vars.keys(), "ERROR: syms key %s not found in vars!" % k
        out[k] = get_vars(v, vars[k]) if isinstance(v, dict) else vars[k]
    return out


def set_vars(syms, vars, vals):
    """packvec.py:98-113."""
    if isinstance(syms, dict):
        for k, v in syms.items():
            assert k in vars.keys(), "ERROR: syms key %s not found in vars!" % k
            assert k in vals.keys(), "ERROR: syms key %s not found in vals!" % k
            if isinstance(v, dict):
                vars[k] = set_vars(v, vars[k], vals[k])
            else:
                vars[k] = vals[k]
    elif syms in vars:
        vars[syms] = vals
    else:
        raise Exception("Can only set variables for a dictionary of symbolic vars"
                        "or a specific key in vars")
    return vars


def get_shapes(x, syms):
    """packvec.py:115-117."""
    return packdict(get_vars(syms, x))[1]
