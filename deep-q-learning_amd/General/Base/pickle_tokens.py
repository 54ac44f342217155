"""Reading the reference's checkpoints -- `params.pickle` / `opt_state.pickle`, written by General/Base/utils.py:21-29 --
WITHOUT unpickling them. The reference's `generate_loading` (utils.py:32-40) calls `pickle.load`, which imports
`jax._src.device_array`, `optax._src.transform` ... and runs their reconstructors; none of that exists here and nothing in
a checkpoint should get to execute. `pickletools.genops` only tokenises the byte stream (opcode, argument, position):
nothing is imported, constructed or called. From the tokens this module takes the haiku module / leaf keys, the shape
tuples and the raw little-endian byte strings of numpy's ndarray state `(version, shape, dtype, is_fortran, rawdata)`.

Supported: a haiku params dict `{module: {'w'|'b': array}}` of jax DeviceArrays / numpy arrays with f4 leaves, and an optax
adam / adamw state `(ScaleByAdamState(count, mu, nu), EmptyState...)` over the same tree. Anything else raises ValueError."""
from __future__ import annotations

import pickletools

import numpy as np

_INT_OPS = ("BININT", "BININT1", "BININT2", "LONG1")
_STR_OPS = ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8")
_BYTES_OPS = ("BINBYTES", "SHORT_BINBYTES", "BINBYTES8")
_DTYPES = {"f4": "<f4", "i4": "<i4", "f8": "<f8", "i8": "<i8"}


def read_arrays(path):
    """[(module key, leaf key, numpy dtype string, shape, raw bytes)] in stream order. Strings fetched again through the
    pickle memo (BINGET) are followed with a memo table of strings only (memo index = number of MEMOIZE tokens before)."""
    with open(path, "rb") as f:
        data = f.read()
    memo, n_memo = {}, 0
    pending_str = None                      # a string token not yet followed by anything else (the MEMOIZE target)
    module = leaf = dtype = None
    ints, shape, out = [], None, []

    def see(s):
        nonlocal module, leaf, dtype
        if s in ("w", "b"):
            leaf = s
        elif s in _DTYPES:
            dtype = _DTYPES[s]
        elif "/" in s or s.startswith("model") or s.startswith("linear"):
            module = s

    for op, arg, _ in pickletools.genops(data):
        n = op.name
        if n == "MEMOIZE":
            if pending_str is not None:
                memo[n_memo] = pending_str
            n_memo += 1
            pending_str = None
            continue
        pending_str = None
        if n in _STR_OPS:
            pending_str = arg
            see(arg)
        elif n in ("BINGET", "LONG_BINGET"):
            if arg in memo:
                see(memo[arg])
        elif n in ("BINPUT", "LONG_BINPUT", "PUT", "GET"):
            raise ValueError("protocol < 4 memo opcodes are not supported by this reader")
        elif n in _INT_OPS:
            ints.append(int(arg))
        elif n in ("TUPLE1", "TUPLE2", "TUPLE3", "EMPTY_TUPLE", "TUPLE"):
            k = {"EMPTY_TUPLE": 0, "TUPLE1": 1, "TUPLE2": 2}.get(n)
            if k is not None and len(ints) >= k:
                shape = tuple(ints[len(ints) - k:]) if k else ()
            ints = []
        elif n in _BYTES_OPS and len(arg) >= 4:
            if dtype is None or shape is None:
                raise ValueError("array bytes before any dtype / shape token")
            if len(arg) != int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize:
                raise ValueError(f"{module}/{leaf}: {len(arg)} bytes do not fill shape {shape} of {dtype}")
            out.append((module, leaf, dtype, shape, bytes(arg)))
    return out


def read_haiku_params(path):
    """{module: {'w' | 'b': float32 ndarray}} in the stream's (= haiku's creation) order."""
    tree = {}
    for module, leaf, dtype, shape, raw in read_arrays(path):
        if dtype != "<f4" or module is None or leaf is None:
            raise ValueError(f"unexpected leaf {module}/{leaf} of {dtype}")
        if leaf in tree.setdefault(module, {}):
            raise ValueError(f"leaf {module}/{leaf} appears twice")
        tree[module][leaf] = np.frombuffer(raw, dtype).reshape(shape).copy()
    if not tree:
        raise ValueError("no arrays found")
    return tree


def read_adam_state(path):
    """(count, mu tree, nu tree, number of trailing EmptyState) of an optax adam / adamw state: one i4 scalar, the leaves of mu,
    then the leaves of nu (jax flattens a dict by sorted key, so `b` precedes `w` here)."""
    arrs = read_arrays(path)
    if not arrs or arrs[0][2] != "<i4" or (len(arrs) - 1) % 2:
        raise ValueError("not an optax ScaleByAdamState stream")
    count = int(np.frombuffer(arrs[0][4], "<i4")[0])
    half = (len(arrs) - 1) // 2
    trees = []
    for part in (arrs[1:1 + half], arrs[1 + half:]):
        t = {}
        for module, leaf, dtype, shape, raw in part:
            if dtype != "<f4":
                raise ValueError("moment leaves must be f4")
            t.setdefault(module, {})[leaf] = np.frombuffer(raw, dtype).reshape(shape).copy()
        trees.append(t)
    # optax.EmptyState is an empty NamedTuple: one NEWOBJ token per instance, after the class name's (single, memoised) string
    n_empty, seen = 0, False
    with open(path, "rb") as f:
        for op, arg, _ in pickletools.genops(f.read()):
            if op.name in _STR_OPS and arg == "EmptyState":
                seen = True
            elif seen and op.name in ("NEWOBJ", "NEWOBJ_EX"):
                n_empty += 1
    return count, trees[0], trees[1], n_empty
